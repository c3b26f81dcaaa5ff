import sys, json, torch
sys.path.insert(0, "/root/repo")
from ria_amd.engine import RxEngine
e = RxEngine("QAM16", "R1_2", max_batch=32768)
info = e.make_frames(1, 0, 32768)
x0 = e.tx(info, peak=0.8)
res = {}
for kind in (0, 2):
    x = x0.clone()
    e.channel_exact_(x, kind, 20.0, 5); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    x = x0.clone(); ev[0].record(); e.channel_exact_(x, kind, 20.0, 5); ev[1].record(); torch.cuda.synchronize()
    res[kind] = round(ev[0].elapsed_time(ev[1]), 2)
    res["crc%d" % kind] = int(x.view(torch.int32).sum().item())
print(json.dumps(res))
