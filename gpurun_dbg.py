import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ria_amd.engine import RxEngine
e = RxEngine("QAM16","R1_2")
n = 2048
info = e.make_frames(1, 0, n); x = e.tx(info, 0.8); e.channel_(x, 2, 20.0, 3)
dbg = torch.zeros((n,4), dtype=torch.int64, device='cuda')
os.environ["RIA_DEBUG_DEMOD_STAMPS"] = hex(dbg.data_ptr())
llr, st = e.demod(x)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
print("cycles per frame-block: fft", d[:,0].mean(), "lts", d[:,1].mean(), "estimator", d[:,2].mean(), "total", d[:,3].mean())
print("max", d.max(axis=0), "min", d.min(axis=0))
s = e.frame_status(st)
print("frames with cfo != 0 (LTS rerun):", (s["cfo_hz"] != 0).sum(), "of", n)
big = d[:,0] > 2*np.median(d[:,0])
print("fft phase slow frames:", big.sum(), "median fft", np.median(d[:,0]), "median est", np.median(d[:,2]))
