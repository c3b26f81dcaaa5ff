import sys, torch
sys.path.insert(0, "/root/repo")
from ria_amd.engine import RxEngine
e = RxEngine("QAM16", "R1_2", max_batch=16)
x = torch.randn((8000, 74 * 512), device="cuda") * 0.1
for _ in range(4):
    e.mcdpsk_demod(x, 10, 1, 1)
torch.cuda.synchronize()
