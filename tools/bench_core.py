#!/usr/bin/env python3
"""Developer aid: the LDPC core alone.  N codewords of pure noise (never converge: every unit runs all its iterations)
through ria_gpu_ldpc_decode_batch (fast_rows_kernel); prints ns per codeword-iteration chip-wide and the cycles one CU
spends per codeword-iteration, the figure DESIGN.md §4 prices against the VALU-issue and LDS-array bounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ria_amd.engine import RxEngine
rate = sys.argv[1] if len(sys.argv) > 1 else "R1_2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 196608
iters = 80
e = RxEngine("QAM16", rate, max_batch=64)
g = torch.Generator(device="cuda"); g.manual_seed(7)
llr = torch.randn((n, 648), generator=g, device="cuda", dtype=torch.float32) * 3.0
out, ok, it = e.ldpc_decode(llr, iters, 0.9375)
torch.cuda.synchronize()
okc = ok.cpu().numpy().astype(bool); itc = it.cpu().numpy().astype(int)
total_it = int(itc[~okc].sum() + (itc[okc] + 1).sum())     # a converged codeword ran lastIterations()+1 check passes
assert okc.mean() < 0.01
best = 1e9
for rep in range(5):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); e.ldpc_decode(llr, iters, 0.9375); t1.record(); torch.cuda.synchronize()
    best = min(best, t0.elapsed_time(t1))
ns = best * 1e6 / total_it
print(f"{os.environ.get('RIA_GPU_LIB', 'default')}: {rate} {n} cw x {iters} it: {best:.2f} ms, {ns:.4f} ns per cw-iteration chip-wide, "
      f"{ns * 256 * 2.4:.0f} CU-cycles at 2.4 GHz")
