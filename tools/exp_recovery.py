#!/usr/bin/env python3
"""Developer aid: how many frames the CRC recovery sees on the bench workload and what its kernels cost (run under
rocprofv3 --kernel-trace for the per-kernel durations; RIA_NO_SPLIT=1 keeps one launch per kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ria_amd import capi
from ria_amd.engine import RxEngine
n = 25000
e = RxEngine("QAM16", "R1_2", max_batch=n)
info = e.make_frames(20261004, 0, n); x = e.tx(info, peak=0.8); e.channel_exact_(x, 2, 20.0, 20261004)
llr, _ = e.demod(x, want_status=False)
d, st = e.decode(llr, flags=capi.DECODE_PHASE0 | capi.DECODE_PERTURB)
s = e.decode_status(st)
print("flagged (needs_recovery)", int(s["needs_recovery"].sum()), "all cw ok", int(s["cw_ok"].all(axis=1).sum()), "valid", int(s["frame_valid"].sum()))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d2, st2 = e.decode(llr, flags=capi.DECODE_FULL)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    d3, st3 = e.decode(llr, flags=capi.DECODE_PHASE0 | capi.DECODE_PERTURB)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("decode full %.3f ms, without recovery %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
s2 = e.decode_status(st2)
print("after recovery: valid", int(s2["frame_valid"].sum()), "recovered", int(s2["frame_valid"].sum() - s["frame_valid"].sum()))
