#!/usr/bin/env python3
"""Developer aid: per-phase cycle stamps of demod_frames_kernel on the bench workload (RIA_DEBUG_DEMOD_STAMPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ria_amd.engine import RxEngine
n = 8192
e = RxEngine("QAM16", "R1_2", max_batch=n)
info = e.make_frames(20261004, 0, n); x = e.tx(info, peak=0.8); e.channel_exact_(x, 2, 20.0, 20261004)
st = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
os.environ["RIA_DEBUG_DEMOD_STAMPS"] = hex(st.data_ptr())
llr, fs = e.demod(x)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
f = e.frame_status(fs)
rer = f["cfo_hz"] != 0
print("frames", n, "rerun fraction", rer.mean())
for name, m in (("all", np.ones(n, bool)), ("no rerun", ~rer), ("rerun", rer)):
    print(name, "fft pass0 %.0f  lts(+rerun fft) %.0f  data symbols %.0f  total %.0f" % tuple(s[m].mean(axis=0)))
