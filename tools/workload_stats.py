#!/usr/bin/env python3
"""Developer aid: decode-cascade workload statistics of the bench configuration (how many codeword
decodes of which kind one 25k-frame step really contains)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ria_amd import capi
from ria_amd.engine import RxEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
e = RxEngine("QAM16", "R1_2", max_batch=B)
info = e.make_frames(20261004, 0, B)
x = e.tx(info, peak=0.8)
e.channel_(x, 2, 20.0, 20261004, first_frame=0)
llr, _ = e.demod(x, want_status=False)
d, st = e.decode(llr, flags=capi.DECODE_PHASE0 | capi.DECODE_PERTURB)
s = e.decode_status(st)
att, it, ok = s["attempts"].astype(int), s["iterations"].astype(int), s["cw_ok"].astype(bool)
n_cw = att.size
print("codewords", n_cw, "ok", int(ok.sum()))
print("first-try ok", int((att == 1).sum()), "mean iters", it[att == 1].mean())
print("phase0 ok (attempts 2..5)", int(((att >= 2) & (att <= 5) & ok).sum()))
casc = att > 5
print("entered cascade", int(casc.sum()), "of which ok", int((casc & ok).sum()), "hopeless", int((casc & ~ok).sum()))
a_idx = att[casc & ok] - 5
print("cascade winners: attempt index histogram (1..34):", np.bincount(a_idx, minlength=35)[1:].tolist())
units_hopeless = int((casc & ~ok).sum()) * 34
units_win = int(a_idx.sum())
print("cascade units (upper bound, attempt-major skipping aside):", units_hopeless + units_win)
print("phase0 units:", int((att >= 2).sum()) * 4, "(list1 may be larger: chain-dependent entries)")
