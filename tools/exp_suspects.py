#!/usr/bin/env python3
"""Developer aid: distribution of the CRC recovery's suspect counts on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import pyoracle as po
from ria_amd import capi
from ria_amd.engine import RxEngine
n = 25000
e = RxEngine("QAM16", "R1_2", max_batch=n)
info = e.make_frames(20261004, 0, n); x = e.tx(info, peak=0.8); e.channel_exact_(x, 2, 20.0, 20261004)
llr, _ = e.demod(x, want_status=False)
d, st = e.decode(llr, flags=capi.DECODE_PHASE0 | capi.DECODE_PERTURB)
s = e.decode_status(st)
fl = np.nonzero(s["needs_recovery"])[0]
g = po.Oracle().gather_table(188).reshape(4, 648)
L = llr.cpu().numpy()[fl]; D = d.cpu().numpy()[fl]
bits = np.unpackbits(D.reshape(len(fl), 4, 40), axis=2, bitorder="little")[:, :, :320]     # LSB-first index as the reference reads it
ch = (L[:, g[:, :320]] < 0).astype(np.uint8)
ns = (bits != ch).reshape(len(fl), -1).sum(axis=1)
print("flagged", len(fl), "suspects: mean %.1f median %d p90 %d p99 %d max %d" % (ns.mean(), np.median(ns), np.percentile(ns, 90), np.percentile(ns, 99), ns.max()))
print("hist", np.histogram(ns, bins=[0, 30, 60, 100, 150, 200, 320, 500, 1300])[0].tolist())
stt = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
os.environ["RIA_DEBUG_REC_STAMPS"] = hex(stt.data_ptr())
os.environ["RIA_NO_SPLIT"] = "1"
d2, st2 = e.decode(llr, flags=capi.DECODE_FULL)
torch.cuda.synchronize()
t = stt.cpu().numpy()[fl].astype(np.float64)
tot = t[:, 6] - t[:, 0]
hdr = t[:, 1] == 0            # never reached the suspect stage: header case or early single-bit success
print("per-frame total cycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
for name, m in (("reached suspects", ~hdr), ("did not", hdr)):
    if m.sum(): print(name, int(m.sum()), "mean %.0f p99 %.0f max %.0f" % (tot[m].mean(), np.percentile(tot[m], 99), tot[m].max()), "recovered", int((t[m, 5] == 1).sum()))
