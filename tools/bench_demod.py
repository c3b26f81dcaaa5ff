#!/usr/bin/env python3
"""Developer aid: the demodulator alone on the bench workload (ria_gpu_demod_batch, samples resident in HBM):
frames/s, ms per batch and the algorithmic GB/s against the HBM peak.  usage: tools/bench_demod.py [--batch N] [--reps R] [--mod QAM16 --rate R1_2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ria_amd.engine import RxEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=100000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--mod", default="QAM16")
ap.add_argument("--rate", default="R1_2")
ap.add_argument("--channel", type=int, default=2)
ap.add_argument("--snr", type=float, default=20.0)
a = ap.parse_args()
e = RxEngine(a.mod, a.rate, max_batch=a.batch)
info = e.make_frames(20261004, 0, a.batch)
x = e.tx(info, peak=0.8)
e.channel_exact_(x, a.channel, a.snr, 20261004, first_frame=0)
torch.cuda.synchronize()
e.demod(x, want_status=False)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(a.reps):
    e.demod(x, want_status=False)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / a.reps
g = e.geo
algo = (g.frame_samples + g.llrs_per_frame) * 4 * a.batch
print(json.dumps({"what": "demodulator alone", "mod": a.mod, "rate": a.rate, "batch": a.batch, "ms_per_batch": round(ms, 3),
                  "frames_per_s": round(a.batch / ms * 1e3), "algorithmic_GBps": round(algo / ms / 1e6, 1), "hbm_frac": round(algo / ms / 1e6 / 8000.0, 4)}))
