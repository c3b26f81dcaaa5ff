#!/usr/bin/env python3
"""Developer aid: timeline of the LAST bench step from a rocprofv3 --kernel-trace CSV: per kernel launch start/end relative to
the step's first kernel, plus the chip's occupancy by kernel family over time.
usage: tools/trace_timeline.py <kernel_trace.csv> [n_last_steps=1]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ria::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ks.sort()
# a step starts with the demodulator kernels of its parts (demod_fft / decide / walk / est, or demod_frames with RIA_DEMOD_FUSED=1)
dem = [k for k in ks if k[2].startswith("demod_")]
# group demodulator launches that start within 30 ms of each other
groups, cur = [], []
for k in dem:
    if cur and k[0] - cur[-1][0] > 30e6:
        groups.append(cur); cur = []
    cur.append(k)
if cur: groups.append(cur)
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# pick a step in the timed region: the group before the last few (the tail of the run holds the per-kernel timing section)
cand = [g for g in groups if len(g) >= 2]
g = cand[min(len(cand) - 1, 2)] if cand else groups[-1]
t0 = g[0][0]
nxt = [h[0][0] for h in groups if h[0][0] > t0]
t1 = min(nxt) if nxt else max(k[1] for k in ks)
print(f"step of {len(g)} parts, {(t1 - t0) / 1e6:.3f} ms to the next step's first kernel")
sel = [k for k in ks if t0 <= k[0] < t1]
for s, e, n, q in sel:
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}  q{q:>4}  {n[:60]}")
