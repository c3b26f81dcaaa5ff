#!/usr/bin/env python3
"""Single-frame latency of the plug-in use case (VERDICT r01 missing #4): the real caller decodes ONE frame per call
(gui::StreamingDecoder, streaming_decoder.cpp:1347-1363 process(), :2821 decodeFixedFrame()).  Measures, per frame,
wall time of
  process path  ria_gpu_rx_frames_host(n=1, RIA_RX_DEMOD_ONLY)       = IWaveform::process + getSoftBits
  decode path   ria_gpu_decode_frames_host(n=1, RIA_DECODE_FULL)     = v2::decodeFixedFrame
  fused         ria_gpu_rx_frames_host(n=1, RIA_DECODE_FULL)
from ordinary host buffers (PCIe both ways, pinned staging inside the library), on AWGN 20 dB and Watterson moderate
20 dB frames of the named shape, next to the compiled reference (oracle/_ref) on one host core for the same frames
(BASELINE.md section 2 quotes 3.47 ms / 50.1 ms per frame for the reference on the survey's Xeon).
Prints one JSON object; --out writes it under profiles/."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pct(a, q):
    return round(float(np.percentile(np.asarray(a) * 1e3, q)), 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--ref-frames", type=int, default=40)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    from ria_amd import capi
    from ria_amd.engine import RxEngine
    e = RxEngine("QAM16", "R1_2", max_batch=64)
    L, h, g = e.lib, e.h, e.geo
    res = {"frame": "OFDM QAM16 R1/2, 18432 samples in, 160 bytes out", "frames_per_channel": a.frames, "unit": "ms per frame"}
    for name, kind in (("awgn_20dB", 0), ("moderate_20dB", 2)):
        info = e.make_frames(4242, 0, a.frames)
        x = e.tx(info, peak=0.8)
        e.channel_exact_(x, kind, 20.0, 4242, first_frame=0)
        xs = x.cpu().numpy()
        llr = np.zeros(g.llrs_per_frame, np.float32)
        out = np.zeros(g.info_bytes_per_frame, np.uint8)
        ds = capi.DecodeStatus()
        fs = capi.FrameStatus()
        t_proc, t_dec, t_fused, ok = [], [], [], 0
        for rep in range(2):                      # first pass warms clocks, staging, code objects
            t_proc, t_dec, t_fused, ok = [], [], [], 0
            for f in range(a.frames):
                fr = np.ascontiguousarray(xs[f])
                t0 = time.perf_counter()
                rc = L.ria_gpu_rx_frames_host(h, fr.ctypes.data, None, 1, capi.RX_DEMOD_ONLY, None, None, llr.ctypes.data, C.byref(fs))
                t1 = time.perf_counter()
                assert rc == 0
                rc = L.ria_gpu_decode_frames_host(h, llr.ctypes.data, g.llrs_per_frame, 1, capi.DECODE_FULL, out.ctypes.data, C.byref(ds))
                t2 = time.perf_counter()
                assert rc == 0
                rc = L.ria_gpu_rx_frames_host(h, fr.ctypes.data, None, 1, capi.DECODE_FULL, out.ctypes.data, C.byref(ds), None, None)
                t3 = time.perf_counter()
                assert rc == 0
                t_proc.append(t1 - t0); t_dec.append(t2 - t1); t_fused.append(t3 - t2)
                ok += int(ds.frame_valid)
        r = {"frames_valid": ok}
        for k, t in (("process", t_proc), ("decodeFixedFrame", t_dec), ("fused", t_fused)):
            r[k] = {"mean": round(float(np.mean(t)) * 1e3, 4), "p50": pct(t, 50), "p90": pct(t, 90), "p99": pct(t, 99)}
        # the compiled reference on ONE host core, same frames
        import pyoracle as po
        if po.Ref.available():
            R = po.Ref()
            kept = R.rx_open(po.QAM16, po.R1_2)    # one configured waveform object, reset() per frame (streaming_decoder.cpp:723)
            R.rx_process_kept(kept, xs[0])
            tp, td = [], []
            for f in range(min(a.ref_frames, a.frames)):
                t0 = time.perf_counter()
                l = R.rx_process_kept(kept, xs[f])
                t1 = time.perf_counter()
                R.decode_fixed_frame(l, po.R1_2, True, 188)
                t2 = time.perf_counter()
                tp.append(t1 - t0); td.append(t2 - t1)
            r["reference_one_core"] = {"process_mean": round(float(np.mean(tp)) * 1e3, 3), "decodeFixedFrame_mean": round(float(np.mean(td)) * 1e3, 3),
                                       "total_mean": round(float(np.mean(tp) + np.mean(td)) * 1e3, 3), "frames": len(tp)}
        res[name] = r
    s = json.dumps(res, indent=1)
    print(s)
    if a.out:
        with open(a.out, "w") as fo:
            fo.write(s + "\n")


if __name__ == "__main__":
    main()
