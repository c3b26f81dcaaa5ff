#!/usr/bin/env python3
"""Throughput of the acquisition / MC-DPSK kernels on synthetic buffers (SURVEY.md §8d C4, C1 shapes).
Prints one JSON object; not the contract bench (bench.py is)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ria_amd.engine import RxEngine


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    e = RxEngine("QAM16", "R1_2")
    rng = np.random.default_rng(1)
    res = {}
    # ZC: preamble (2512) at a random offset of a 4512-sample noisy buffer
    pre = e.zc_preamble(5)
    n = 20000
    base = torch.randn((n, 4512), device="cuda") * 0.2
    offs = rng.integers(0, 2000, n)
    pt = torch.from_numpy(pre).cuda()
    for i in range(0, n, 1):
        pass
    idx = torch.from_numpy(offs).cuda()[:, None] + torch.arange(2512, device="cuda")[None, :]
    base.scatter_add_(1, idx, pt[None, :].expand(n, -1))
    t = timed(lambda: e.sync_zc(base, 0.3, 15))
    r = e.sync_zc(base, 0.3, 15)
    # algorithmic work of ZCSync::detect on this shape: per root 113 coarse + 63 fine + 3 extra lags, 1016-term complex
    # correlations with the received energy (11 flops per term), four roots
    zc_flop = 4 * (113 + 63 + 3) * 1016 * 11
    res["zc"] = {"buffers": n, "samples": 4512, "ms": round(t * 1e3, 2), "preambles_per_s": round(n / t), "detected": int(r["detected"].sum()),
                 "GBps_algorithmic": round(n * 4512 * 4 / t / 1e9, 2),
                 "roofline": {"bound": "valu", "achieved": round(n * zc_flop / t / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s fp32 vector (no FMA: contraction is off for bit-exactness, so half of the FMA peak is the ceiling)",
                              "frac": round(n * zc_flop / t / 1e12 / 157.3, 4)}}
    # dual chirp: 57600-sample preamble in 120000-sample buffers
    ch = torch.from_numpy(e.chirp_preamble()).cuda()
    n = 2048
    buf = torch.randn((n, 120000), device="cuda") * 0.1
    offs = rng.integers(0, 62000, n)
    idx = torch.from_numpy(offs).cuda()[:, None] + torch.arange(57600, device="cuda")[None, :]
    buf.scatter_add_(1, idx, ch[None, :].expand(n, -1))
    t = timed(lambda: e.sync_chirp(buf, 0.15))
    r = e.sync_chirp(buf, 0.15)
    exact = int((r["up_chirp_start"] == offs).sum())
    # the 131072-point transform pair of each of the two searches is 12 register-resident passes over a 1 MiB complex
    # array (read + write): 24 MiB of cache-side traffic per buffer (the 64-buffer chunks stay inside the Infinity Cache)
    fft_bytes = 12 * 2 * 131072 * 8
    res["chirp"] = {"buffers": n, "samples": 120000, "ms": round(t * 1e3, 2), "preambles_per_s": round(n / t), "success": int(r["success"].sum()),
                    "up_start_exact": exact, "GBps_algorithmic": round(n * 120000 * 4 / t / 1e9, 2),
                    "roofline": {"bound": "cache (Infinity Cache / L2) bandwidth of the FFT passes", "achieved": round(n * fft_bytes / t / 1e12, 2), "unit": "TB/s of FFT pass traffic over the whole call",
                                 "peak": 8.0, "peak_note": "HBM peak for scale; the passes of a 64-buffer chunk (160 MiB) are served on-die",
                                 "frac": round(n * fft_bytes / t / 1e12 / 8.0, 3)}}
    # LTS light sync on 21000-sample spans of noise (worst case: full search)
    n = 4000
    x = torch.randn((n, 21000), device="cuda") * 0.1
    t = timed(lambda: e.sync_lts(x, None, 0.5))
    # worst case (noise only): 1152 coarse offsets x 1152-term complex autocorrelation with both energies (14 flops per
    # term, 32 B of LDS reads per term) + the 65-tap Hilbert filter over the span
    lts_flop = 1152 * 1152 * 14 + 11520 * 65 * 2
    res["lts"] = {"buffers": n, "samples": 21000, "ms": round(t * 1e3, 2), "spans_per_s": round(n / t),
                  "roofline": {"bound": "lds", "achieved": round(n * 1152 * 1152 * 32 / t / 1e12, 2), "peak": round(256 * 256 * 2.4e9 / 1e12, 1), "unit": "TB/s of LDS reads (256 B/clk/CU)",
                               "frac": round(n * 1152 * 1152 * 32 / t / (256 * 256 * 2.4e9), 4), "TFLOPs": round(n * lts_flop / t / 1e12, 2)}}
    # Schmidl-Cox (OFDM-COX searchForSync): 8064-sample preamble at a random offset of 30000-sample noisy buffers
    pre = torch.from_numpy(e.cox_preamble()).cuda()
    n = 2048
    buf = torch.randn((n, 30000), device="cuda") * 0.02
    offs = rng.integers(0, 15000, n)
    idx = torch.from_numpy(offs).cuda()[:, None] + torch.arange(len(pre), device="cuda")[None, :]
    buf.scatter_add_(1, idx, pre[None, :].expand(n, -1))
    t = timed(lambda: e.sync_cox(buf, 0.8))
    r = e.sync_cox(buf, 0.8)
    exact = int((r["start_sample"] == offs + 5 * 1152).sum())
    res["cox"] = {"buffers": n, "samples": 30000, "ms": round(t * 1e3, 2), "buffers_per_s": round(n / t), "found": int(r["found"].sum()),
                  "first_lts_exact": exact, "metric_offsets_per_s": round(n * ((30000 - 6913) // 8 + 1) / t)}
    # MC-DPSK: C1 shape, 10 carriers DBPSK, one codeword (74 symbols)
    n = 8000
    x = torch.randn((n, 74 * 512), device="cuda") * 0.1
    t = timed(lambda: e.mcdpsk_demod(x, 10, 1, 1))
    res["mcdpsk"] = {"frames": n, "samples": 74 * 512, "ms": round(t * 1e3, 2), "frames_per_s": round(n / t),
                     "GBps_algorithmic": round(n * 74 * 512 * 4 / t / 1e9, 2)}
    # config C2: OFDM DQPSK R1/2, 10 000 frames, AWGN, FFT + LLR kernels only (ria_gpu_demod_batch)
    e2 = RxEngine("DQPSK", "R1_2", max_batch=10000)
    n = 10000
    info = e2.make_frames(7, 0, n)
    x = e2.tx(info, peak=0.8)
    e2.channel_(x, 0, 15.0, 7, first_frame=0)
    t = timed(lambda: e2.demod(x, want_status=False))
    fs = int(e2.geo.frame_samples)
    res["c2_dqpsk_demod"] = {"frames": n, "samples": fs, "ms": round(t * 1e3, 2), "frames_per_s": round(n / t),
                             "GBps_algorithmic": round(n * (fs * 4 + int(e2.geo.llrs_per_frame) * 4) / t / 1e9, 2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
