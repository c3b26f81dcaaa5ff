#!/usr/bin/env python3
"""Pin the C restatement (oracle/ria_oracle.c) against the compiled unmodified reference
(oracle/_ref/libria_ref.so).  Build-container only (needs /root/reference to have been compiled by
`make -C oracle ref`).  Everything is compared BIT-EXACT (float32 bit patterns, bytes, counters).

Usage: python oracle/check_against_ref.py [--frames N]
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyoracle as po  # noqa: E402


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def zc_test_buffer(pre, buf_len, off, snr_db, cfo_hz, rng):
    """Preamble at `off` in a noise-padded buffer; CFO by analytic-signal rotation of the preamble
    (SURVEY.md §8d C4); noise sigma from the preamble's rms."""
    n = len(pre)
    sig = np.zeros(buf_len, np.float64)
    seg = pre[:max(0, min(n, buf_len - off))].astype(np.float64)
    if cfo_hz != 0.0 and len(seg):
        spec = np.fft.fft(seg)
        h = np.zeros(len(seg)); h[0] = 1; h[1:(len(seg) + 1) // 2] = 2
        if len(seg) % 2 == 0: h[len(seg) // 2] = 1
        ana = np.fft.ifft(spec * h)
        seg = np.real(ana * np.exp(2j * np.pi * cfo_hz * (off + np.arange(len(seg))) / 48000.0))
    sig[off:off + len(seg)] = seg
    rms = np.sqrt(np.mean(pre[pre != 0].astype(np.float64) ** 2))
    sigma = rms * 10 ** (-snr_db / 20.0)
    return (sig + rng.normal(0, sigma, buf_len)).astype(np.float32)


def cox_cases():
    """(buffer length, preamble offset, snr dB, cfo Hz, threshold, initial noise floor)"""
    return [(30000, 0, 30, 0.0, 0.8, 0.0), (30000, 5000, 20, 12.5, 0.8, 0.0), (40000, 12000, 10, -30.0, 0.8, 0.0),
            (30000, 3000, 5, 40.0, 0.8, 0.0), (26000, 7777, 15, -8.0, 0.7, 1e-4), (50000, 30000, 25, 0.0, 0.8, 0.0),
            (20000, 15000, 20, 0.0, 0.8, 0.0), (9215, 0, 20, 0.0, 0.8, 0.0), (3999, 0, 20, 0.0, 0.8, 0.0),
            (30000, 4000, 0, 0.0, 0.8, 0.0), (30000, 4000, 30, 0.0, 0.95, 0.0), (30000, -1, 20, 0.0, 0.8, 0.0),
            (9300, 100, 25, 5.0, 0.8, 0.0), (16000, 500, 25, 5.0, 0.8, 0.0), (30000, 2000, 12, 20.0, 0.5, 0.0),
            (60000, 2000, 30, 0.0, 0.8, 0.0, 1), (60000, 6000, 22, 15.0, 0.8, 0.0, 1), (30000, 2000, 18, -20.0, 0.6, 0.0),
            (30000, 9000, 40, 45.0, 0.8, 0.0), (24000, 64, 25, 0.0, 0.8, 0.0), (30000, 2000, 28, 0.0, 0.8, 5e-3)]


def cox_test_buffer(R, buf_len, off, snr_db, cfo_hz, rng, variant=0):
    """Schmidl-Cox preamble + one modulated frame of random coded bytes at `off` (off < 0: noise only).
    variant 1: a preamble cut off after its STS part (Schmidl-Cox plateau without LTS confirmation: the search
    must carry on), silence, then a complete transmission."""
    coded = rng.integers(0, 256, 324, dtype=np.uint8)
    tx = R.cox_transmit(coded)
    if variant == 1:
        tx = np.concatenate([tx[:5 * 1152], np.zeros(12 * 1152, np.float32), tx])
    if off < 0:
        rms = np.sqrt(np.mean(tx[tx != 0].astype(np.float64) ** 2))
        return rng.normal(0, rms * 10 ** (-snr_db / 20.0), buf_len).astype(np.float32)
    return zc_test_buffer(tx, buf_len, off, snr_db, cfo_hz, rng)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    args = ap.parse_args()
    if not po.Ref.available():
        print("oracle/_ref/libria_ref.so missing: run `make -C oracle ref` in the build container")
        return 2
    O, R = po.Oracle(), po.Ref()
    rng = np.random.default_rng(2024)
    fails = 0

    def check(name, cond, extra=""):
        nonlocal fails
        print(("PASS " if cond else "FAIL ") + name + (" " + extra if extra else ""))
        if not cond:
            fails += 1

    # 1. LDPC encoder, every rate
    for rate in (po.R1_4, po.R1_3, po.R1_2, po.R2_3, po.R3_4, po.R5_6):
        k = O.code(rate).k
        ok = True
        for _ in range(8):
            info = rng.integers(0, 256, k // 8, dtype=np.uint8)
            ok &= np.array_equal(O.ldpc_encode(rate, info), R.ldpc_encode(rate, info)[:81])
        check(f"ldpc_encode rate={rate} (edges={O.code(rate).n_edges})", ok)

    # 2. channel interleaver permutations
    for bps in (188, 106, 110, 204, 255, 306, 94, 60, 20, 10):
        inv = R.channel_interleaver_inv(bps)
        step = O.lib.ro_channel_interleaver_step(bps, 648)
        mine = (np.arange(648) * step) % 648
        check(f"channel_interleaver bps={bps} step={step}", np.array_equal(inv, mine))

    # 3. TX chain + 4. channel + 5. RX + 7. decode, per mode
    modes = [(po.QAM16, po.R1_2), (po.DQPSK, po.R1_2), (po.QPSK, po.R1_2), (po.QAM64, po.R3_4),
             (po.QAM32, po.R3_4), (po.QAM16, po.R3_4), (po.DQPSK, po.R1_4), (po.BPSK, po.R1_2),
             (po.DBPSK, po.R1_4), (po.DQPSK, po.R3_4), (po.QAM16, po.R2_3)]
    for mod, rate in modes:
        g = O.geom(mod, rate)
        cap = 4 * g.bytes_per_cw - 19
        tx_ok = rx_ok = ch_ok = dec_ok = True
        n_dec = 0
        worst = ""
        for f in range(args.frames):
            payload = rng.integers(0, 256, cap, dtype=np.uint8)
            s_ref, info_ref, coded_ref, bps = R.tx_frame(mod, rate, payload, f)
            s_mine, info_mine, coded_mine = O.tx_frame(mod, rate, payload, f)
            tx_ok &= (bps == g.bits_per_symbol and np.array_equal(info_ref[:len(info_mine)], info_mine)
                      and np.array_equal(coded_ref, coded_mine) and bits_equal(s_ref, s_mine))
            x = s_ref * np.float32(0.8 / np.abs(s_ref).max())
            kind, snr = [(0, 20.0), (0, 12.0), (2, 20.0), (1, 15.0), (3, 20.0), (4, 25.0)][f % 6]
            cfo = [0.0, 0.0, 0.0, 2.5, -7.0, 0.4][f % 6]
            y_ref = R.channel(kind, snr, 1000 + f, x)
            y_mine = O.channel(kind, snr, 1000 + f, x)
            ch_ok &= bits_equal(y_ref, y_mine)
            abs_pos = [0, 0, 12345, 4800, 77, 0][f % 6]
            llr_ref, aux_ref, h_ref, _ = R.rx_process(mod, rate, y_ref, cfo, abs_pos)
            llr_mine, aux_mine, = O.rx_process(mod, rate, y_ref, cfo, abs_pos)
            aux_m = np.array([aux_mine.snr_db, aux_mine.cfo_hz, aux_mine.fading_index, aux_mine.noise_variance,
                              aux_mine.lts_phase_slope, aux_mine.snr_linear, aux_mine.corr_phase,
                              aux_mine.snr_symbol_count], np.float32)
            same = bits_equal(llr_ref, llr_mine) and bits_equal(h_ref, np.array(aux_mine.h, np.float32)) \
                and bits_equal(aux_ref[1:], aux_m[1:])
            if not same and not worst:
                nd = int(np.sum(llr_ref.view(np.uint32) != llr_mine.view(np.uint32))) if len(llr_ref) == len(llr_mine) else -1
                worst = f"[frame {f} kind={kind} cfo={cfo}: {nd} LLRs differ, aux ref={aux_ref} mine={aux_m}]"
            rx_ok &= same
            d_ref, ok_ref = R.decode_fixed_frame(llr_ref, rate, True, g.bits_per_symbol)
            d_mine, ok_mine, iters, att = O.decode_fixed_frame(llr_ref, rate, True, g.bits_per_symbol, flags=7)
            same = np.array_equal(ok_ref, ok_mine) and np.array_equal(d_ref[:len(d_mine)], d_mine)
            dec_ok &= same
            n_dec += int(ok_ref.all())
        check(f"tx        mod={mod} rate={rate}", tx_ok)
        check(f"channel   mod={mod} rate={rate}", ch_ok)
        check(f"rx llr    mod={mod} rate={rate}", rx_ok, worst)
        check(f"decode    mod={mod} rate={rate} ({n_dec}/{args.frames} frames fully decoded by ref)", dec_ok)

    # 6. raw LDPC decode incl. iteration counts, noisy codewords around the waterfall
    for rate in (po.R1_4, po.R1_2, po.R2_3, po.R3_4, po.R5_6):
        c = O.code(rate)
        ok = True
        n_succ = 0
        for t in range(40):
            info = rng.integers(0, 256, (c.k + 7) // 8, dtype=np.uint8)
            cw = np.unpackbits(O.ldpc_encode(rate, info))[:648].astype(np.float32)
            sigma = [0.5, 0.7, 0.8, 0.9, 1.0][t % 5] * (1.3 if rate == po.R1_4 else 1.0)
            llr = (2.0 * (1.0 - 2.0 * cw) + rng.normal(0, 2.0 * sigma, 648)).astype(np.float32) * np.float32(2.0 / (2 * sigma) ** 2)
            for factor, mi in ((0.9375, 80), (0.75, 50)):
                ok_r, out_r, it_r = R.ldpc_decode(rate, llr, mi, factor)
                ok_m, out_m, it_m = O.ldpc_decode(rate, llr, mi, factor)
                ok &= (ok_r == ok_m and it_r == it_m and np.array_equal(out_r, out_m))
                n_succ += ok_r
        check(f"ldpc_decode rate={rate} ({n_succ}/80 converged)", ok)

    # 8. ZC acquisition (sync::ZCSync): preamble synthesis and detect() over SNR x CFO x offset x root
    for root in (1, 3, 5, 7):
        check(f"zc_generate root={root}", bits_equal(O.zc_generate(root), R.zc_generate(root)))
    for buf_len, label in ((4512, "C4 buffer"), (3000, "short buffer"), (1500, "one repetition")):
        ok = True
        n_det = 0
        worst = ""
        for t in range(24):
            root = (1, 3, 5, 7)[t % 4]
            snr_db = (-10, -5, 0, 5, 10, 25)[t % 6]
            cfo = (-20.0, -10.0, 0.0, 10.0, 20.0)[t % 5]
            off = int(rng.integers(0, max(1, buf_len - 2512))) if buf_len > 2512 else 0
            x = zc_test_buffer(R.zc_generate(root), buf_len, off, snr_db, cfo, rng)
            for mask, known in ((15, 0.0), (12, 0.0), (15, cfo)):
                a, b = O.zc_detect(x, 0.3, mask, known), R.zc_detect(x, 0.3, mask, known)
                same = bits_equal(a, b)
                ok &= same
                if not same and not worst:
                    worst = f"t={t} mask={mask} known={known} oracle={a} ref={b}"
                n_det += int(b[0])
        check(f"zc_detect {label} ({n_det}/72 detected by ref)", ok, worst)

    # 9. dual-chirp acquisition (sync::ChirpSync::detectDualChirp): FFT-131072 path and the time-domain fallback
    check("chirp_generate", bits_equal(O.chirp_generate(), R.chirp_generate()))
    chirp = R.chirp_generate()
    ok = True
    n_det = 0
    worst = ""
    cases = [(120000, 20000, 5, 0.0), (120000, 1000, 0, 25.0), (120000, 61000, -5, -50.0), (120000, 40000, -10, 50.0),
             (120000, 30000, 10, -25.0), (70000, 5000, 5, 10.0), (60000, 1200, 0, 0.0), (57600, 0, 15, 0.0),
             (100000, 50000, 5, 0.0), (131072 + 5000, 60000, 0, 30.0)]
    for buf_len, off, snr_db, cfo in cases:
        x = zc_test_buffer(chirp, buf_len, off, snr_db, cfo, rng)
        a, b = O.chirp_detect(x, 0.15), R.chirp_detect(x, 0.15)
        same = bits_equal(a, b)
        ok &= same
        if not same and not worst:
            worst = f"len={buf_len} off={off} snr={snr_db} cfo={cfo} oracle={a} ref={b}"
        n_det += int(b[0])
    check(f"chirp_detect ({n_det}/{len(cases)} detected by ref)", ok, worst)

    # Schmidl-Cox acquisition: searchForSync (LTS template, metric, plateau, LTS refinement, coarse CFO, noise-floor state)
    for mod, rate in ((po.QAM16, po.R1_2), (po.DQPSK, po.R1_4)):
        a, b = O.cox_lts_template(mod, rate), R.cox_lts_template(mod, rate)
        check(f"cox LTS passband template mod={mod} rate={rate}", bits_equal(a[0], b[0]) and bits_equal(a[1], b[1]))
    ok, n_det, worst = True, 0, ""
    for ci, (buf_len, off, snr_db, cfo, thr, nf0, *var) in enumerate(cox_cases()):
        x = cox_test_buffer(R, buf_len, off, snr_db, cfo, rng, var[0] if var else 0)
        (a, na), (b, nb) = O.cox_search(x, thr, nf0), R.cox_search(x, thr, nf0)
        same = bits_equal(a, b) and bits_equal([na], [nb])
        ok &= same
        if not same and not worst:
            worst = f"len={buf_len} off={off} snr={snr_db} cfo={cfo} oracle={a},{na} ref={b},{nb}"
        n_det += int(b[0])
    check(f"cox_search ({n_det}/{len(cox_cases())} found by ref)", ok, worst)

    # config 4 impairments: SimulatedChannel::applyTxCFO (tools/cli_simulator.cpp:298-341, compiled into the reference
    # library from the tool's own translation unit) and WattersonChannel's cfo_hz / random_cfo_max_hz + applyCFO
    ok, worst = True, ""
    zc5 = R.zc_generate(5)
    for name, x in (("zc", zc5), ("chirp", chirp), ("n=1", zc5[100:101]), ("n=2", zc5[100:102]), ("n=63", zc5[:63]),
                    ("noise300", rng.standard_normal(300).astype(np.float32)), ("noise5000", rng.standard_normal(5000).astype(np.float32))):
        for cfo in (-50.0, -25.0, 0.0, 25.0, 50.0, 13.7, 0.0005):
            for ph in (0.0, 1.25, -3.0):
                (a, pa), (b, pb) = O.apply_tx_cfo(x, cfo, ph), R.apply_tx_cfo(x, cfo, ph)
                same = bits_equal(a, b) and bits_equal([pa], [pb])
                ok &= same
                if not same and not worst:
                    worst = f"{name} cfo={cfo} phase={ph}"
    check("applyTxCFO (7 inputs x 7 offsets x 3 accumulators)", ok, worst)
    ok, worst = True, ""
    s0, _, _ = O.tx_frame(po.QAM16, po.R1_2, rng.integers(0, 256, 141, dtype=np.uint8), 1)
    frame = s0 * np.float32(0.8 / np.abs(s0).max())
    acq = np.zeros(120000, np.float32); acq[30000:30000 + len(chirp)] = chirp
    for name, x, snr in (("frame", frame, 15.0), ("chirp buffer", acq, 0.0), ("255 samples", frame[:255], 10.0)):
        for kind in range(5):
            for cfo, rmax in ((0.0, 0.0), (25.0, 0.0), (-50.0, 0.0), (10.5, 0.0), (0.0, 30.0), (5.0, 50.0), (0.0005, 0.0)):
                (a, ca), (b, cb) = O.channel_cfo(kind, snr, 77 + kind, x, cfo, rmax), R.channel_cfo(kind, snr, 77 + kind, x, cfo, rmax)
                same = bits_equal(a, b) and bits_equal([ca], [cb])
                ok &= same
                if not same and not worst:
                    worst = f"{name} kind={kind} cfo={cfo} random_max={rmax}"
    check("WattersonChannel cfo_hz / random_cfo_max_hz + applyCFO (3 inputs x 5 presets x 7 settings)", ok, worst)

    # the scenarios of the reference's own test programs (tools/test_zc_sync.cpp, test_spreading.cpp, test_chase_cache.cpp):
    # restatement vs the builders that call the reference (oracle/ref_shim_tools.cpp); gen_golden.py additionally runs the
    # programs themselves (make -C oracle tools) and compares what they print with those builders
    zo, zr = O.tool_zc_cases(), R.tool_zc_cases()
    check("tools/test_zc_sync.cpp scenarios (50 signals + ZCSyncResult)", all(
        (bits_equal(zo[k], zr[k]) if zo[k].dtype == np.float32 else np.array_equal(zo[k], zr[k])) for k in zo))
    ok, worst = True, ""
    for m in (0, 2, 4):
        for snr in (-16.0, -14.0, -12.0, -10.0, -8.0, -6.0, -4.0, -2.0, 0.0):
            for t in range(20):
                a, b = O.tool_spreading_case(snr, m, 1000 + t), R.tool_spreading_case(snr, m, 1000 + t)
                same = (np.array_equal(a["tx"], b["tx"]) and bits_equal(a["frame"], b["frame"]) and bits_equal(a["soft"], b["soft"])
                        and a["ok"] == b["ok"] and np.array_equal(a["decoded"], b["decoded"]) and a["bit_errors"] == b["bit_errors"])
                ok &= same
                if not same and not worst:
                    worst = f"spreading={m} snr={snr} seed={1000 + t}"
    check("tools/test_spreading.cpp testAtSNR (3 modes x 9 SNRs x 20 trials: frame, soft bits, decode)", ok, worst)
    (lo, oo), (lr, orr) = O.tool_chase_llrs(), R.tool_chase_llrs()
    check("tools/test_chase_cache.cpp receptions and decodes (400 codewords, 350 decodes)", bits_equal(lo, lr) and np.array_equal(oo, orr))

    import gen_golden
    ok, worst = True, ""
    for snr, seed in gen_golden.zc_dbpsk_cases():
        a, b = O.tool_zc_dbpsk_case(snr, seed), R.tool_zc_dbpsk_case(snr, seed)
        same = (np.array_equal(a["tx"], b["tx"]) and bits_equal(a["signal"], b["signal"]) and bits_equal(a["zc7"], b["zc7"]) and a["stage"] == b["stage"]
                and bits_equal(a["soft"], b["soft"]) and np.array_equal(a["decoded"], b["decoded"]) and a["ok"] == b["ok"] and a["bit_errors"] == b["bit_errors"])
        ok &= same
        if not same and not worst:
            worst = f"snr={snr} seed={seed}"
    check("tools/test_zc_dbpsk.cpp testAtSNR (130 cases: signal, ZC result, stage reached, soft bits, decode)", ok, worst)

    print("\n%s: %d failing group(s)" % ("PINNED" if fails == 0 else "MISMATCH", fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
