#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the compiled unmodified reference (oracle/_ref/libria_ref.so).

Build-container only.  The fixtures are DATA (inputs and the reference's outputs at each stage tap);
no reference source travels.  Re-run after `make -C oracle ref`:

    python oracle/gen_golden.py

Stage taps per frame: payload -> frame info bytes -> interleaved coded bytes
-> TX samples -> channel output -> LLRs (+ estimator state: H[59], noise variance, phase slope,
CFO, fading index) -> decodeFixedFrame status and bytes.
"""
import os
import re
import subprocess
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyoracle as po  # noqa: E402

OUT = os.path.join(os.path.dirname(po.HERE), "tests", "golden")

# (name, mod, rate, [(channel kind, snr_db, cfo_hz, abs_pos)], keep_tx)
FRAME_SETS = [
    ("qam16_r12", po.QAM16, po.R1_2,
     [(0, 20.0, 0.0, 0), (0, 12.0, 0.0, 0), (2, 20.0, 0.0, 0), (2, 20.0, 0.0, 0), (2, 15.0, 0.0, 0),
      (1, 15.0, 2.5, 4800), (3, 20.0, -7.0, 12345), (0, 25.0, 0.4, 0), (2, 25.0, 0.0, 0), (4, 25.0, 0.0, 0)]),
    ("dqpsk_r12", po.DQPSK, po.R1_2,
     [(0, 20.0, 0.0, 0), (0, 10.0, 0.0, 0), (2, 20.0, 0.0, 0), (1, 15.0, 3.0, 960)]),
    ("qam64_r34", po.QAM64, po.R3_4, [(0, 30.0, 0.0, 0), (0, 24.0, 0.0, 0), (1, 30.0, 0.0, 0)]),
    ("qam32_r34", po.QAM32, po.R3_4, [(0, 28.0, 0.0, 0), (1, 25.0, 0.0, 0)]),
    ("qpsk_r12", po.QPSK, po.R1_2, [(0, 12.0, 0.0, 0), (2, 15.0, 0.0, 0)]),
    ("dqpsk_r14", po.DQPSK, po.R1_4, [(0, 6.0, 0.0, 0), (3, 10.0, 0.0, 0)]),
    ("qam16_r34", po.QAM16, po.R3_4, [(0, 22.0, 0.0, 0), (2, 25.0, 0.0, 0)]),
    # added later: own RNG stream (5th element) so that the sets above keep their recorded draws
    ("d8psk_r12", po.D8PSK, po.R1_2, [(0, 25.0, 0.0, 0), (2, 25.0, 0.0, 0), (1, 20.0, 1.5, 2400), (3, 24.0, 0.0, 0)], 8801),
    ("d8psk_r14", po.D8PSK, po.R1_4, [(0, 18.0, 0.0, 0), (2, 22.0, 0.0, 0)], 8802),
    # QAM256 only exists behind the OFDM-COX waveform object (OFDMChirpWaveform::configure maps it to DQPSK,
    # ofdm_chirp_waveform.cpp:82-89): 6th element = record through OFDMNvisWaveform (same modulator / demodulator classes)
    ("qam256_r34", po.QAM256, po.R3_4, [(0, 38.0, 0.0, 0), (0, 31.0, 0.0, 0), (1, 36.0, 1.0, 4800), (2, 40.0, 0.0, 0)], 8803, True),
    ("qam256_r12", po.QAM256, po.R1_2, [(0, 30.0, 0.0, 0), (1, 30.0, 0.0, 0), (0, 24.0, -2.0, 777)], 8804, True),
    # R1/3: rate-table entry with the (324,324) default code parameters (ldpc_decoder.cpp:21-36) and 27-byte codewords
    ("qam16_r13", po.QAM16, po.R1_3, [(0, 16.0, 0.0, 0), (2, 18.0, 0.0, 0), (1, 13.0, 0.0, 0)], 8805),
    ("dqpsk_r13", po.DQPSK, po.R1_3, [(0, 7.0, 0.0, 0), (3, 11.0, 0.0, 0)], 8806),
]


CHIRP_CASES = [(120000, 20000, 5, 0.0), (120000, 1000, 0, 25.0), (120000, 61000, -5, -50.0), (120000, 40000, -10, 50.0),
               (120000, 30000, 10, -25.0), (120000, 62400, 0, 0.0), (70000, 5000, 5, 10.0), (60000, 1200, 0, 0.0),
               (57600, 0, 15, 0.0), (100000, 50000, 5, 0.0), (136072, 60000, 0, 30.0), (52000, 0, 10, 0.0),
               (120000, 15000, -15, 0.0), (120000, 33333, 3, 75.0), (120000, 44444, 3, -75.0), (120000, 10000, 20, 120.0)]


def chirp_buffer(chirp, case, idx):
    import zlib
    from check_against_ref import zc_test_buffer
    buf_len, off, snr_db, cfo = case
    x = zc_test_buffer(chirp, buf_len, off, snr_db, cfo, np.random.default_rng(9000 + idx))
    return x, zlib.crc32(x.tobytes())


# (buffer length, preamble offset (<0: noise only), snr dB, cfo Hz, threshold, initial noise floor, variant)
COX_CASES = [(30000, 0, 30, 0.0, 0.8, 0.0, 0), (30000, 5000, 20, 12.5, 0.8, 0.0, 0), (40000, 12000, 10, -30.0, 0.8, 0.0, 0),
             (26000, 7777, 15, -8.0, 0.7, 1e-4, 0), (50000, 30000, 25, 0.0, 0.8, 0.0, 0), (20000, 15000, 20, 0.0, 0.8, 0.0, 0),
             (9215, 0, 20, 0.0, 0.8, 0.0, 0), (3999, 0, 20, 0.0, 0.8, 0.0, 0), (30000, 4000, 30, 0.0, 0.95, 0.0, 0),
             (30000, -1, 20, 0.0, 0.8, 0.0, 0), (9300, 100, 25, 5.0, 0.8, 0.0, 0), (16000, 500, 25, 5.0, 0.8, 0.0, 0),
             (60000, 2000, 30, 0.0, 0.8, 0.0, 1), (60000, 6000, 22, 15.0, 0.8, 0.0, 1), (30000, 2000, 18, -20.0, 0.6, 0.0, 0),
             (30000, 9000, 40, 45.0, 0.8, 0.0, 0), (24000, 64, 25, 0.0, 0.8, 0.0, 0), (30000, 2000, 28, 0.0, 0.8, 5e-3, 0),
             (9216, 0, 30, 0.0, 0.8, 0.0, 0), (9217, 0, 30, 0.0, 0.8, 0.0, 0), (12000, 1000, 30, 0.0, 0.8, 0.0, 0),
             (40000, 500, 30, 0.0, 0.8, 0.0, 2), (40000, 3000, 24, -10.0, 0.8, 0.0, 2)]


def cox_buffer(tx, case, idx):
    """tx = Schmidl-Cox preamble + one modulated frame (float32).  variant 1: the STS part alone (a plateau the
    LTS refinement has to judge), silence, then the complete transmission.  variant 2: an out-of-band 10 kHz tone
    burst first (Schmidl-Cox metric ~1 over a long plateau, LTS confirmation fails: the search must carry on)."""
    import zlib
    from check_against_ref import zc_test_buffer
    buf_len, off, snr_db, cfo = int(case[0]), int(case[1]), float(case[2]), float(case[3])
    rng = np.random.default_rng(9500 + idx)
    if int(case[6]) == 1:
        tx = np.concatenate([tx[:5 * 1152], np.zeros(12 * 1152, np.float32), tx])
    if int(case[6]) == 2:
        tone = (0.3 * np.sin(2 * np.pi * 10000.0 * np.arange(9000) / 48000.0)).astype(np.float32)
        tx = np.concatenate([tone, np.zeros(3 * 1152, np.float32), tx])
    if off < 0:
        rms = np.sqrt(np.mean(tx[tx != 0].astype(np.float64) ** 2))
        x = rng.normal(0, rms * 10 ** (-snr_db / 20.0), buf_len).astype(np.float32)
    else:
        x = zc_test_buffer(tx, buf_len, off, snr_db, cfo, rng)
    return x, zlib.crc32(x.tobytes())


def cox_fixture(R):
    """OFDM-COX: the reference's preamble + frame audio, LTS passband templates, and searchForSync results
    (found, first-LTS position, cfo, noise floor after) for buffers rebuilt from the recipe."""
    rng = np.random.default_rng(515)
    rec = {"cases": np.array(COX_CASES, np.float32)}
    for name, mod, rate in (("qam16_r12", po.QAM16, po.R1_2), ("dqpsk_r14", po.DQPSK, po.R1_4)):
        tI, tQ = R.cox_lts_template(mod, rate)
        rec[f"tI_{name}"], rec[f"tQ_{name}"] = tI, tQ
        rec[f"preamble_{name}"] = R.cox_transmit(None, mod, rate)
    payload = rng.integers(0, 256, 141, dtype=np.uint8)
    _, info, coded, bps = R.tx_frame(po.QAM16, po.R1_2, payload, 77)
    tx = R.cox_transmit(coded)
    rec["tx"], rec["info"] = tx, info[:160]
    crcs, res, e2e_idx, e2e_llr, e2e_dec = [], [], [], [], []
    for i, case in enumerate(COX_CASES):
        x, crc = cox_buffer(tx, case, i)
        out, nf = R.cox_search(x, float(case[4]), float(case[5]))
        crcs.append(crc)
        res.append(np.concatenate([out, [nf]]).astype(np.float32))
        # end to end (detectSync -> process -> getSoftBits -> decodeFixedFrame) where a whole frame follows the sync point
        if out[0] and int(out[1]) + 18432 <= len(x) and float(case[5]) == 0.0:
            o3, llr, _ = R.cox_rx(x, float(case[4]))
            assert np.array_equal(o3, out) and len(llr) >= 2632
            d, ok = R.decode_fixed_frame(llr[:2632], po.R1_2, True, bps)
            e2e_idx.append(i); e2e_llr.append(llr[:2632]); e2e_dec.append(np.concatenate([ok, d[:160]]))
    rec["e2e_case"] = np.array(e2e_idx, np.int32)
    rec["e2e_llr"] = np.stack(e2e_llr)
    rec["e2e_dec"] = np.stack(e2e_dec)      # [4 ok flags | 160 info bytes]
    rec["buffer_crc"] = np.array(crcs, np.uint32)
    rec["results"] = np.stack(res)
    return rec


MCDPSK_CASES = [(10, 1, 1, 20.0, 0.0, 0.0), (10, 1, 1, 0.0, 0.0, 0.0), (10, 2, 1, 8.0, 0.0, 0.0), (10, 1, 2, -3.0, 0.0, 0.0),
                (10, 1, 4, -6.0, 0.0, 0.0), (8, 2, 1, 15.0, 6.5, 0.4), (10, 1, 1, 5.0, -12.0, -1.1), (5, 2, 2, 6.0, 0.0, 0.0)]


def mcdpsk_frame(R_or_O, case, idx):
    """Noisy (and optionally CFO-shifted) training+ref+data frame of one 81-byte codeword, float32."""
    nc, bps, sp, snr_db, cfo, ph0 = case
    rng = np.random.default_rng(7000 + idx)
    data = rng.integers(0, 256, 81, dtype=np.uint8)
    tx = R_or_O.mcdpsk_modulate(nc, bps, sp, data)
    x = tx.astype(np.float64) + rng.normal(0, np.sqrt(np.mean(tx.astype(np.float64) ** 2)) * 10 ** (-snr_db / 20.0), len(tx))
    if cfo:
        from scipy.signal import hilbert
        x = np.real(hilbert(x) * np.exp(2j * np.pi * cfo * np.arange(len(x)) / 48000.0))
    if idx == 3:
        x[-6 * 512:] *= 0.02      # trailing near-silence: exercises the valid_symbols trimming
    return data, tx, x.astype(np.float32)


def mcdpsk_fixture(R):
    rec = {"cases": np.array(MCDPSK_CASES, np.float32)}
    for i, case in enumerate(MCDPSK_CASES):
        data, tx, x = mcdpsk_frame(R, case, i)
        llr, aux = R.mcdpsk_demod(case[0], case[1], case[2], x, float(case[4]), float(case[5]))
        rec[f"data_{i}"] = data; rec[f"tx_crc_{i}"] = np.array([__import__("zlib").crc32(tx.tobytes())], np.uint32); rec[f"rx_{i}"] = x; rec[f"llr_{i}"] = llr; rec[f"aux_{i}"] = aux
    return rec


def lts_buffers(O, n_cases, seed):
    """Connected-mode frames (2 LTS + data) behind silence / inside a burst, noise, CFO, negated first LTS."""
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_cases):
        s, info, coded = O.tx_frame(po.QAM16, po.R1_2, rng.integers(0, 256, 141, dtype=np.uint8), t)
        s = s * np.float32(0.5 / np.abs(s).max())
        lead = int(rng.integers(0, 6000)) if t % 4 else 0
        sigma = [0.002, 0.02, 0.08, 0.2][t % 4]
        x = np.concatenate([np.zeros(lead, np.float32), s, np.zeros(1500, np.float32)])
        if t % 5 == 4:
            x[lead:lead + 1152] *= -1
        if t % 7 == 6:
            x = x[lead + 3000:]
        x = (x + rng.normal(0, sigma, len(x))).astype(np.float32)
        out.append((x[:21000] if len(x) >= 21000 else np.pad(x, (0, 21000 - len(x))), [0.0, 3.0, -7.5][t % 3]))
    return out


def lts_fixture(R, O):
    bufs = lts_buffers(O, 12, 606)
    res = [R.detect_data_sync(x, cfo, 0.5) for x, cfo in bufs]
    return {"buffers": np.stack([x for x, _ in bufs]), "cfo": np.array([c for _, c in bufs], np.float32), "results": np.stack(res)}


# (mod, rate, burst frames N, lead samples, channel kind, snr dB, cfo Hz (applied from sample 0, known to the receiver),
#  abs_base, marker transmitted)
BURST_CASES = [(po.QAM16, po.R1_2, 3, 1777, 0, 22.0, 0.0, 0, 1), (po.QAM16, po.R1_2, 4, 0, 2, 20.0, 0.0, 0, 1),
               (po.QAM16, po.R1_2, 2, 2600, 0, 18.0, 3.0, 4800, 1), (po.DQPSK, po.R1_2, 4, 900, 1, 15.0, 0.0, 0, 1),
               (po.QAM16, po.R1_2, 3, 1234, 0, 22.0, 0.0, 0, 0), (po.QAM64, po.R3_4, 8, 500, 0, 28.0, -2.0, 96000, 1),
               (po.QAM16, po.R1_2, 4, 3100, 2, 14.0, 0.0, 0, 1)]
BURST_BPC = {po.R1_4: 20, po.R1_2: 40, po.R2_3: 54, po.R3_4: 60, po.R5_6: 67}


def burst_buffer(O, case, idx):
    """One burst-interleaved group as the air interface carries it, rebuilt from the recipe with the oracle's TX pieces
    (each pinned bit-for-bit to the reference): N serialized frames -> encodeFixedFrame -> BurstInterleaver::interleave
    -> [2 LTS][data] per frame, first LTS of the group negated (streaming_encoder.cpp:302-389); scaled to peak 0.5,
    `lead` samples of silence in front, 3000 behind, optional CFO (analytic rotation from sample 0), channel over the
    whole capture.  -> (rx buffer float32, info bytes [N, 4*bpc], crc32 of the buffer)"""
    import zlib
    mod, rate, n, lead, kind, snr, cfo, abs_base, marker = case
    rng = np.random.default_rng(31000 + idx)
    bpc = BURST_BPC[rate]
    infos = np.stack([O.make_frame(rng.integers(0, 256, 4 * bpc - 19, dtype=np.uint8), 500 + 10 * idx + f, rate) for f in range(n)])
    g = O.geom(mod, rate)
    coded = np.stack([O.encode_fixed_frame(infos[f], rate, True, g.bits_per_symbol) for f in range(n)])
    phys = O.burst_interleave(coded) if n >= 2 else coded
    frames = []
    for f in range(n):
        a = O.modulate(mod, rate, phys[f])
        if f == 0 and marker:
            a[:1152] = -a[:1152]
        frames.append(a)
    s = np.concatenate(frames)
    s = s * np.float32(0.5 / np.abs(s).max())
    x = np.concatenate([np.zeros(lead, np.float32), s, np.zeros(3000, np.float32)])
    if cfo:
        n_all = len(x)
        spec = np.fft.fft(x.astype(np.float64))
        h = np.zeros(n_all); h[0] = 1; h[1:(n_all + 1) // 2] = 2
        if n_all % 2 == 0:
            h[n_all // 2] = 1
        x = np.real(np.fft.ifft(spec * h) * np.exp(2j * np.pi * cfo * (abs_base + np.arange(n_all)) / 48000.0)).astype(np.float32)
    x = O.channel(kind, snr, 7100 + idx, x)
    return x, infos, zlib.crc32(x.tobytes())


def burst_fixture(R, O):
    """8f-3 + a9: burst groups through ONE reference waveform object in StreamingDecoder's call order
    (oracle/ref_shim.cpp ref_burst_rx), and the reference's own TX of the first case (pins the recipe's TX pieces)."""
    rec = {"cases": np.array(BURST_CASES, np.float64)}
    for i, case in enumerate(BURST_CASES):
        mod, rate, n, lead, kind, snr, cfo, abs_base, marker = case
        x, infos, crc = burst_buffer(O, case, i)
        s_ref, cl, cp = R.burst_tx(mod, rate, infos, bool(marker))
        coded = np.stack([O.encode_fixed_frame(infos[f], rate, True, O.geom(mod, rate).bits_per_symbol) for f in range(n)])
        assert np.array_equal(cl, coded) and np.array_equal(cp, O.burst_interleave(coded) if n >= 2 else coded)
        frames = np.concatenate([O.modulate(mod, rate, cp[f]) for f in range(n)])
        if marker:
            frames[:1152] = -frames[:1152]
        assert np.array_equal(frames.view(np.uint32), s_ref.view(np.uint32)), "oracle TX pieces != reference burst TX"
        r = R.burst_rx(mod, rate, x, n, known_cfo=float(cfo), abs_base=int(abs_base), bpc=BURST_BPC[rate])
        assert r["n_soft"] > 0, (i, r["n_soft"])
        rec[f"crc_{i}"] = np.array([crc], np.uint32)
        rec[f"infos_{i}"] = infos
        for k in ("sync", "llr", "cfo_used", "cfo_after", "logical", "dec_data", "dec_ok"):
            rec[f"{k}_{i}"] = r[k]
        print("burst case", i, "sync", r["sync"], "cfo_after", r["cfo_after"], "frames decoded", r["dec_ok"].all(axis=1).astype(int),
              "== tx", [(r["dec_data"][f] == infos[f]).all() for f in range(n)])
    return rec


def burst_interleaver_fixture(R):
    """fec::BurstInterleaver permutations recorded from the reference: random coded bytes in, physical bytes out, and
    the soft-bit de-interleave as an index map (physical flat index of every logical position)."""
    rng = np.random.default_rng(808)
    rec = {}
    for n in (1, 2, 3, 4, 7, 8):
        lb = rng.integers(0, 256, (n, 324), dtype=np.uint8)
        rec[f"logical_bytes_{n}"] = lb
        rec[f"physical_bytes_{n}"] = R.burst_interleave(lb)
        probe = np.arange(n * 2592, dtype=np.float32).reshape(n, 2592)
        rec[f"deint_index_{n}"] = R.burst_deinterleave(probe).astype(np.int32)
    return rec


def robust_fixture(R):
    """robustDecodeSingleCW (streaming_decoder.cpp:1028-1058) on noisy codewords at the edge of convergence, so that
    every number of tries 1..5 and the all-fail outcome occur: ok, tries, iterations, bytes recorded from the reference."""
    rng = np.random.default_rng(60606)
    rec = {}
    # the remaining rates were added later with a generator of their own (60607), so that the first three keep their recorded draws
    for rate, sig in ((po.R1_4, (1.3, 1.4, 1.5)), (po.R1_2, (0.8, 0.84, 0.88)), (po.R3_4, (0.5, 0.53, 0.56)),
                      (po.R1_3, (0.8, 0.84, 0.88)), (po.R2_3, (0.6, 0.64, 0.68)), (po.R5_6, (0.47, 0.51, 0.55))):
        if rate == po.R1_3:
            rng = np.random.default_rng(60607)
        k = {po.R1_4: 162, po.R1_3: 324, po.R1_2: 324, po.R2_3: 432, po.R3_4: 486, po.R5_6: 540}[rate]
        llrs, res = [], []
        for t in range(3000):
            info = rng.integers(0, 256, (k + 7) // 8, dtype=np.uint8)
            if k % 8:
                info[-1] &= (0xFF << (8 - k % 8)) & 0xFF
            bits = np.unpackbits(R.ldpc_encode(rate, info)[:81])[:648].astype(np.float32)
            sigma = sig[t % 3]
            llr = np.clip(((1.0 - 2.0 * bits) + rng.normal(0, sigma, 648)) * (2.0 / sigma ** 2), -20, 20).astype(np.float32)
            ok, out, it, tries = R.robust_decode(rate, llr)
            llrs.append(llr)
            res.append(np.concatenate([[int(ok), tries, it], out[:(k + 7) // 8]]).astype(np.int32))
        res = np.stack(res)
        # keep the interesting ones: every retry outcome, plus a few plain successes and failures
        keep = []
        for tries in (1, 2, 3, 4, 5):
            keep += [i for i in range(len(res)) if res[i, 1] == tries and res[i, 0] == 1][:8]
        keep += [i for i in range(len(res)) if res[i, 0] == 0][:8]
        rec[f"llr_{rate}"] = np.stack(llrs)[keep]
        rec[f"res_{rate}"] = res[keep]
        print("robust rate", rate, "tries histogram", np.bincount(res[keep][:, 1], minlength=6), "ok", int(res[keep][:, 0].sum()), "of", len(keep))
    return rec


# (carriers, bits per symbol, spreading, channel kind, snr dB): the MC-DPSK rungs of the ladder (waveform_selection.hpp:112-222)
# at two marginal SNRs each (sigma from the rms of the non-zero samples, as sim::WattersonChannel defines it), one faded
HARQ_CASES = [(10, 1, 4, 0, -22.0), (10, 1, 4, 0, -20.0), (10, 1, 2, 0, -19.0), (10, 1, 2, 0, -17.0), (10, 1, 1, 0, -16.0),
              (10, 1, 1, 0, -14.0), (10, 2, 1, 0, -12.0), (10, 2, 1, 0, -10.0), (10, 1, 1, 1, -9.0), (10, 2, 1, 2, -3.0)]
HARQ_TRIALS, HARQ_SEED = 192, 424242


def harq_inputs(case_index, n=HARQ_TRIALS):
    """(info [n, 21], seeds [n, 4]) of a HARQ case: the sweep's own per-trial recipe (ria_amd/sweep.py)"""
    sys.path.insert(0, os.path.dirname(po.HERE))
    from ria_amd import sweep
    trials = np.arange(n)
    info = sweep.trial_payloads(HARQ_SEED, case_index, trials, 16)
    seeds = np.stack([sweep.trial_seed32(HARQ_SEED, case_index, t, trials) for t in range(4)], axis=1)
    return info, seeds


def harq_fixture(R):
    """BASELINE config 5, MC-DPSK rungs: per trial the reference's modulator -> WattersonChannel(seed) -> demodulator ->
    robustDecodeSingleCW / ChaseCache / robustDecodeSingleCW of the sum, up to 4 transmissions (ref_shim.cpp ref_harq_trials).
    Recorded: transmissions to success, checksum of every reception's soft bits and of every cache sum, decoder tries,
    decoded bytes, the demodulator's fading index."""
    rec = {"cases": np.array(HARQ_CASES, np.float64), "trials": np.int32(HARQ_TRIALS), "seed": np.int64(HARQ_SEED)}
    for i, (nc, bps, sp, kind, snr) in enumerate(HARQ_CASES):
        info, seeds = harq_inputs(i)
        r = R.harq_trials(nc, bps, sp, kind, snr, info, seeds)
        for k in ("tx_to_success", "llr_crc", "acc_crc", "tries", "decoded", "fading"):
            rec[f"{k}_{i}"] = r[k]
        good = (r["tx_to_success"] > 0) & (r["decoded"] == info[:, :20]).all(axis=1)
        print("harq case", i, HARQ_CASES[i], "tx-to-success histogram", np.bincount(r["tx_to_success"], minlength=5), "correct", int(good.sum()))
    return rec


def chirp_fixture(R):
    chirp = R.chirp_generate()
    rec = {"preamble_crc": np.array([__import__("zlib").crc32(chirp.tobytes())], np.uint32), "cases": np.array(CHIRP_CASES, np.float32)}
    crcs, res = [], []
    for i, case in enumerate(CHIRP_CASES):
        x, crc = chirp_buffer(chirp, case, i)
        crcs.append(crc)
        res.append(R.chirp_detect(x, 0.15))
    rec["buffer_crc"] = np.array(crcs, np.uint32)
    rec["results"] = np.stack(res)
    return rec


# ---- config 4: the two CFO impairments and the acquisition grid (recorded from the reference)
TXCFO_CASES = [(2512, -50.0, 0.0), (2512, 25.0, 1.25), (2512, 0.0005, 0.5), (300, 13.7, -3.0), (4096, -25.0, 0.0), (1, 10.0, 0.25),
               (2, -10.0, 0.0), (63, 50.0, 2.0), (5000, 50.0, -1.0)]
CHANCFO_CASES = [(0, 25.0, 0.0), (0, -50.0, 0.0), (1, 10.5, 0.0), (2, 50.0, 0.0), (3, -25.0, 0.0), (4, 0.0, 30.0), (0, 5.0, 50.0),
                 (2, 0.0005, 0.0), (0, 0.0, 0.0)]          # (preset, cfo_hz, random_cfo_max_hz)
ACQ_GRID = [(c, s) for c in (-50.0, -25.0, 0.0, 25.0, 50.0) for s in (-10.0, -5.0, 0.0, 5.0, 10.0)]
ACQ_KINDS = (("zc", 4512, 2000), ("chirp", 120000, 62400))
ACQ_PER_POINT = 2
ACQ_SEED = 20261005


def txcfo_input(R, n, idx):
    pre = R.zc_generate(5)
    if n <= len(pre):
        return pre[:n].copy()
    return (np.random.default_rng(7100 + idx).standard_normal(n) * 0.3).astype(np.float32)


def cfo_fixture(R):
    """applyTxCFO outputs, WattersonChannel outputs with cfo_hz / random_cfo_max_hz, and the config-4 grid: per grid
    point and kind ACQ_PER_POINT buffers built from the recipe of ria_amd/sweep.py (acq_recipe / make_acq_buffers ==
    pyoracle.acq_buffer), their checksums and the detectors' result records (ZC with known_cfo = the grid CFO)."""
    import zlib
    sys.path.insert(0, os.path.dirname(po.HERE))
    from ria_amd.sweep import acq_recipe
    rec = {"txcfo_cases": np.array(TXCFO_CASES, np.float64), "chancfo_cases": np.array(CHANCFO_CASES, np.float64)}
    for i, (n, cfo, ph) in enumerate(TXCFO_CASES):
        y, p1 = R.apply_tx_cfo(txcfo_input(R, n, i), cfo, ph)
        rec[f"txcfo_y_{i}"] = y
        rec[f"txcfo_phase_{i}"] = np.float32(p1)
    x = np.load(os.path.join(OUT, "channel_vectors.npz"))["x"]          # 6000 samples, 200 leading zeros
    for i, (kind, cfo, rmax) in enumerate(CHANCFO_CASES):
        y, actual = R.channel_cfo(kind, 15.0, 177 + i, x, cfo, rmax)
        rec[f"chancfo_y_{i}"] = y
        rec[f"chancfo_actual_{i}"] = np.float32(actual)
    y, _ = R.channel_cfo(0, 10.0, 5, x[:255], 25.0, 0.0)                # below applyCFO's 256-sample gate
    rec["chancfo_short"] = y
    pres = {"zc": R.zc_generate(5), "chirp": R.chirp_generate()}
    crc, res = {"zc": [], "chirp": []}, {"zc": [], "chirp": []}
    for gi, (cfo, snr) in enumerate(ACQ_GRID):
        for ki, (kind, buf_len, max_off) in enumerate(ACQ_KINDS):
            offs, seeds = acq_recipe(ACQ_SEED, gi, ki, np.arange(ACQ_PER_POINT), max_off)
            for q in range(ACQ_PER_POINT):
                buf = po.acq_buffer(R, pres[kind], buf_len, int(offs[q]), int(seeds[q]), snr, cfo, "tx")
                crc[kind].append(zlib.crc32(buf.tobytes()))
                res[kind].append(R.zc_detect(buf, 0.3, 15, cfo) if kind == "zc" else R.chirp_detect(buf, 0.15))
    for kind in ("zc", "chirp"):
        rec[f"acq_{kind}_crc"] = np.array(crc[kind], np.uint32)
        rec[f"acq_{kind}_results"] = np.stack(res[kind])
    rec["acq_seed"] = np.int64(ACQ_SEED)
    return rec


# ---- the MC-DPSK plug-in object itself (src/waveform/mc_dpsk_waveform.cpp) in StreamingDecoder's call order
# (carriers, modulation, spreading, data_sync (0 dual chirp, 1 ZC), snr dB, transmitter CFO Hz, known CFO Hz, lead silence, channel preset)
MCWF_CASES = [(10, po.DBPSK, 1, 0, 6.0, 0.0, 0.0, 3000, 0), (10, po.DBPSK, 4, 0, -6.0, 12.0, 0.0, 700, 0), (10, po.DQPSK, 1, 1, 10.0, 3.0, 0.0, 1488, 0),
              (10, po.DBPSK, 2, 1, 2.0, -18.0, -20.0, 930, 0), (8, po.DQPSK, 1, 0, 12.0, -30.0, 0.0, 5000, 1), (10, po.DBPSK, 1, 1, 8.0, 40.0, 38.5, 0, 0),
              (5, po.DBPSK, 1, 1, 15.0, 0.0, 0.0, 2015, 0), (10, po.DBPSK, 1, 1, -25.0, 0.0, 0.0, 1000, 0),
              # the ZC search's coarse grid (step 31) can miss the 8-sample main lobe: a reference lock 608 samples early, reproduced as it is
              (5, po.DBPSK, 1, 1, 15.0, 0.0, 0.0, 2000, 0), (10, po.DQPSK, 2, 1, 4.0, 9.0, 8.0, 3100, 2)]


def mcwf_buffer(checker, case, idx):
    """(info21, buffer): one R1/4 codeword through the plug-in's TX, the simulator's TX CFO, lead / tail silence and the channel"""
    nc, mod, sp, data_sync, snr, cfo, known, lead, kind = case
    rng = np.random.default_rng(8300 + idx)
    info = rng.integers(0, 256, 21, dtype=np.uint8)
    info[-1] &= 0xC0
    coded = checker.ldpc_encode(po.R1_4, info)[:81]
    tx = checker.mcdpsk_wf_tx(nc, mod, po.R1_4, sp, data_sync, coded)
    tx, _ = checker.apply_tx_cfo(tx, cfo)
    buf = np.concatenate([np.zeros(lead, np.float32), tx, np.zeros(1024, np.float32)])
    return info, checker.channel(kind, snr, 9100 + idx, buf)


def mcwf_fixture(R):
    import zlib
    rec = {"cases": np.array(MCWF_CASES, np.float64)}
    for i, case in enumerate(MCWF_CASES):
        nc, mod, sp, data_sync, snr, cfo, known, lead, kind = case
        info, x = mcwf_buffer(R, case, i)
        sync4, llr, aux5 = R.mcdpsk_wf_rx(nc, mod, po.R1_4, sp, data_sync, x, known)
        rec[f"crc_{i}"] = np.uint32(zlib.crc32(x.tobytes())); rec[f"sync_{i}"] = sync4; rec[f"llr_{i}"] = llr; rec[f"aux_{i}"] = aux5
        dec = np.zeros(23, np.int32)
        if len(llr) >= 648:
            ok, out, it, tries = R.robust_decode(po.R1_4, llr[:648])
            dec[:3] = [int(ok), it, tries]; dec[3:3 + 20] = out[:20]
        rec[f"dec_{i}"] = dec
        rec[f"sizes_{i}"] = R.mcdpsk_wf_sizes(nc, mod, po.R1_4, sp, 3)
        print("mc-dpsk waveform case", i, case, "sync", sync4, "soft", len(llr), "decoded", dec[:3], "payload ok", bool(dec[0]) and np.array_equal(dec[3:23].astype(np.uint8), info[:20]))
    return rec


TOOL_SNRS = (-16.0, -14.0, -12.0, -10.0, -8.0, -6.0, -4.0, -2.0, 0.0)   # tools/test_spreading.cpp:166
TOOL_MODES = (0, 2, 4)                                                    # NONE, TIME_2X, TIME_4X (:181-185)
TOOL_TRIALS = 20                                                          # seeds 1000 + trial (:167,:198)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def run_tool(name):
    """run one of the reference's own test programs (compiled unmodified by `make -C oracle tools`) and return its stdout"""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "tools", name)
    return subprocess.run([exe], capture_output=True, text=True, check=False, timeout=600).stdout


def zc_dbpsk_cases():
    """(snr, seed) of tools/test_zc_dbpsk.cpp:741-747 (sweep) and :773-777 (first step of the floor search, which ends there)"""
    out = []
    snr = np.float32(-15.0)
    while snr <= np.float32(10.0):
        out += [(float(snr), int(np.float32(snr * np.float32(1000)) + np.float32(t)) & 0xFFFFFFFF) for t in range(10)]
        snr = np.float32(snr + np.float32(2.5))
    out += [(-5.0, int(np.float32(-5000.0) + np.float32(t + 1000)) & 0xFFFFFFFF) for t in range(20)]
    return out


def tool_tables_fixture(R):
    """The scenarios of the reference's own test programs for this path (tools/test_zc_sync.cpp, tools/test_spreading.cpp,
    tools/test_chase_cache.cpp, tools/test_zc_dbpsk.cpp), rebuilt by oracle/ref_shim_tools.cpp, with the reference's results; the programs are also run
    and what they print (correlations, pass counts, the success table) is checked against the rebuilt scenarios and recorded.
    Inputs are recorded as CRC-32 only: the restatement regenerates them anywhere (tests/test_oracle_golden.py)."""
    rec = {}
    z = R.tool_zc_cases()
    rec["zc_crc"] = np.array([crc(z["signals"][i, :z["lengths"][i]]) for i in range(len(z["lengths"]))], np.uint32)
    for k in ("lengths", "test", "type", "param", "res7"):
        rec["zc_" + k] = z[k]
    passed = (z["res7"][:, 0] == 1) & (z["res7"][:, 1] == z["type"])
    txt = run_tool("test_zc_sync")
    printed = [float(v) for v in re.findall(r"\(.*corr=([-0-9.]+)", txt)]          # the PASS / FAIL lines of tests 0-3
    assert len(printed) == 30 and all(abs(a - float(b)) < 0.00051 for a, b in zip(printed, z["res7"][:30, 3])), "test_zc_sync correlations"
    counts = [int(passed[(z["test"] == t)].sum()) for t in range(5)]
    cfo_ok = int((passed & (np.abs(z["res7"][:, 4] - z["param"]) < 5.0))[z["test"] == 3].sum())   # test 3 also wants |cfo error| < 5 Hz
    counts[3] = cfo_ok
    tool_counts = [int(a) for a, b in re.findall(r"Result: (\d+)/(\d+)", txt)]
    assert tool_counts == counts, (tool_counts, counts)
    rec["zc_tool_pass_counts"] = np.array(counts, np.int32)
    # ---- test_spreading
    shp = (len(TOOL_MODES), len(TOOL_SNRS), TOOL_TRIALS)
    rec["sp_frame_crc"] = np.zeros(shp, np.uint32); rec["sp_soft_crc"] = np.zeros(shp, np.uint32)
    rec["sp_decoded"] = np.zeros(shp + (40,), np.uint8); rec["sp_bit_errors"] = np.zeros(shp, np.int32); rec["sp_ok"] = np.zeros(shp, np.uint8)
    rec["sp_tx"] = np.zeros((TOOL_TRIALS, 40), np.uint8); rec["sp_n_soft"] = np.zeros(len(TOOL_MODES), np.int32)
    for mi, m in enumerate(TOOL_MODES):
        for si, snr in enumerate(TOOL_SNRS):
            for t in range(TOOL_TRIALS):
                c = R.tool_spreading_case(snr, m, 1000 + t)
                rec["sp_frame_crc"][mi, si, t] = crc(c["frame"]); rec["sp_soft_crc"][mi, si, t] = crc(c["soft"])
                rec["sp_decoded"][mi, si, t] = c["decoded"]; rec["sp_bit_errors"][mi, si, t] = c["bit_errors"]; rec["sp_ok"][mi, si, t] = c["ok"]
                rec["sp_tx"][t] = c["tx"]; rec["sp_n_soft"][mi] = len(c["soft"])
    table = ((rec["sp_ok"] == 1) & (rec["sp_bit_errors"] == 0)).sum(axis=2).astype(np.int32)
    txt = run_tool("test_spreading")
    printed = np.array([int(a) for a in re.findall(r"\|\s+(\d+)/20 \|", txt)], np.int32).reshape(len(TOOL_MODES), len(TOOL_SNRS))
    assert np.array_equal(printed, table), (printed, table)
    rec["sp_tool_success_table"] = table
    rec["sp_snrs"] = np.array(TOOL_SNRS, np.float32); rec["sp_modes"] = np.array(TOOL_MODES, np.int32)
    # ---- test_chase_cache
    l, ok = R.tool_chase_llrs()
    rec["chase_llr_crc"] = np.array([crc(v) for v in l], np.uint32)
    rec["chase_ok"] = ok
    t2 = ok[:200].reshape(100, 2).sum(0); t3 = ok[200:].reshape(50, 3).sum(0)
    txt = run_tool("test_chase_cache")
    printed = [int(a) for a in re.findall(r":\s+(\d+)/(?:100|50) \(", txt)]
    assert printed == [int(t2[0]), int(t2[1]), int(t3[0]), int(t3[1]), int(t3[2])], (printed, t2, t3)
    rec["chase_tool_counts"] = np.array(printed, np.int32)
    # ---- test_zc_dbpsk: the 11 x 10 sweep (seed = uint32(snr * 1000 + trial)) and the first step of the floor search (-5 dB, 20 trials)
    cases = zc_dbpsk_cases()
    rec["zcd_snr"] = np.array([c[0] for c in cases], np.float32); rec["zcd_seed"] = np.array([c[1] for c in cases], np.uint32)
    L = {k: [] for k in ("sig_crc", "zc7", "stage", "soft_crc", "decoded", "ok", "bit_errors", "tx")}
    for snr, seed in cases:
        c = R.tool_zc_dbpsk_case(snr, seed)
        L["sig_crc"].append(crc(c["signal"])); L["zc7"].append(c["zc7"]); L["stage"].append(c["stage"]); L["soft_crc"].append(crc(c["soft"]))
        L["decoded"].append(c["decoded"]); L["ok"].append(c["ok"]); L["bit_errors"].append(c["bit_errors"]); L["tx"].append(c["tx"])
    rec["zcd_sig_crc"] = np.array(L["sig_crc"], np.uint32); rec["zcd_zc7"] = np.stack(L["zc7"]); rec["zcd_stage"] = np.array(L["stage"], np.int32)
    rec["zcd_soft_crc"] = np.array(L["soft_crc"], np.uint32); rec["zcd_decoded"] = np.stack(L["decoded"]); rec["zcd_ok"] = np.array(L["ok"], np.uint8)
    rec["zcd_bit_errors"] = np.array(L["bit_errors"], np.int32); rec["zcd_tx"] = np.stack(L["tx"])
    good = (rec["zcd_ok"] == 1) & (rec["zcd_bit_errors"] == 0)
    sync = rec["zcd_zc7"][:110, 0].reshape(11, 10).sum(1).astype(np.int32); dec = good[:110].reshape(11, 10).sum(1).astype(np.int32)
    txt = run_tool("test_zc_dbpsk")
    rows = re.findall(r"^\s+(-?\d+\.\d)\s+(\d+\.\d)\s+(\d+\.\d)\s+(\d\.\d+)\s+(\d\.\d+)\s*$", txt, re.M)
    assert len(rows) == 11 and [int(float(r[1])) for r in rows] == (sync * 10).tolist() and [int(float(r[2])) for r in rows] == (dec * 10).tolist(), (rows, sync, dec)
    corr = rec["zcd_zc7"][:110, 3].reshape(11, 10)
    for r, cc in zip(rows, corr):   # the program sums result.sync_corr in float, trial by trial, and prints the mean to 3 decimals
        acc = np.float32(0)
        for v in cc:
            acc = np.float32(acc + v)
        assert abs(float(np.float32(acc / np.float32(10))) - float(r[4])) < 0.00051, (r, cc)
    floor = re.search(r"SNR=-5.0 dB: ([0-9.]+)% success", txt)
    assert floor and int(float(floor.group(1))) == int(good[110:].sum()) * 5, (floor, good[110:].sum())
    rec["zcd_tool_sync_counts"] = sync; rec["zcd_tool_decode_counts"] = dec; rec["zcd_tool_floor_m5_count"] = np.int32(good[110:].sum())
    print("tool tables: zc", counts, "spreading", table.tolist(), "chase", printed, "zc_dbpsk sync", sync.tolist(), "decode", dec.tolist(), "floor(-5 dB)", int(good[110:].sum()))
    return rec


def main():
    if not po.Ref.available():
        print("needs oracle/_ref/libria_ref.so (make -C oracle ref)")
        return 2
    os.makedirs(OUT, exist_ok=True)
    R = po.Ref()
    O = po.Oracle()
    rng = np.random.default_rng(20261004)

    # ---- frames
    only = os.environ.get("RIA_GOLDEN_ONLY", "")
    for entry in FRAME_SETS:
        name, mod, rate, chans = entry[:4]
        if only and name not in only.split(","):
            continue
        rng_frames = np.random.default_rng(entry[4]) if len(entry) > 4 else rng
        rec = {}
        nvis = len(entry) > 5 and entry[5]
        bytes_per_cw = {po.R1_4: 20, po.R1_3: 27, po.R1_2: 40, po.R2_3: 54, po.R3_4: 60, po.R5_6: 67}[rate]
        cap = 4 * bytes_per_cw - 19
        rec["mod"], rec["rate"] = np.int32(mod), np.int32(rate)
        L = {k: [] for k in ("payload", "info", "coded", "tx", "rx", "llr", "aux", "h", "dec_data", "dec_ok",
                             "chan", "seq")}
        for f, (kind, snr, cfo, abs_pos) in enumerate(chans):
            payload = rng_frames.integers(0, 256, cap, dtype=np.uint8)
            seq = 100 + f
            s, info, coded, bps = R.tx_frame(mod, rate, payload, seq, nvis=nvis)
            x = s * np.float32(0.8 / np.abs(s).max())  # tools/test_waveform_simple.cpp:365-371
            y = R.channel(kind, snr, 4242 + f, x)
            llr, aux, h, _ = R.rx_process(mod, rate, y, cfo, abs_pos, nvis=nvis)
            d, ok = R.decode_fixed_frame(llr, rate, True, bps)
            L["payload"].append(payload); L["info"].append(info[:4 * bytes_per_cw]); L["coded"].append(coded)
            L["tx"].append(s if f < 2 else np.zeros(0, np.float32))
            L["rx"].append(y); L["llr"].append(llr); L["aux"].append(aux); L["h"].append(h)
            L["dec_data"].append(d[:4 * bytes_per_cw]); L["dec_ok"].append(ok)
            L["chan"].append(np.array([kind, snr, cfo, abs_pos, 4242 + f], np.float64)); L["seq"].append(seq)
        rec["bps"] = np.int32(bps)
        rec["tx0"], rec["tx1"] = L["tx"][0], L["tx"][1]
        for k in ("payload", "info", "coded", "rx", "llr", "aux", "h", "dec_data", "dec_ok", "chan"):
            rec[k] = np.stack(L[k])
        rec["seq"] = np.array(L["seq"], np.int32)
        np.savez_compressed(os.path.join(OUT, f"frames_{name}.npz"), **rec)
        print(name, "frames", len(chans), "decoded", [int(o.all()) for o in L["dec_ok"]])

    if only == "harq":
        np.savez_compressed(os.path.join(OUT, "harq_trials.npz"), **harq_fixture(R))
        return 0
    if only == "robust":
        np.savez_compressed(os.path.join(OUT, "robust_ldpc.npz"), **robust_fixture(R))
        return 0
    if only == "mcwf":
        np.savez_compressed(os.path.join(OUT, "mcdpsk_waveform.npz"), **mcwf_fixture(R))
        return 0
    if only == "cfo":
        np.savez_compressed(os.path.join(OUT, "cfo_impairment.npz"), **cfo_fixture(R))
        return 0
    if only == "tools":
        np.savez_compressed(os.path.join(OUT, "ref_tool_tables.npz"), **tool_tables_fixture(R))
        return 0
    if only == "burst":
        np.savez_compressed(os.path.join(OUT, "burst_chain.npz"), **burst_fixture(R, O))
        np.savez_compressed(os.path.join(OUT, "burst_interleaver.npz"), **burst_interleaver_fixture(R))
        return 0
    if only:
        return 0
    # ---- raw LDPC vectors per rate (encode + decode with iteration counts)
    rec = {}
    for rate in (po.R1_4, po.R1_2, po.R2_3, po.R3_4, po.R5_6):
        k = {po.R1_4: 162, po.R1_2: 324, po.R2_3: 432, po.R3_4: 486, po.R5_6: 540}[rate]
        infos, cws, llrs, res = [], [], [], []
        for t in range(24):
            info = rng.integers(0, 256, (k + 7) // 8, dtype=np.uint8)
            if k % 8:
                info[-1] &= (0xFF << (8 - k % 8)) & 0xFF
            cw = R.ldpc_encode(rate, info)[:81]
            bits = np.unpackbits(cw)[:648].astype(np.float32)
            sigma = [0.45, 0.6, 0.7, 0.8, 0.9, 1.0][t % 6] * (1.4 if rate == po.R1_4 else 1.0) * (0.75 if rate >= po.R3_4 else 1.0)
            llr = ((1.0 - 2.0 * bits) + rng.normal(0, sigma, 648)) * (2.0 / sigma ** 2)
            llr = np.clip(llr, -20, 20).astype(np.float32)
            row = []
            for factor, mi in ((0.9375, 80), (0.75, 50), (0.5, 80)):
                ok, out, it = R.ldpc_decode(rate, llr, mi, factor)
                row.append(np.concatenate([[int(ok), it], out[:(k + 7) // 8]]).astype(np.int32))
            infos.append(info); cws.append(cw); llrs.append(llr); res.append(np.stack(row))
        rec[f"info_{rate}"] = np.stack(infos)
        rec[f"cw_{rate}"] = np.stack(cws)
        rec[f"llr_{rate}"] = np.stack(llrs)
        rec[f"res_{rate}"] = np.stack(res)  # [t, 3 configs, 2 + nbytes]
    rec["configs"] = np.array([[0.9375, 80], [0.75, 50], [0.5, 80]], np.float32)
    np.savez_compressed(os.path.join(OUT, "ldpc_vectors.npz"), **rec)
    print("ldpc vectors written")

    # ---- channel interleaver permutations
    rec = {}
    for bps in (188, 106, 110, 204, 255, 306, 94, 60, 20, 10):
        rec[f"inv_{bps}"] = R.channel_interleaver_inv(bps)
    np.savez_compressed(os.path.join(OUT, "channel_interleaver.npz"), **rec)

    # ---- channel model output (bit-exact restatement target) on a short deterministic input
    t = np.arange(6000, dtype=np.float32)
    x = (0.3 * np.sin(2 * np.pi * 1500.0 * t / 48000.0)).astype(np.float32)
    x[:200] = 0
    rec = {"x": x}
    for kind in range(5):
        rec[f"y_{kind}"] = R.channel(kind, 15.0, 77 + kind, x)
    np.savez_compressed(os.path.join(OUT, "channel_vectors.npz"), **rec)
    # ---- ZC acquisition (sync::ZCSync): preambles and detect() results on noisy, CFO-shifted buffers
    from check_against_ref import zc_test_buffer
    rng = np.random.default_rng(4242)
    rec = {f"preamble_{root}": R.zc_generate(root) for root in (1, 3, 5, 7)}
    bufs, res, par = [], [], []
    for t in range(40):
        root = (1, 3, 5, 7)[t % 4]
        snr_db = (-10, -5, 0, 5, 10)[t % 5]
        cfo = (-20.0, -10.0, 0.0, 10.0, 20.0)[(t // 2) % 5]
        off = int(rng.integers(0, 2000))
        x = zc_test_buffer(rec[f"preamble_{root}"], 4512, off, snr_db, cfo, rng)
        mask, known = ((15, 0.0), (12, 0.0), (15, cfo))[t % 3]
        bufs.append(x); par.append((root, snr_db, cfo, off, mask, known)); res.append(R.zc_detect(x, 0.3, mask, known))
    rec.update(buffers=np.stack(bufs), params=np.array(par, np.float32), results=np.stack(res))
    np.savez_compressed(os.path.join(OUT, "zc_sync.npz"), **rec)
    # ---- dual-chirp acquisition: the buffers (480 KB each) are regenerated by the tests from the recipe
    # (seeded numpy noise + the preamble); the fixture holds the recipe, a checksum per buffer and the
    # reference's DualChirpResult
    rec = chirp_fixture(R)
    np.savez_compressed(os.path.join(OUT, "chirp_sync.npz"), **rec)
    np.savez_compressed(os.path.join(OUT, "mcdpsk.npz"), **mcdpsk_fixture(R))
    np.savez_compressed(os.path.join(OUT, "lts_sync.npz"), **lts_fixture(R, O))
    np.savez_compressed(os.path.join(OUT, "cox_sync.npz"), **cox_fixture(R))
    np.savez_compressed(os.path.join(OUT, "burst_chain.npz"), **burst_fixture(R, O))
    np.savez_compressed(os.path.join(OUT, "burst_interleaver.npz"), **burst_interleaver_fixture(R))
    np.savez_compressed(os.path.join(OUT, "robust_ldpc.npz"), **robust_fixture(R))
    np.savez_compressed(os.path.join(OUT, "harq_trials.npz"), **harq_fixture(R))
    np.savez_compressed(os.path.join(OUT, "cfo_impairment.npz"), **cfo_fixture(R))
    np.savez_compressed(os.path.join(OUT, "mcdpsk_waveform.npz"), **mcwf_fixture(R))
    np.savez_compressed(os.path.join(OUT, "ref_tool_tables.npz"), **tool_tables_fixture(R))
    print("done ->", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
