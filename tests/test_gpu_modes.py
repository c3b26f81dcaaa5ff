"""GPU parity tests (-m gpu) of the execution modes and ABI paths around the hot path: the batch split the bench
runs in, per-frame offsets into one capture, the burst marker / burst chain, the decode flag
without channel de-interleave.  Everything through the C ABI, bit-exact against the oracle / the reference fixtures."""
import os
import subprocess
import threading

import numpy as np
import pytest

import pyoracle as po
from test_gpu_parity import bits, dev, engine
from test_oracle_golden import burst_cases, burst_cfo_feedback

pytestmark = pytest.mark.gpu

STATUS_FIELDS = ("cw_ok", "iterations", "attempts", "frame_valid", "needs_recovery", "reserved")


def _bench_batch(e, n, first, seed=20261004, kind=2, snr=20.0):
    """n frames of the bench workload (bench.py): make_frames -> tx(peak 0.8) -> the reference's own channel stream"""
    info = e.make_frames(seed, first, n)
    x = e.tx(info, peak=0.8)
    e.channel_exact_(x, kind, snr, seed, first_frame=first)
    return info, x


def test_rx_batch_split_modes_are_bit_identical_and_match_the_reference(oracle):
    """The bench's own execution mode: ria_gpu_rx_batch cuts a batch of >= 4096 frames into parts on internal
    streams that share one workspace (slots / offsets).  9 000 faded bench frames through the default split (3), 2
    and 4 parts and the single-stream path, each twice on the same handle (workspace reuse): payload bytes and every
    ria_decode_status field identical across all runs; a 640-frame sample spread over the part boundaries equals
    oracle.decode_fixed_frame (restatement of frame_v2.cpp:1335-1883) on the GPU's own LLRs, the LLRs equal
    oracle.rx_process on a subsample, and - where oracle/_ref is present - the compiled reference itself."""
    import torch
    from ria_amd.engine import RxEngine
    e = RxEngine("QAM16", "R1_2", max_batch=9000)
    n, first = 9000, 25000 * 5 + 321
    info, x = _bench_batch(e, n, first)
    runs = {}
    for parts in (1, 0, 3, 4, 2, 1):          # 0 = library default (3 unless the environment says otherwise)
        e.set_split_parts(parts)
        for rep in range(2):
            out, st = e.rx(x)
            torch.cuda.synchronize()
            runs.setdefault(parts, []).append((out.cpu().numpy().copy(), e.decode_status(st).copy()))
    # the two-codewords-per-wave retry kernels (ldpc_dual.hip.h) on the same batch, one stream and the default split
    # (an experiment record: only in libraries built with -DRIA_WITH_DUAL_DECODER; the default build refuses the option)
    from ria_amd import capi
    try:
        e.set_dual_decoder(1)
        has_dual = True
    except capi.RiaError:
        has_dual = False
    for parts in (1, 0) if has_dual else ():
        e.set_split_parts(parts)
        e.set_dual_decoder(1)
        out, st = e.rx(x)
        torch.cuda.synchronize()
        runs.setdefault(("dual", parts), []).append((out.cpu().numpy().copy(), e.decode_status(st).copy()))
    e.set_dual_decoder(0)
    base_out, base_st = runs[1][0]
    for parts, lst in runs.items():
        for rep, (o, s) in enumerate(lst):
            assert np.array_equal(o, base_out), f"parts {parts} run {rep}: payload bytes differ from the single-stream run"
            for k in STATUS_FIELDS:
                assert np.array_equal(s[k], base_st[k]), f"parts {parts} run {rep}: status field {k}"
    # the workload must exercise the retry machinery in every part
    att = base_st["attempts"]
    for lo, hi in ((0, 2250), (2250, 4500), (4500, 6750), (6750, 9000)):
        assert (att[lo:hi] > 5).any() and (att[lo:hi] == 1).any()
    # sample across the part boundaries of the 2-, 3- and 4-way cuts vs the oracle (and the reference)
    e.set_split_parts(0)
    out, st, llr, fst = e.rx(x, want_llr=True)
    assert np.array_equal(out.cpu().numpy(), base_out)
    llr = llr.cpu().numpy()
    sample = np.unique(np.concatenate([np.arange(0, 64), np.arange(2218, 2282), np.arange(2968, 3032), np.arange(4468, 4532),
                                       np.arange(5968, 6032), np.arange(6718, 6782), np.arange(8936, 9000),
                                       np.random.default_rng(5).integers(0, n, 192)]))
    y = x[torch.from_numpy(sample).cuda()].cpu().numpy()
    ref = po.Ref() if po.Ref.available() else None
    if ref is not None:
        ref.rx_process(po.QAM16, po.R1_2, y[0])
    bad = []

    def work(lo, hi):
        for q in range(lo, hi):
            f = int(sample[q])
            if q % 8 == 0:
                lo_, _ = oracle.rx_process(po.QAM16, po.R1_2, y[q])
                if not np.array_equal(bits(lo_), bits(llr[f])):
                    bad.append((f, "llr"))
            d, ok, iters, attn = oracle.decode_fixed_frame(llr[f], po.R1_2, True, 188, flags=7)
            if not (np.array_equal(base_st["cw_ok"][f], ok) and np.array_equal(base_out[f], d)
                    and np.array_equal(base_st["iterations"][f], iters.astype(np.uint16))
                    and np.array_equal(base_st["attempts"][f], attn.astype(np.uint8))):
                bad.append((f, "oracle"))
            if ref is not None:
                rl = ref.rx_process(po.QAM16, po.R1_2, y[q])[0]
                rd, rok = ref.decode_fixed_frame(rl, po.R1_2, True, 188)
                if not np.array_equal(base_st["cw_ok"][f], rok) or any(rok[c] and not np.array_equal(base_out[f][40 * c:40 * c + 40], rd[40 * c:40 * c + 40]) for c in range(4)):
                    bad.append((f, "reference"))
    nt = 16
    th = [threading.Thread(target=work, args=(k * len(sample) // nt, (k + 1) * len(sample) // nt)) for k in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not bad, bad[:8]
    assert e.lib.ria_gpu_debug_queue_fault(e.h) == 0, "a persistent work-queue loop left through its iteration bound"
    e.close()


def test_frame_offsets_into_one_capture(oracle):
    """frame_offsets_dev: frames at arbitrary (odd, unaligned, overlapping-free) sample offsets inside ONE capture give
    the same LLRs, status and payload as the same frames laid out as rows; 4 200 frames so that the split path of the
    fused call walks the offset array too."""
    import torch
    from ria_amd.engine import RxEngine
    e = RxEngine("QAM16", "R1_2", max_batch=4200)
    n = 4200
    info, x = _bench_batch(e, n, 777, seed=99, kind=1, snr=17.0)
    fl = e.geo.frame_samples
    rng = np.random.default_rng(12)
    gaps = rng.integers(0, 37, n)                      # 0..36 samples of junk between frames: most offsets not 16-byte aligned
    offs = (np.cumsum(gaps) + np.arange(n) * fl).astype(np.uint64)
    cap = torch.full((int(offs[-1]) + fl + 5,), 0.123, dtype=torch.float32, device="cuda")
    idx = (torch.from_numpy(offs.astype(np.int64)).cuda()[:, None] + torch.arange(fl, device="cuda")[None, :]).reshape(-1)
    cap[idx] = x.reshape(-1)
    cfo = rng.uniform(-4, 4, n).astype(np.float32)
    abs_pos = rng.integers(0, 1 << 20, n).astype(np.uint64)
    a = e.rx(x, cfo_hz=cfo, abs_pos=abs_pos, want_llr=True)
    b = e.rx(cap, cfo_hz=cfo, abs_pos=abs_pos, want_llr=True, offsets=offs)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
    l1, _ = e.demod(cap, cfo_hz=cfo, abs_pos=abs_pos, offsets=offs)
    assert torch.equal(l1.view(torch.int32), a[2].view(torch.int32))
    # and the offsets mean what the header says: a few frames against the oracle run on the capture slice itself
    caph = cap.cpu().numpy()
    for f in (0, 1, 2099, 2100, 4199):
        lo, _ = oracle.rx_process(po.QAM16, po.R1_2, caph[int(offs[f]):int(offs[f]) + fl], float(cfo[f]), int(abs_pos[f]))
        assert np.array_equal(bits(lo), bits(b[2][f].cpu().numpy())), f
    e.close()


@pytest.mark.parametrize("mod,rate", [("QAM16", "R1_2"), ("DQPSK", "R1_4")])
def test_decode_without_channel_deinterleave(oracle, mod, rate):
    """RIA_DECODE_NO_CHANNEL_DEINTERLEAVE = decodeFixedFrame(soft, rate, use_channel_interleave=false, bps)
    (frame_v2.cpp:1347-1357): frames encoded without the per-codeword channel interleaver decode through the flag,
    bit-exact against the oracle and the reference; the same soft bits without the flag do not."""
    from ria_amd import capi
    e = engine(mod, rate)
    pm, pr = capi.MOD[mod], capi.RATE[rate]
    g = oracle.geom(pm, pr)
    rng = np.random.default_rng(77)
    frames, infos = [], []
    for f in range(24):
        info = oracle.make_frame(rng.integers(0, 256, 4 * g.bytes_per_cw - 19, dtype=np.uint8), f, pr)
        coded = oracle.encode_fixed_frame(info, pr, False, g.bits_per_symbol)
        s = oracle.modulate(pm, pr, coded)
        s = s * np.float32(0.8 / np.abs(s).max())
        kind, snr = [(0, 14.0), (2, 18.0), (1, 12.0)][f % 3] if mod == "QAM16" else [(0, 2.0), (2, 6.0), (1, 3.0)][f % 3]
        frames.append(oracle.channel(kind, snr, 300 + f, s))
        infos.append(info)
    llr, _ = e.demod(dev(np.stack(frames)))
    out, st = e.decode(llr, flags=capi.DECODE_FULL | capi.DECODE_NO_CHANNEL_DEINTERLEAVE)
    out, s = out.cpu().numpy(), e.decode_status(st)
    llr_h = llr.cpu().numpy()
    ref = po.Ref() if po.Ref.available() else None
    n_ok = 0
    for f in range(24):
        d, ok, iters, att = oracle.decode_fixed_frame(llr_h[f], pr, False, g.bits_per_symbol, flags=7)
        assert np.array_equal(s["cw_ok"][f], ok) and np.array_equal(out[f], d), f
        assert np.array_equal(s["iterations"][f], iters.astype(np.uint16)) and np.array_equal(s["attempts"][f], att.astype(np.uint8))
        if ref is not None:
            rd, rok = ref.decode_fixed_frame(llr_h[f], pr, False, g.bits_per_symbol)
            assert np.array_equal(rok, ok) and np.array_equal(rd[:len(d)], d)
        n_ok += int(ok.all() and np.array_equal(d, infos[f]))
    assert n_ok >= 8
    out2, st2 = e.decode(llr, flags=capi.DECODE_FULL)
    assert not e.decode_status(st2)["frame_valid"].any()


def test_burst_chain_end_to_end_vs_reference_golden(oracle, golden):
    """SURVEY.md 8f rank 3 + a9 marker, through the C ABI in StreamingDecoder's order: ria_gpu_sync_lts_batch on the
    capture (marker from the negated first LTS) -> ria_gpu_demod_batch frame by frame at offsets inside the capture
    (meta flag bit0 on the group's first frame, CFO fed back with the host's 2 Hz clamp) -> ria_gpu_burst_deinterleave_batch
    -> ria_gpu_decode_batch.  Soft bits, CFOs, de-interleaved soft bits, codeword flags and bytes all identical to
    what ONE reference OFDMChirpWaveform + BurstInterleaver + decodeFixedFrame produced (tests/golden/burst_chain.npz)."""
    import torch
    from ria_amd import capi
    names = {v: k for k, v in capi.MOD.items()}, {v: k for k, v in capi.RATE.items()}
    for i, case, x, g in burst_cases(golden, oracle):
        mod, rate, n, lead, kind, snr, cfo0, abs_base, marker = case
        n, abs_base = int(n), int(abs_base)
        e = engine(names[0][int(mod)], names[1][int(rate)])
        fl = e.geo.frame_samples
        r = e.sync_lts(dev(x[None, :21000]), dev(np.array([cfo0], np.float32)), 0.5)[0]
        gs = g[f"sync_{i}"]
        assert int(r["detected"]) == 1 and int(r["start_sample"]) == int(gs[1]) and int(r["burst_interleaved"]) == int(gs[3]) == marker
        assert np.float32(r["correlation"]).view(np.uint32) == gs[2].view(np.uint32)
        start = int(r["start_sample"])
        cap = dev(x)
        cfo = np.float32(cfo0)
        llrs = []
        for f in range(n):
            assert cfo == g[f"cfo_used_{i}"][f]
            fl_flag = 1 if (f == 0 and int(r["burst_interleaved"])) else 0
            llr, st = e.demod(cap, cfo_hz=np.array([cfo], np.float32), abs_pos=np.array([abs_base + start], np.uint64),
                              flags=np.array([fl_flag], np.uint32), offsets=np.array([start + f * fl], np.uint64))
            assert np.array_equal(bits(llr.cpu().numpy()[0]), bits(g[f"llr_{i}"][f])), f"case {i} frame {f}: soft bits"
            fs = e.frame_status(st)
            assert fs["cfo_hz"][0] == g[f"cfo_after_{i}"][f]
            cfo = burst_cfo_feedback(cfo, fs["cfo_hz"][0])
            llrs.append(llr)
        phys = torch.cat(llrs).contiguous()
        logical = e.burst_deinterleave(phys, n)
        assert np.array_equal(bits(logical.cpu().numpy()[:, :2592]), bits(g[f"logical_{i}"])), f"case {i}: de-interleave"
        out, st = e.decode(logical)
        s = e.decode_status(st)
        assert np.array_equal(s["cw_ok"], g[f"dec_ok_{i}"]), f"case {i}: {s['cw_ok']} vs {g[f'dec_ok_{i}']}"
        assert np.array_equal(out.cpu().numpy(), g[f"dec_data_{i}"]), f"case {i}: bytes"
        # the same group in ONE fused call per frame position is what a batched Monte-Carlo harness does: marker flag on
        # a whole batch of first frames (here: the same frame 3 times, unmarked copy in between)
        if marker:
            three = cap[start:start + fl].repeat(3, 1).contiguous()
            l3, _ = e.demod(three, cfo_hz=np.full(3, cfo0, np.float32), abs_pos=np.full(3, abs_base + start, np.uint64),
                            flags=np.array([1, 0, 1], np.uint32))
            l3 = l3.cpu().numpy()
            assert np.array_equal(bits(l3[0]), bits(g[f"llr_{i}"][0])) and np.array_equal(bits(l3[2]), bits(l3[0]))
            lo, _ = oracle.rx_process(int(mod), int(rate), x[start:start + fl], float(cfo0), abs_base + start)
            assert np.array_equal(bits(l3[1]), bits(lo))


def test_cpp_adaptor_burst_marker_is_latched(oracle, golden, tmp_path):
    """The IWaveform mirror (ria_amd/host/gpu_waveform.hpp) keeps the reference's TWO marker flags
    (ofdm_chirp_waveform.hpp:136-140): process() consumes the one-shot, wasBurstInterleaved() still reports the group
    after process() - which is when StreamingDecoder asks (streaming_decoder.cpp:1380-1383).  Driven in the decoder's
    order by a g++-built program over the C ABI; soft bits of every frame of the group against the reference."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_adaptor_test")
    lib = os.path.join(root, "ria_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(root, "tests", "helpers", "host_adaptor_test.cpp"),
                           "-L" + lib, "-lria_gpu", "-Wl,-rpath," + lib])
    for i, case, x, g in burst_cases(golden, oracle):
        mod, rate, n, lead, kind, snr, cfo0, abs_base, marker = case
        if i not in (0, 2, 4):
            continue
        fin = str(tmp_path / f"burst{i}.f32")
        x.tofile(fin)
        outp = str(tmp_path / f"b{i}")
        lines = subprocess.check_output([exe, str(int(mod)), str(int(rate)), fin, repr(float(cfo0)), str(int(abs_base)), outp, "3",
                                         str(int(n))]).decode().strip().split("\n")
        head = lines[0].split()
        gs = g[f"sync_{i}"]
        assert int(head[0]) == 1 and int(head[1]) == int(gs[1]) and np.float32(float(head[2])) == gs[2] and int(head[3]) == marker
        for f in range(int(n)):
            t = lines[1 + f].split()
            assert int(t[0]) == 1 and int(t[1]) == g[f"llr_{i}"].shape[1]
            assert np.float32(float(t[2])) == g[f"cfo_used_{i}"][f] and np.float32(float(t[3])) == g[f"cfo_after_{i}"][f]
            assert int(t[4]) == marker, "wasBurstInterleaved() must stay latched after process()"
            llr = np.fromfile(f"{outp}.{f}.llr", np.float32)
            assert np.array_equal(bits(llr), bits(g[f"llr_{i}"][f])), (i, f)


def test_robust_single_cw_decode_vs_reference_golden(golden):
    """ria_gpu_ldpc_decode_robust_batch = robustDecodeSingleCW (streaming_decoder.cpp:1028-1058): success, number of
    decodes, iteration count of the last one and its bytes, against vectors recorded from the reference."""
    g = golden("robust_ldpc")
    for rate, rn in ((po.R1_4, "R1_4"), (po.R1_2, "R1_2"), (po.R3_4, "R3_4"), (po.R1_3, "R1_3"), (po.R2_3, "R2_3"), (po.R5_6, "R5_6")):
        e = engine("QAM16", rn)
        r = g[f"res_{rate}"]
        out, ok, it, tries = e.ldpc_decode_robust(dev(g[f"llr_{rate}"]))
        assert np.array_equal(ok.cpu().numpy(), r[:, 0].astype(np.uint8)), rn
        assert np.array_equal(tries.cpu().numpy(), r[:, 1].astype(np.uint8)), rn
        assert np.array_equal(it.cpu().numpy(), r[:, 2].astype(np.int16)), rn
        nb = out.shape[1]
        assert np.array_equal(out.cpu().numpy(), r[:, 3:3 + nb].astype(np.uint8)), rn


def test_harq_chain_vs_reference_golden(golden):
    """BASELINE config 5, MC-DPSK rungs (DBPSK 4x / 2x / no spreading, DQPSK) at two marginal SNRs each plus two faded
    cases, 192 trials per case: MC-DPSK modulator -> reference-identical channel (per-trial seeds) -> demodulator ->
    robustDecodeSingleCW -> ChaseCache::store -> robustDecodeSingleCW of the sum, up to 4 transmissions.  Per trial the
    number of transmissions to success, per reception the checksum of the 648 soft bits and of the cache sum, the decoder
    tries, the decoded bytes and the demodulator's fading index: all identical to what the reference's own classes
    produced (tests/golden/harq_trials.npz, recorded by oracle/ref_shim.cpp ref_harq_trials)."""
    import gen_golden
    from ria_amd import sweep
    e = engine("DQPSK", "R1_4")
    g = golden("harq_trials")
    for i, (nc, bps, sp, kind, snr) in enumerate(gen_golden.HARQ_CASES):
        info, seeds = gen_golden.harq_inputs(i)
        r = sweep.run_harq_trials(e, nc, bps, sp, kind, snr, info, seeds, want_crc=True)
        for k in ("tx_to_success", "llr_crc", "acc_crc", "tries", "decoded"):
            bad = np.nonzero((r[k] != g[f"{k}_{i}"]).reshape(len(info), -1).any(axis=1))[0]
            assert len(bad) == 0, f"case {i} {gen_golden.HARQ_CASES[i]}: {k} differs for trials {bad[:6]}"
        m = r["llr_crc"] != 0
        assert np.array_equal(r["fading"][m].view(np.uint32), g[f"fading_{i}"][m].view(np.uint32)), (i, "fading index")
    # a trial's outcome is a function of (seed, point, trial) only: any sub-batch reproduces its rows
    info, seeds = gen_golden.harq_inputs(6)
    sub = np.array([5, 17, 18, 40, 101, 190])
    r = sweep.run_harq_trials(e, 10, 2, 1, 0, -12.0, info[sub], seeds[sub])
    assert np.array_equal(r["tx_to_success"], g["tx_to_success_6"][sub]) and np.array_equal(r["decoded"], g["decoded_6"][sub])


def test_adaptive_ladder_sweep_table():
    """BASELINE config 5 end to end on one GPU at reduced trials: 3 channels x SNR -14..30 dB step 2 = 69 points, the
    ladder (recommendWaveformAndRate fed with the demodulator's measured fading index) picks the rung, MC-DPSK rungs run
    the HARQ chain above, OFDM rungs the fused RX.  Assertions = the reference's own operating-point claims
    (waveform_selection.hpp:88-111): on AWGN every rung is clean from its floor upwards (4x spreading from -14 dB, 2x from
    -8, none from -4, MC-DPSK DQPSK from +5, OFDM from 10 dB), the picked rungs are the table's, HARQ is exercised only
    below the floors, fading never makes a point better than AWGN by more than noise."""
    from ria_amd import sweep, capi
    from ria_amd.engine import RxEngine
    cache = {}

    def engines(mod, rate):
        if (mod, rate) not in cache:
            cache[(mod, rate)] = RxEngine(mod, rate, max_batch=256)
        return cache[(mod, rate)]

    lib = capi.load()
    trials, seed = 64, 4321
    snrs = np.arange(-14.0, 30.1, 2.0)
    rows, modes, fad = {}, {}, {}
    pi = 0
    for ch in (0, 1, 2):
        for snr in snrs:
            p = sweep.SweepPoint(ch, float(snr))
            fad[(ch, snr)] = sweep.measured_fading_index(engines, p, seed, pi)
            rec = sweep.ladder_mode(lib, snr, fad[(ch, snr)])
            modes[(ch, snr)] = (rec.waveform, rec.modulation, rec.code_rate, rec.spreading)
            rows[(ch, snr)] = sweep.run_ladder_chunk(engines, p, seed, pi, 0, trials, fading=fad[(ch, snr)])
            pi += 1
    assert pi == 69
    for snr in snrs:                       # AWGN: measured fading index below the 0.15 "true AWGN" threshold at usable SNR
        if snr >= 10:
            assert fad[(0, snr)] < 0.15, (snr, fad[(0, snr)])
    # the rungs the table prescribes on AWGN
    assert modes[(0, -14.0)] == (4, 0, 0, 4) and modes[(0, -8.0)] == (4, 0, 0, 4) and modes[(0, -6.0)] == (4, 0, 0, 2)
    assert modes[(0, -2.0)] == (4, 0, 0, 1) and modes[(0, 6.0)][:3] == (4, 2, 0) and modes[(0, 26.0)][:3] == (5, 8, 4)
    assert modes[(0, 22.0)][:3] == (5, 7, 4) and modes[(0, 18.0)][:2] == (5, 6) and modes[(0, 12.0)][:2] == (5, 2)
    for snr in snrs:
        r = rows[(0, snr)]
        assert r[0] == trials
        if modes[(0, snr)][0] == 4:
            assert r[1] == 0, f"AWGN {snr} dB {modes[(0, snr)]}: {r[1]} of {trials} trials failed above the rung's floor"
            assert r[5] == trials, f"AWGN {snr} dB: HARQ retransmissions above the floor"
        else:
            # OFDM rungs: the reference itself loses the ~1.2 % of frames whose codeword 1..3 starts with 0xD5
            # (reassembleCodewords strips two bytes, frame_v2.cpp:972-979) at any SNR - reproduced, not fixed
            assert r[1] <= 4, f"AWGN {snr} dB {modes[(0, snr)]}: {r[1]} of {trials} trials failed above the rung's floor"
    for ch in (1, 2):
        for snr in snrs:
            r = rows[(ch, snr)]
            assert r[0] == trials and r[5] >= trials
            if modes[(ch, snr)][0] == 4 and snr >= -4:
                assert r[1] <= trials // 4, (ch, snr, r)
    fer = {k: v[1] / trials for k, v in rows.items()}
    print("ladder table:", {f"{k[0]}:{k[1]:+.0f}": (modes[k], round(fad[k], 2), round(fer[k], 3), round(rows[k][5] / trials, 2)) for k in rows})


def test_config_c2_dqpsk_demod_only_full_size(oracle):
    """BASELINE config 2 at its full size: OFDM DQPSK R1/2, 10 000 frames, AWGN, FFT + LLR kernels only
    (ria_gpu_demod_batch).  Every 200th frame's 2 650 soft bits and estimator scalars bit-identical to the oracle's
    processPresynced; size-independent property over the whole batch: the hard decisions of the soft bits, differentially
    consistent with the transmitted coded bits at 15 dB (few bit errors per frame).  Reports the demodulator's rate on
    this shape against its algorithmic bytes (31 104 x 4 in + 2 650 x 4 out per frame)."""
    import torch
    from ria_amd.engine import RxEngine
    e = RxEngine("DQPSK", "R1_2", max_batch=10000)
    n = 10000
    info = e.make_frames(777, 0, n)
    x = e.tx(info, peak=0.8)
    e.channel_exact_(x, 0, 15.0, 777)
    assert x.shape[1] == 31104 and e.geo.llrs_per_frame == 2650
    llr, st = e.demod(x)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        llr, st = e.demod(x)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 5
    algo = n * (31104 * 4 + 2650 * 4)
    print(f"C2 demod only: {n / ms * 1e3:.0f} frames/s, {algo / ms / 1e6:.1f} GB/s algorithmic ({algo / ms / 1e6 / 8000 * 100:.2f} % of the HBM peak), {ms:.3f} ms per 10 000 frames")
    L = llr.cpu().numpy()
    xs = x[::200].cpu().numpy()
    fs = e.frame_status(st)
    for q, f in enumerate(range(0, n, 200)):
        lo, aux = oracle.rx_process(po.DQPSK, po.R1_2, xs[q])
        assert np.array_equal(bits(L[f]), bits(lo)), f
        assert np.float32(aux.noise_variance) == fs["noise_variance"][f] and np.float32(aux.fading_index) == fs["fading_index"][f]
    # whole batch: decode everything (AWGN 15 dB: essentially every frame), payload equals what was transmitted
    out, dst = e.decode(llr)
    ok = (out == info).all(dim=1).float().mean().item()
    assert ok > 0.97, ok


def test_config3_literal_size_batch_properties(oracle):
    """BASELINE config 3 at its literal size: 100 000 QAM16 R1/2 frames through Watterson moderate 20 dB in ONE ria_gpu_rx_batch
    call (the bench step).  Size-independent properties over the whole batch: (1) the call is a function of its input - a
    second call returns the same bytes and status records; (2) batch-size independence - the same frames in four calls of
    25 000 (other part boundaries, other work-list sizes, other queue interleavings) give the same records frame for frame;
    (3) decodeFixedFrame's contract holds on every frame - frame_valid exactly where all four codewords are reported decoded,
    no bytes for a codeword that is not, nothing left flagged for recovery, attempts within the cascade's 1 ... 39; (4) the decoded share is
    the workload's (64 % +- 2); and a sample of 96 frames spread over the batch equals the CPU chain bit for bit."""
    import torch
    from ria_amd.engine import RxEngine
    n = 100000
    e = RxEngine("QAM16", "R1_2", max_batch=n)
    info, x = _bench_batch(e, n, 0)
    out1, st1 = e.rx(x)
    out2, st2 = e.rx(x)
    assert torch.equal(out1, out2) and torch.equal(st1, st2)
    outs, sts = [], []
    for k in range(4):
        o, s_ = e.rx(x[25000 * k:25000 * (k + 1)])
        outs.append(o.clone()); sts.append(s_.clone())
    assert torch.equal(torch.cat(outs), out1) and torch.equal(torch.cat(sts), st1)
    s = e.decode_status(st1)
    assert not s["needs_recovery"].any() and not (s["reserved"][:, 1] == 0xEE).any()
    okf = s["cw_ok"].all(axis=1)
    assert np.array_equal(okf, s["frame_valid"].astype(bool))
    o = out1.cpu().numpy()
    assert (o.reshape(n, 4, 40)[~s["cw_ok"].astype(bool)] == 0).all()        # a codeword that is not reported decoded carries no bytes
    assert s["attempts"].min() >= 1 and s["attempts"].max() <= 39
    share = okf.mean()
    assert 0.62 < share < 0.66, share
    same = (o == info.cpu().numpy()).all(axis=1)
    assert (same & ~okf).sum() == 0 and same[okf].mean() > 0.97      # CRC-16 recoveries the reference accepts are not all the sent frame
    idx = np.linspace(0, n - 1, 96).astype(int)
    xs = x[torch.from_numpy(idx).to(x.device)].cpu().numpy()
    for q, f in enumerate(idx):
        llr, _ = oracle.rx_process(po.QAM16, po.R1_2, xs[q])
        d, ok, it, att = oracle.decode_fixed_frame(llr, po.R1_2, True, 188, flags=7)
        assert np.array_equal(ok, s["cw_ok"][f]) and np.array_equal(d, o[f]), f
        assert np.array_equal(s["iterations"][f], it.astype(np.uint16)) and np.array_equal(s["attempts"][f], att.astype(np.uint8)), f


def test_config4_and_config5_drivers_run_and_resume(tmp_path):
    """tools/run_ladder_sweep.py (config 5) and tools/run_acquisition_grid.py (config 4) as a user runs them on one GPU: a sweep
    interrupted after some points and resumed gives the table of an uninterrupted one (per-point counters are functions of
    (seed, point, global trial) only), and a second --resume has nothing left to do; the grid driver's counters have the shape of
    the committed 20 000-preamble table."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    sweep_cmd = [sys.executable, os.path.join(root, "tools", "run_ladder_sweep.py"), "--trials", "256", "--chunk", "128", "--channels", "0,2",
                 "--snr-min", "-14", "--snr-max", "22", "--snr-step", "12"]
    full = str(tmp_path / "full.jsonl"); part = str(tmp_path / "part.jsonl")
    a = json.loads(subprocess.run(sweep_cmd + ["--out", full], env=env, capture_output=True, text=True, check=True, timeout=600).stdout.strip().splitlines()[-1])
    assert a["points"] == 8 and a["points_run_now"] == 8
    lines = open(full).read().splitlines()
    open(part, "w").write("\n".join(lines[:3]) + "\n")                  # the run "died" after three points
    b = json.loads(subprocess.run(sweep_cmd + ["--out", part, "--resume"], env=env, capture_output=True, text=True, check=True, timeout=600).stdout.strip().splitlines()[-1])
    assert b["points_run_now"] == 5
    strip = lambda rows: [{k: v for k, v in r.items() if k != "seconds"} for r in rows]
    assert strip(b["table"]) == strip(a["table"])
    assert strip([json.loads(x) for x in open(part).read().splitlines()]) == strip(a["table"])
    c = json.loads(subprocess.run(sweep_cmd + ["--out", part, "--resume"], env=env, capture_output=True, text=True, check=True, timeout=600).stdout.strip().splitlines()[-1])
    assert c["points_run_now"] == 0 and strip(c["table"]) == strip(a["table"])
    g = json.loads(subprocess.run([sys.executable, os.path.join(root, "tools", "run_acquisition_grid.py"), "--preambles", "64"], env=env,
                                  capture_output=True, text=True, check=True, timeout=600).stdout.strip().splitlines()[-1])
    assert len(g["table"]) == 25 and all(r["n"] == 64 for r in g["table"])
    # the table of profiles/*_acquisition_grid.json at 20 000 preambles per point: the dual chirp finds 99-100 % everywhere, ZC (given
    # the grid's CFO as known_cfo) nothing at -10 dB and 67-78 % from 0 dB up, and whatever it detects it times correctly
    assert all(r["chirp_success"] >= 60 and r["chirp_timing_ok"] >= 60 for r in g["table"])
    assert all(r["zc_detected"] == 0 for r in g["table"] if r["snr_db"] == -10.0)
    assert all(30 <= r["zc_detected"] <= 60 for r in g["table"] if r["snr_db"] >= 0.0)
    assert all(r["zc_timing_ok"] == r["zc_detected"] for r in g["table"])


def test_bench_rccl_code_path_on_one_rank():
    """bench.py's N > 1 collectives (RCCL broadcast of the seed, all-reduce of counters and time, barrier) executed on this
    box's GPU with a one-rank process group (RIA_BENCH_FORCE_DIST=1): the line must carry the same metric and workload as
    the plain run.  (A real N > 1 run needs an N-GPU node: the driver's job.)"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RIA_BENCH_FORCE_DIST="1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4096", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["unit"] == "frames/s" and line["value"] > 1e5
    assert line["config"]["frames_decoded_last_step"] > 2000


def test_demod_with_cfo_matches_oracle_over_many_offsets(oracle):
    """The CFO correction phase is a float recurrence over every sample (channel_equalizer.cpp:132-144), walked by one
    lane with markers every 72 samples and re-walked by 16 lanes.  96 frames with CFOs
    from 0.02 to 60 Hz of both signs, random absolute positions (= random start phases incl. near +-pi and near 0),
    channel CFO applied or not: every soft bit and the corrected CFO / final phase identical to the oracle's walk."""
    e = engine("QAM16", "R1_2")
    rng = np.random.default_rng(2027)
    n = 96
    frames, cfos, poss = [], [], []
    for f in range(n):
        s, info, _ = oracle.tx_frame(po.QAM16, po.R1_2, rng.integers(0, 256, 141, dtype=np.uint8), f)
        x = s * np.float32(0.8 / np.abs(s).max())
        cfo = np.float32([0.02, 0.3, 1.0, 2.5, 7.0, 23.0, 60.0, 0.011][f % 8] * (1 if f % 3 else -1))
        if f % 4 == 0:      # the channel really carries that offset: analytic-signal rotation
            spec = np.fft.fft(x.astype(np.float64)); h = np.zeros(len(x)); h[0] = 1; h[1:len(x) // 2] = 2; h[len(x) // 2] = 1
            x = np.real(np.fft.ifft(spec * h) * np.exp(2j * np.pi * float(cfo) * np.arange(len(x)) / 48000.0)).astype(np.float32)
        frames.append(oracle.channel(0 if f % 2 else 1, 22.0, 5000 + f, x))
        cfos.append(cfo)
        poss.append(int(rng.integers(0, 1 << 22)) if f % 5 else 0)
    cfos = np.array(cfos, np.float32); poss = np.array(poss, np.uint64)
    llr, st = e.demod(dev(np.stack(frames)), cfo_hz=cfos, abs_pos=poss)
    llr, fs = llr.cpu().numpy(), e.frame_status(st)
    for f in range(n):
        lo, aux = oracle.rx_process(po.QAM16, po.R1_2, frames[f], float(cfos[f]), int(poss[f]))
        nd = int((bits(llr[f]) != bits(lo)).sum())
        assert nd == 0, f"frame {f} cfo {cfos[f]} pos {poss[f]}: {nd} soft bits differ"
        assert np.float32(aux.cfo_hz) == fs["cfo_hz"][f] and np.float32(aux.corr_phase).view(np.uint32) == fs["corr_phase"][f].view(np.uint32), f
