"""GPU parity tests that read like the reference's own test programs for this path (run with -m gpu on an MI355X):
  tools/test_zc_sync.cpp      tests 0-4  -> ria_gpu_sync_zc_batch
  tools/test_spreading.cpp    the 3 x 9 x 20 success table (MC-DPSK DBPSK 1x / 2x / 4x + LDPC R1/2, -16 ... 0 dB)
                                           -> ria_gpu_mcdpsk_demod_batch + ria_gpu_ldpc_decode_batch
  tools/test_chase_cache.cpp  tests 1-3  -> ria_gpu_chase_combine_batch + ria_gpu_ldpc_decode_batch
  tools/test_zc_dbpsk.cpp     testAtSNR, 130 cases -> ria_gpu_sync_zc_batch -> ria_gpu_mcdpsk_demod_batch -> ria_gpu_ldpc_decode_batch
The inputs are those programs' own signals: rebuilt by the CPU restatement (oracle.tool_*), CRC-32 checked against what the
compiled reference produced with the library's mt19937 / normal_distribution (tests/golden/ref_tool_tables.npz, made by
oracle/gen_golden.py from oracle/ref_shim_tools.cpp while the programs themselves ran and their printed tables were compared).
Every output field is compared bit for bit with the reference's, and each program's own PASS conditions are asserted."""
import numpy as np
import pytest

from test_gpu_parity import bits, dev, engine
from test_oracle_golden import crc32

pytestmark = pytest.mark.gpu


def test_zc_sync_program_scenarios(oracle, golden):
    e = engine("QAM16", "R1_2")
    g = golden("ref_tool_tables")
    z = oracle.tool_zc_cases()
    n = len(z["lengths"])
    assert n == 50 and all(crc32(z["signals"][i, :z["lengths"][i]]) == g["zc_crc"][i] for i in range(n))
    got = np.zeros((n, 7), np.float32)
    for ln in sorted(set(z["lengths"].tolist())):                     # 3 512-sample (500 either side) and 4 512-sample buffers
        idx = np.nonzero(z["lengths"] == ln)[0]
        r = e.sync_zc(dev(np.ascontiguousarray(z["signals"][idx, :ln])), threshold=0.2, root_mask=15)   # zc.detect(signal, 0.2f)
        for k, name in enumerate(("detected", "frame_type", "start_sample", "correlation", "cfo_hz", "snr_estimate", "root_detected")):
            got[idx, k] = r[name].astype(np.float32)
    assert np.array_equal(bits(got), bits(g["zc_res7"])), np.nonzero(bits(got) != bits(g["zc_res7"]))
    passed = (got[:, 0] == 1) & (got[:, 1] == z["type"])
    counts = [int(passed[z["test"] == t].sum()) for t in range(5)]
    counts[3] = int((passed & (np.abs(got[:, 4] - z["param"]) < 5.0))[z["test"] == 3].sum())   # test 3: detected and |CFO error| < 5 Hz
    assert counts == g["zc_tool_pass_counts"].tolist()
    # test_zc_sync.cpp:119,153,202,237,299: 4/4, 4/4, >= 80 %, >= 60 %, >= 90 %
    assert counts[0] == 4 and counts[1] == 4 and counts[2] >= 0.8 * 15 and counts[3] >= 0.6 * 7 and counts[4] >= 0.9 * 20


def test_spreading_program_table(oracle, golden):
    import torch
    e = engine("QAM16", "R1_2")                                        # its LDPC code is the (648, 324) one the program uses
    g = golden("ref_tool_tables")
    modes, snrs = g["sp_modes"].tolist(), g["sp_snrs"].tolist()
    table = np.zeros((len(modes), len(snrs)), np.int32)
    for mi, m in enumerate(modes):
        cases = [(si, t, oracle.tool_spreading_case(float(snr), int(m), 1000 + t)) for si, snr in enumerate(snrs) for t in range(20)]
        for si, t, c in cases:
            assert crc32(c["frame"]) == g["sp_frame_crc"][mi, si, t], (m, si, t)
        X = dev(np.stack([c["frame"] for _, _, c in cases]))
        llr, st = e.mcdpsk_demod(X, 10, 1, m if m in (2, 4) else 1)    # processTraining -> setReference -> demodulateSoft (:96-107)
        soft = llr.cpu().numpy()
        assert soft.shape[1] == g["sp_n_soft"][mi] == 650
        for k, (si, t, c) in enumerate(cases):
            assert crc32(soft[k]) == g["sp_soft_crc"][mi, si, t], (m, si, t)
        # LDPCCodec::decode of 650 soft bits = LDPCDecoder::decodeSoft's multi-block branch (ldpc_decoder.cpp:305-406): the first
        # 648 at the decoder's default min-sum factor 0.75 for up to 80 iterations, hard decisions kept either way; the two
        # left-over soft bits as a zero-padded block of their own, whose verdict is what lastDecodeSuccess() then reports
        tail = torch.zeros((len(cases), 648), dtype=torch.float32, device=llr.device)
        tail[:, :2] = llr[:, 648:650]
        out, _, _ = e.ldpc_decode(llr[:, :648].contiguous(), 80, 0.75)
        _, ok_tail, _ = e.ldpc_decode(tail, 80, 0.75)
        out, ok_tail = out.cpu().numpy(), ok_tail.cpu().numpy()
        for k, (si, t, c) in enumerate(cases):
            ok = int(ok_tail[k])
            assert ok == g["sp_ok"][mi, si, t], (m, si, t)
            dec = out[k, :40] if ok else np.zeros(40, np.uint8)
            errs = int(np.unpackbits(dec ^ g["sp_tx"][t]).sum()) if ok else 0
            assert np.array_equal(dec, g["sp_decoded"][mi, si, t]) and errs == g["sp_bit_errors"][mi, si, t], (m, si, t)
            table[mi, si] += int(ok and errs == 0)                     # result.decoded && result.bit_errors == 0 (:199)
    assert np.array_equal(table, g["sp_tool_success_table"]), table   # the table the program printed
    assert (table[:, -1] == 20).all() and table[2].sum() > table[1].sum() > table[0].sum()


def test_chase_cache_program_scenarios(oracle, golden):
    import torch
    e = engine("QAM16", "R1_2")
    g = golden("ref_tool_tables")
    # test 1 (:90-150): store 1.0, store 2.0 -> count 2, LLR 3.0
    acc = torch.zeros((1, 648), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert e.chase_combine(acc, cnt, torch.full((1, 648), 1.0, device="cuda")).cpu().numpy()[0] == 1 and cnt.cpu().numpy()[0] == 1
    assert e.chase_combine(acc, cnt, torch.full((1, 648), 2.0, device="cuda")).cpu().numpy()[0] == 1 and cnt.cpu().numpy()[0] == 2
    assert (acc.cpu().numpy() == 3.0).all()
    # tests 2 and 3 (:154-262): the receptions, their sums through the combine kernel, LDPCCodec::decode of each
    l, _ = oracle.tool_chase_llrs()
    assert [crc32(v) for v in l] == g["chase_llr_crc"].tolist()
    L = dev(l)
    rows = []
    a2 = L[0:200:2].clone(); c2 = torch.ones(100, dtype=torch.int32, device="cuda")
    e.chase_combine(a2, c2, L[1:200:2].contiguous())
    b = L[200:].reshape(50, 4, 648)
    a4 = b[:, 0].clone(); c4 = torch.ones(50, dtype=torch.int32, device="cuda")
    e.chase_combine(a4, c4, b[:, 1].contiguous())
    s2 = a4.clone()
    e.chase_combine(a4, c4, b[:, 2].contiguous()); e.chase_combine(a4, c4, b[:, 3].contiguous())
    assert c2.cpu().numpy().tolist() == [2] * 100 and c4.cpu().numpy().tolist() == [4] * 50
    assert np.array_equal(bits(a2.cpu().numpy()), bits(l[0:200:2] + l[1:200:2]))
    lb = l[200:].reshape(50, 4, 648)
    assert np.array_equal(bits(a4.cpu().numpy()), bits(((lb[:, 0] + lb[:, 1]) + lb[:, 2]) + lb[:, 3]))
    dec = lambda rows: e.ldpc_decode(rows.contiguous(), 80, 0.75)[1].cpu().numpy()     # decoder default factor, 80 iterations
    ok = np.zeros(350, np.uint8)
    ok[0:200:2] = dec(L[0:200:2]); ok[1:200:2] = dec(a2)
    ok[200::3] = dec(b[:, 0]); ok[201::3] = dec(s2); ok[202::3] = dec(a4)
    assert np.array_equal(ok, g["chase_ok"]), np.nonzero(ok != g["chase_ok"])
    t2, t3 = ok[:200].reshape(100, 2).sum(0), ok[200:].reshape(50, 3).sum(0)
    assert [int(t2[0]), int(t2[1]), int(t3[0]), int(t3[1]), int(t3[2])] == g["chase_tool_counts"].tolist()
    assert t2[1] > t2[0] + 10 and t3[2] > t3[1] > t3[0]                # the program's PASS conditions (:194,:251)


def test_zc_dbpsk_program_chain(oracle, golden):
    """tools/test_zc_dbpsk.cpp: all 130 testAtSNR cases of its sweep and floor search through the device: ZC detection on the
    41 400-sample capture (the long-buffer path), then - exactly as the program chains them - the MC-DPSK demodulator from the
    ZC's start_sample with the ZC's CFO estimate and 65 data symbols, 648 soft bits, one LDPC decode at the decoder's default
    factor.  Cases the program abandons (no sync, start too late for a whole frame) are abandoned at the same stage."""
    import torch
    e = engine("QAM16", "R1_2")
    g = golden("ref_tool_tables")
    n = len(g["zcd_snr"])
    sigs = [oracle.tool_zc_dbpsk_case(float(g["zcd_snr"][i]), int(g["zcd_seed"][i]))["signal"] for i in range(n)]
    assert all(crc32(sigs[i]) == g["zcd_sig_crc"][i] for i in range(n)) and len({len(s) for s in sigs}) == 1
    X = dev(np.stack(sigs))
    r = e.sync_zc(X, threshold=0.2, root_mask=15)
    zc7 = np.stack([r[k].astype(np.float32) for k in ("detected", "frame_type", "start_sample", "correlation", "cfo_hz", "snr_estimate", "root_detected")], axis=1)
    assert np.array_equal(bits(zc7), bits(g["zcd_zc7"])), np.nonzero(bits(zc7) != bits(g["zcd_zc7"]))
    need, L = (8 + 1 + 65) * 512, X.shape[1]
    stage = np.zeros(n, np.int32)
    rows = []
    for i in range(n):
        if not r["detected"][i]:
            continue
        start = int(r["start_sample"][i])
        stage[i] = 1
        if start < 0 or start >= L - 1000:                              # test_zc_dbpsk.cpp:142-147
            continue
        stage[i] = 2
        if L - 500 - start < need:                                      # process() is not ready without training + reference + 65 symbols
            continue
        stage[i] = 4
        rows.append(i)
    frames = torch.stack([X[i, int(r["start_sample"][i]):int(r["start_sample"][i]) + need] for i in rows]).contiguous()
    llr, _ = e.mcdpsk_demod(frames, 10, 1, 1, cfo_hz=dev(r["cfo_hz"][rows].astype(np.float32)), phase0=dev(np.zeros(len(rows), np.float32)))
    soft = llr[:, :648].contiguous()
    out, ok, _ = e.ldpc_decode(soft, 80, 0.75)
    soft, out, ok = soft.cpu().numpy(), out.cpu().numpy(), ok.cpu().numpy()
    good = np.zeros(n, bool)
    for k, i in enumerate(rows):
        assert crc32(soft[k]) == g["zcd_soft_crc"][i], i
        assert int(ok[k]) == g["zcd_ok"][i], i
        if ok[k]:
            stage[i] = 5
            errs = int(np.unpackbits(out[k, :40] ^ g["zcd_tx"][i]).sum())
            assert np.array_equal(out[k, :40], g["zcd_decoded"][i]) and errs == g["zcd_bit_errors"][i], i
            good[i] = errs == 0
    assert np.array_equal(stage, g["zcd_stage"]), np.nonzero(stage != g["zcd_stage"])
    # the table the program printed: Sync% and Decode% per SNR, and the first step of its floor search
    assert r["detected"][:110].reshape(11, 10).sum(1).tolist() == g["zcd_tool_sync_counts"].tolist()
    assert good[:110].reshape(11, 10).sum(1).tolist() == g["zcd_tool_decode_counts"].tolist()
    assert int(good[110:].sum()) == int(g["zcd_tool_floor_m5_count"])
