"""GPU test of the MC-DPSK plug-in adaptor (ria_amd/host/gpu_waveform.hpp: GpuMcDpskWaveform + robustDecodeSingleCW), built with
g++ against the C ABI only and driven in gui::StreamingDecoder's call order, against records taken from the reference's own
MCDPSKWaveform class (tests/golden/mcdpsk_waveform.npz) and the reference-recorded HARQ chain (tests/golden/harq_trials.npz)."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle as po
from test_gpu_parity import bits
from test_oracle_golden import mcwf_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def adaptor_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("adaptor") / "host_adaptor_test")
    lib = os.path.join(ROOT, "ria_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "helpers", "host_adaptor_test.cpp"),
                           "-L" + lib, "-lria_gpu", "-Wl,-rpath," + lib])
    return exe


def _fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a, np.float32).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_mcdpsk_adaptor_in_streaming_decoder_order_vs_reference_class(oracle, golden, adaptor_exe, tmp_path):
    n_dec = 0
    coded = ((np.arange(81) * 37 + 11) & 0xFF).astype(np.uint8)
    for i, (case, info, x, sync4, llr, aux5, dec, sizes) in enumerate(mcwf_cases(golden, oracle)):
        nc, mod, sp, data_sync, snr, cfo, known, lead, kind = case
        fin, out = str(tmp_path / f"mc{i}.f32"), str(tmp_path / f"mc{i}")
        x.tofile(fin)
        t = subprocess.check_output([adaptor_exe, str(mod), str(nc), fin, repr(float(known)), str(sp), out, "4", str(data_sync)]).decode().strip().split("\n")
        assert [int(v) for v in t[0].split()] == list(sizes), (case, t[0], sizes)
        for which, line in enumerate(t[1:3]):             # TX audio of the adaptor = the reference class's (restated by the oracle)
            exp = oracle.mcdpsk_wf_tx(nc, mod, po.R1_4, sp, which, coded)
            n, h = line.split()
            assert int(n) == len(exp) and int(h) == _fnv(exp), (case, which)
        s = t[3].split()
        got = np.array([float(s[0]), float(s[1]), float(s[2]), float(s[3])], np.float32)
        if not sync4[0]:
            assert int(s[0]) == 0 and got[2] == sync4[2], (case, t[3], sync4)
            assert len(t) == 4
            continue
        assert np.array_equal(bits(got), bits(sync4)), (case, got, sync4)
        p = t[4].split()
        assert int(p[0]) == int(aux5[0]) and int(p[1]) == len(llr) and int(p[5]) == int(aux5[4]), (case, t[4], aux5)
        assert np.array_equal(bits(np.array([float(p[2]), float(p[3]), float(p[4])], np.float32)), bits(aux5[1:4])), (case, t[4], aux5)
        assert np.array_equal(bits(np.fromfile(out + ".llr", np.float32)), bits(llr)), case
        d = t[5].split()
        assert [int(d[0]), int(d[1])] == [int(dec[0]), int(dec[2])], (case, t[5], dec)
        if dec[0]:
            assert np.array_equal(np.array(d[2:], np.uint8), dec[3:23].astype(np.uint8))
            n_dec += int(np.array_equal(np.array(d[2:], np.uint8), info[:20]))
    assert n_dec >= 7


def test_mcdpsk_adaptor_harq_chain_vs_reference_golden(oracle, golden, adaptor_exe, tmp_path):
    """A data codeword's receptions through process -> getSoftBits -> robustDecodeSingleCW -> chase combine -> robust decode of
    the sum (streaming_decoder.cpp:2758-2800) on the C++ class, for trials of the reference-recorded HARQ fixture that need
    1, 2, 3, 4 receptions or never decode."""
    from ria_amd import sweep
    g = golden("harq_trials")
    seed, n_trials = int(g["seed"]), int(g["trials"])
    seen = set()
    for ci in (0, 7, 9):
        nc, bps, sp, kind, snr = g["cases"][ci]
        nc, bps, sp, kind = int(nc), int(bps), int(sp), int(kind)
        trials = np.arange(n_trials)
        info = sweep.trial_payloads(seed, ci, trials, 16)
        seeds = np.stack([sweep.trial_seed32(seed, ci, t, trials) for t in range(4)], axis=1)
        tts = g[f"tx_to_success_{ci}"]
        picks = []
        for want in (1, 2, 3, 4, 0):
            idx = np.nonzero(tts == want)[0]
            picks += list(idx[:2])
        for q in picks:
            coded = oracle.ldpc_encode(po.R1_4, info[q])[:81]
            tx = oracle.mcdpsk_modulate(nc, bps, sp, coded)
            rx = np.concatenate([oracle.channel(kind, float(snr), int(seeds[q, t]), tx) for t in range(4)])
            fin = str(tmp_path / f"harq_{ci}_{q}.f32")
            rx.tofile(fin)
            mod = po.DBPSK if bps == 1 else po.DQPSK
            lines = subprocess.check_output([adaptor_exe, str(mod), str(nc), fin, "0", str(sp), "x", "5", "4"]).decode().strip().split("\n")
            last = lines[-1].split()
            exp_tts = int(tts[q])
            assert (int(last[0]) if int(last[1]) else 0) == exp_tts, (ci, q, lines, exp_tts)
            assert len(lines) == (exp_tts if exp_tts else 4)
            for t, line in enumerate(lines):
                f = line.split()
                assert [int(f[2]), int(f[3])] == list(g[f"tries_{ci}"][q, t]), (ci, q, t, line, g[f"tries_{ci}"][q, t])
                assert np.float32(float(f[4])) == g[f"fading_{ci}"][q, t]
            if exp_tts:
                assert np.array_equal(np.array(last[5:], np.uint8), g[f"decoded_{ci}"][q])
            seen.add(exp_tts)
    assert seen >= {1, 2, 3, 4}, seen
