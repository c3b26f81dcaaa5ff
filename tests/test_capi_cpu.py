"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol
include/ria_gpu.h declares.  No compute calls here (no GPU in the build container)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from ria_amd import capi
    L = capi.load()
    header = open(os.path.join(ROOT, "include", "ria_gpu.h")).read()
    declared = set(re.findall(r"\b(ria_(?:gpu|link)_[a-z0-9_]+)\s*\(", header))
    declared -= {"ria_gpu_config", "ria_gpu_geometry", "ria_gpu_handle"}
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.ria_gpu_abi_version() == 1


def test_struct_layouts_match_header():
    from ria_amd import capi
    assert C.sizeof(capi.Config) == 64
    assert C.sizeof(capi.Geometry) == 64
    assert C.sizeof(capi.FrameMeta) == 16
    assert C.sizeof(capi.FrameStatus) == 32
    assert C.sizeof(capi.DecodeStatus) == 20
    assert C.sizeof(capi.McdpskConfig) == 16
    assert C.sizeof(capi.LinkRecommendation) == 24


def test_create_rejects_bad_config_without_a_gpu():
    from ria_amd import capi
    L = capi.load()
    cfg = capi.Config()
    L.ria_gpu_default_config(C.byref(cfg))
    assert (cfg.modulation, cfg.code_rate, cfg.fft_size, cfg.num_carriers) == (6, 2, 1024, 59)
    h = C.c_void_p()
    cfg.abi_version = 99
    assert L.ria_gpu_create(C.byref(cfg), C.byref(h)) == -1 and not h
    L.ria_gpu_default_config(C.byref(cfg))
    cfg.fft_size = 512
    assert L.ria_gpu_create(C.byref(cfg), C.byref(h)) == -4 and not h


def test_product_never_imports_the_oracle():
    """The oracle is a checker only: nothing under ria_amd/ may reference it."""
    for d, _, files in os.walk(os.path.join(ROOT, "ria_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "pyoracle" not in src and "ria_oracle" not in src and "libria_ref" not in src, f


def test_link_adaptation_ladder_matches_reference_grid(golden):
    """recommendWaveformAndRate / recommendDataMode / selectOFDMCodeRate / capInitialOFDMRate
    (waveform_selection.hpp) over a dense (snr, fading) grid recorded from the reference; host scalars,
    so this runs without a GPU through the C ABI."""
    import ctypes as C
    import numpy as np
    from ria_amd import capi
    L = capi.load()
    g = golden("link_adaptation")
    o = capi.LinkRecommendation()
    for i, s in enumerate(g["snr"]):
        for j, f in enumerate(g["fading"]):
            L.ria_link_recommend(float(s), float(f), C.byref(o))
            got = [o.waveform, o.modulation, o.code_rate, o.spreading, o.num_carriers, o.estimated_throughput_bps]
            assert np.array_equal(np.array(got, np.float32), g["recommend"][i, j]), (s, f, got, g["recommend"][i, j])
            for w, wave in enumerate((4, 5)):
                L.ria_link_data_mode(float(s), wave, float(f), C.byref(o))
                got = [o.modulation, o.code_rate, o.spreading, o.num_carriers]
                assert np.array_equal(np.array(got, np.float32), g["data_mode"][i, j, w]), (s, f, wave, got)
            assert L.ria_link_ofdm_code_rate(float(s), float(f)) == g["rate"][i, j]
            assert L.ria_link_cap_initial_rate(float(s), float(f), 4) == g["cap"][i, j, 0]
            assert L.ria_link_cap_initial_rate(float(s), float(f), 3) == g["cap"][i, j, 1]


def test_host_built_cox_tables_match_reference_golden(golden, tmp_path):
    """The LTS passband templates and the Schmidl-Cox preamble the library builds on the host
    (ria_amd/csrc/host_tables.hpp: build_cox_template / build_cox_preamble) against the reference's
    (demodulator.cpp:108-141, modulator.cpp:479-532), bit for bit, without a GPU."""
    import subprocess
    import numpy as np
    exe = str(tmp_path / "htc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "helpers", "host_tables_check.cpp")])
    g = golden("cox_sync")
    for name, mod, rate in (("qam16_r12", 6, 2), ("dqpsk_r14", 2, 0)):
        out = str(tmp_path / f"{name}.f32")
        subprocess.check_call([exe, str(mod), str(rate), out])
        a = np.fromfile(out, np.float32)
        assert np.array_equal(a[:1152].view(np.uint32), g[f"tI_{name}"].view(np.uint32))
        assert np.array_equal(a[1152:2304].view(np.uint32), g[f"tQ_{name}"].view(np.uint32))
        assert np.array_equal(a[2305:].view(np.uint32), g[f"preamble_{name}"].view(np.uint32))


def test_shipped_decoder_layouts_are_valid(tmp_path):
    """ria_amd/csrc/core_layouts.inc (LDS layouts of the LDPC decoder annealed offline, tools/gen_core_layouts.cpp) must
    pass the structural validation against the generated H for all six rates, corrupted copies must be rejected, and
    the shipped layouts must be at least as good as what the create-time annealer reaches."""
    import subprocess
    exe = str(tmp_path / "layouts_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "helpers", "layouts_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split() for l in out.stdout.strip().split("\n")]
    assert len(rows) == 6
    for rate, ok, cost, floor, rej, fresh in rows:
        assert ok == "1" and fresh == "1" and rej.split("/")[0] == rej.split("/")[1]
        assert int(floor) <= int(cost) <= int(floor) * 1.6, (rate, cost, floor)
