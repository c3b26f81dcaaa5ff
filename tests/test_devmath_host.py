"""CPU: ria_amd/csrc/devmath.h (the float functions the kernels use) compiled for the host must be
bit-identical to glibc's sinf/cosf/logf/atan2f/hypotf — the libm the reference links against.
Sampled here (every 97th float + 20 M random pairs); the exhaustive run (stride 1) is 0 mismatches."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_devmath_matches_host_libm():
    src = os.path.join(ROOT, "tests", "helpers", "devmath_host_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "chk")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, src, "-lm"])
        out = subprocess.run([exe, "97"], capture_output=True, text=True, timeout=600)
        bad, n = map(int, out.stdout.split())
        assert n > 80_000_000
        assert bad == 0, f"{bad} of {n} results differ from libm"
