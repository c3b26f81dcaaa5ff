"""CPU: ria_amd/csrc/sort_exact.hpp (the introsort the GPU CRC recovery runs on one lane, and the host
recovery uses too) must leave the first 30 positions exactly as libstdc++'s std::sort does, equal keys
included, and its heapsort branch must equal std::partial_sort(first, last, last)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sort_exact_matches_std_sort():
    src = os.path.join(ROOT, "tests", "helpers", "sort_exact_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "chk")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "mismatches 0" in out.stdout
