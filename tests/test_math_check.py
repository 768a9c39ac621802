"""The packed (2 x int16) loop-filter arithmetic and the byte shuffles of alfalfa_amd/csrc/vp8_math.hh, compiled for the
host, against the scalar functions of the same header (tests/cpp/math_check.cc)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_loopfilter_math_matches_scalar(tmp_path):
    exe = str(tmp_path / "math_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "math_check.cc")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK")
