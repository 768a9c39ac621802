"""The C++ mirror of the reference's class API (include/alfalfa_amd/alfalfa.hh) over the C ABI: our counterpart of the
reference's own golden test (src/tests/decoding.test: sha1sum of `decode-to-stdout FILE`), built with plain g++."""
import hashlib
import os
import subprocess

import pytest

from conftest import GOLDEN, GOLDEN_DIR, ROOT

BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
EXE = os.path.join(BUILD, "decode_to_stdout")


def build_exe():
    from alfalfa_amd import build as b
    b.build()
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "decode_to_stdout.cc")
    hdr = os.path.join(ROOT, "include", "alfalfa_amd", "alfalfa.hh")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(b.LIB)):
        libdir = os.path.dirname(b.LIB)
        subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE,
                        "-L" + libdir, "-lalfalfa_amd", "-Wl,-rpath," + libdir], check=True)
    return EXE


def test_cpp_mirror_builds_and_fails_loudly_without_gpu():
    from alfalfa_amd import capi
    exe = build_exe()
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "qcif_q30.ivf")], capture_output=True)
    if capi.device_count() == 0:
        assert r.returncode != 0 and b"no HIP device" in r.stderr and r.stdout == b""
    r = subprocess.run([exe, "/nonexistent.ivf"], capture_output=True)
    assert r.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_decode_to_stdout_sha1_matches_reference(name):
    exe = build_exe()
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, name + ".ivf")], capture_output=True, check=True)
    assert hashlib.sha1(r.stdout).hexdigest() == GOLDEN[name]["display_sha1"]
