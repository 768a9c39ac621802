"""Builds the in-tree native library (HIP kernels + host parser + C ABI) for gfx950 with hipcc.

    python -m alfalfa_amd.build          # -> alfalfa_amd/lib/libalfalfa_amd.so

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libalfalfa_amd.so")
SOURCES = ["parser.cpp", "runtime.cpp", "kernels.hip", "parse_kernels.hip"]
HEADERS = ["parser.hh", "bool_reader.hh", "parse_common.hh", "tok_fsm.hh", "coeff_pack.hh", "recon_inl.hh", "runtime_types.inc", "runtime_pool.inc", "runtime_tokens.inc", "runtime_records.inc", "runtime_ctx.inc", "runtime_submit.inc", "runtime_decode.inc", "runtime_rasters.inc", "runtime_lf_search.inc", "vp8_tables.h", "vp8_math.hh", "device_types.h",
           os.path.join("..", "..", "include", "alfalfa_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC] + FLAGS + os.environ.get("AA_EXTRA_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
