"""A WAVE of token workers replayed on the host (tests/cpp/wave_sim.cc): many lanes in lock step over the statements the GPU
lanes run (alfalfa_amd/csrc/tok_fsm.hh), each at its own place in its own frame of its own stream, sharing the tables, the job
queue and ONE coefficient pool -- against the product's host parser, record by record and block by block.  tests/test_fsm_sim.py
checks the lane's algorithm one lane at a time; this checks what only shows with several: a macroblock-boundary pass while
wave-mates decode, lanes waiting for a chunk while others finish and give theirs back, frames handed back for lack of memory
and run again, lanes reused for frame after frame, jobs arriving while the wave is busy.  Both coefficient formats.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import alfalfa_amd as aa
from alfalfa_amd import capi
from conftest import ROOT, golden_frames

BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
LIB = os.path.join(BUILD, "libwave_sim.so")
CSRC = os.path.join(ROOT, "alfalfa_amd", "csrc")


def wave_lib():
    srcs = [os.path.join(ROOT, "tests", "cpp", "wave_sim.cc"), os.path.join(CSRC, "parser.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("tok_fsm.hh", "coeff_pack.hh", "parse_common.hh", "parser.hh", "bool_reader.hh")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-Wall", "-Wextra", "-fPIC", "-shared"] + srcs + ["-o", LIB], check=True)
    L = C.CDLL(LIB)
    L.wave_sim_run.argtypes = [C.c_uint16, C.c_uint16, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t),
                               C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                               C.POINTER(capi.FrameHeader), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    return L


def run_wave(w, h, streams, lanes, pool_chunks=0, packed=False, seed=1, burst=1000, burst_gap=1):
    """streams: list of lists of frames (bytes) of one size -> stats; asserts every frame's records equal the host parser's"""
    L = wave_lib()
    flat = [fr for st in streams for fr in st]
    n = len(flat)
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    counts = (C.c_int * len(streams))(*[len(st) for st in streams])
    data = (C.c_char_p * n)(*flat)
    sizes = (C.c_size_t * n)(*[len(f) for f in flat])
    hdrs = (capi.FrameHeader * n)()
    mbs = np.zeros(n * nmb, dtype=capi.MB_INFO_DTYPE)
    cfs = np.zeros((n, 25 * nmb, 16), dtype=np.int16)
    stats = (C.c_uint64 * 8)()
    rc = L.wave_sim_run(w, h, len(streams), counts, data, sizes, lanes, pool_chunks, int(packed), seed, burst, burst_gap,
                        hdrs, mbs.ctypes.data, cfs.ctypes.data, stats)
    assert rc == 0, rc
    k = 0
    for s, st in enumerate(streams):
        host = aa.Parser(w, h)
        for f, fr in enumerate(st):
            hh, hmb, hcf = host.parse(fr)
            got = hdrs[k].as_dict()
            assert got == hh, (s, f, {x: (got[x], hh[x]) for x in hh if got[x] != hh[x]})
            a = mbs[k * nmb:(k + 1) * nmb].view(np.uint8).reshape(-1, 80)
            b = hmb.reshape(-1).view(np.uint8).reshape(-1, 80)
            assert (a == b).all(), "stream %d frame %d: macroblock records differ (first mb %d)" % (s, f, int(np.nonzero((a != b).any(axis=1))[0][0]))
            assert (cfs[k, :hh["num_coeff_blocks"]] == hcf).all(), "stream %d frame %d: coefficient blocks differ" % (s, f)
            k += 1
    return {"periods": stats[0], "boundary_passes": stats[1], "handed_back": stats[2], "peak_chunks_out": stats[3], "peak_busy_lanes": stats[4]}


def qcif_streams(n_synth):
    import vp8_synth
    out = [golden_frames(name)[2] for name in ("qcif_q30", "qcif_q30_lf24", "qcif_allkey_q20")]
    out += [vp8_synth.feature_stream(176, 144, 500 + k, 6).frames for k in range(n_synth)]
    return out


FORMATS = pytest.mark.parametrize("packed", [False, True], ids=["dense", "packed"])


@FORMATS
@pytest.mark.parametrize("lanes", [1, 5, 22, 64])
def test_a_wave_of_lanes_matches_the_host_parser(lanes, packed):
    """goldens + synthetic feature streams (SPLITMV, golden / altref, segmentation with and without map updates, 1-8 partitions)
    of one size, their frames interleaved in the queue; the lanes are reused frame after frame"""
    st = run_wave(176, 144, qcif_streams(9), lanes, packed=packed, seed=lanes)
    assert st["peak_busy_lanes"] == min(lanes, sum(len(s) for s in qcif_streams(9)))
    assert st["boundary_passes"] > 0


@FORMATS
def test_jobs_arriving_while_the_wave_is_busy(packed):
    """three jobs every 40 periods: lanes finish, idle, and pick up frames that arrive later -- beside lanes in mid-frame"""
    run_wave(176, 144, qcif_streams(5), 8, packed=packed, seed=3, burst=3, burst_gap=40)


@FORMATS
def test_a_scarce_pool_lanes_wait_frames_are_handed_back_and_run_again(packed):
    """One pool for the wave, with chunks for two or three frames while 16 lanes want one or more each: lanes wait at macroblock
    boundaries (their wave-mates keep decoding), take the chunks that finished frames give back, or hand their frame back after
    the time limit (TOK_NO_MEMORY) and get it again later.  Every chunk comes back exactly once (checked in the harness)."""
    import vp8_synth
    w, h = 320, 176
    streams = [vp8_synth.feature_stream(w, h, 700 + k, 5).frames for k in range(8)]
    plenty = run_wave(w, h, streams, 16, packed=packed, seed=9)
    assert plenty["handed_back"] == 0
    need = max(2, plenty["peak_chunks_out"] // 16 + 1)          # what one frame takes at most, roughly
    scarce = run_wave(w, h, streams, 16, pool_chunks=need + 1, packed=packed, seed=9)
    assert scarce["peak_chunks_out"] <= need + 1
    assert scarce["periods"] > plenty["periods"]                 # lanes stood still for lack of memory
    assert scarce["handed_back"] > 0                             # ... and some gave their frame back and ran it again


@FORMATS
def test_a_wave_on_multi_chunk_frames(packed):
    """frames big enough for several 64-KB chunks each (CIF key frames at a low quantiser), so that lanes switch chunks in mid-frame
    while their wave-mates draw from the same pool"""
    w, h, frames = golden_frames("cif_q60_lf40s5")
    st = run_wave(w, h, [frames[:3]] * 6, 6, packed=packed, seed=2)
    assert st["peak_chunks_out"] >= 6
