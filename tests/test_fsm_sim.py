"""The DEVICE parse algorithm replayed on the host (tests/cpp/fsm_sim.cc: the same parse_common.hh / tok_fsm.hh statements
the GPU lanes run, one lane at a time) against the product's host parser: every macroblock record and every coefficient
block byte for byte, on the golden streams, the synthetic feature streams (SPLITMV, golden/altref, segmentation with and
without map updates, 1-8 partitions, ...) and truncated frames.  CPU only; the GPU run of the same comparison is
tests/test_gpu_device_parse.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import alfalfa_amd as aa
from alfalfa_amd import capi
from conftest import GOLDEN, ROOT, golden_frames

BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
LIB = os.path.join(BUILD, "libfsm_sim.so")
CSRC = os.path.join(ROOT, "alfalfa_amd", "csrc")


def sim_lib():
    srcs = [os.path.join(ROOT, "tests", "cpp", "fsm_sim.cc"), os.path.join(CSRC, "parser.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("tok_fsm.hh", "coeff_pack.hh", "parse_common.hh", "parser.hh", "bool_reader.hh")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        tmp = "%s.%d.tmp" % (LIB, os.getpid())             # (pytest-xdist workers may build at the same time: rename is atomic)
        subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-Wall", "-Wextra", "-fPIC", "-shared"] + os.environ.get("AA_SIM_FLAGS", "").split() + srcs + ["-o", tmp], check=True)
        os.replace(tmp, LIB)
    L = C.CDLL(LIB)
    L.fsm_sim_create.restype = C.c_void_p
    L.fsm_sim_create.argtypes = [C.c_uint16, C.c_uint16]
    L.fsm_sim_destroy.argtypes = [C.c_void_p]
    L.fsm_sim_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(capi.FrameHeader), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.fsm_sim_segmap.argtypes = [C.c_void_p, C.c_void_p]
    L.fsm_sim_probs.argtypes = [C.c_void_p, C.c_void_p]
    L.fsm_sim_handover_check.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int]
    L.fsm_sim_set_pool_chunks.argtypes = [C.c_void_p, C.c_uint32]
    L.fsm_sim_last_chunks.argtypes = [C.c_void_p]
    L.fsm_sim_last_chunks.restype = C.c_uint32
    L.fsm_sim_set_packed.argtypes = [C.c_void_p, C.c_int]
    L.fsm_sim_last_words.argtypes = [C.c_void_p]
    L.fsm_sim_last_words.restype = C.c_uint32
    L.fsm_sim_body_create.restype = C.c_void_p
    L.fsm_sim_body_create.argtypes = [C.c_uint16, C.c_uint16]
    L.fsm_sim_body_destroy.argtypes = [C.c_void_p]
    L.fsm_sim_body_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.fsm_sim_body_guarded.argtypes = [C.c_uint16, C.c_uint16, C.c_char_p, C.c_size_t, C.c_int]
    return L


class Sim:
    def __init__(self, w, h, packed=False):
        self.L = sim_lib()
        self.h = self.L.fsm_sim_create(w, h)
        self.L.fsm_sim_set_packed(self.h, int(packed))
        self.mbw, self.mbh = (w + 15) // 16, (h + 15) // 16
        n = self.mbw * self.mbh
        self.mb = np.zeros(n, dtype=capi.MB_INFO_DTYPE)
        self.cf = np.zeros((n * 25 + 1) * 16, dtype=np.int16)

    def __del__(self):
        self.L.fsm_sim_destroy(self.h)

    def frame(self, data, expect=0):
        hdr, steps = capi.FrameHeader(), C.c_uint32()
        rc = self.L.fsm_sim_frame(self.h, data, len(data), C.byref(hdr), self.mb.ctypes.data, self.cf.ctypes.data, C.byref(steps))
        assert rc == expect, rc
        h = hdr.as_dict()
        return h, self.mb.copy(), self.cf[:h["num_coeff_blocks"] * 16].copy(), steps.value

    def segmap(self):
        out = np.zeros(self.mbw * self.mbh, np.uint8)
        self.L.fsm_sim_segmap(self.h, out.ctypes.data)
        return out

    def probs(self):
        out = np.zeros(1101, np.uint8)
        self.L.fsm_sim_probs(self.h, out.ctypes.data)
        return out


def check_stream(w, h, frames, packed=False):
    host, sim = aa.Parser(w, h), Sim(w, h, packed)
    for i, fr in enumerate(frames):
        hh, hmb, hcf = host.parse(fr)
        sh, smb, scf, steps = sim.frame(fr)
        assert sh == hh, (i, {k: (sh[k], hh[k]) for k in hh if sh[k] != hh[k]})
        a, b = smb.view(np.uint8).reshape(-1, 80), hmb.reshape(-1).view(np.uint8).reshape(-1, 80)
        if not (a == b).all():
            bad = np.nonzero((a != b).any(axis=1))[0]
            m = int(bad[0])
            raise AssertionError("frame %d: %d macroblock records differ, first mb %d (col %d row %d): device-algorithm %r host %r"
                                 % (i, len(bad), m, m % sim.mbw, m // sim.mbw, smb[m], hmb.reshape(-1)[m]))
        assert (scf.reshape(-1, 16) == hcf).all(), "frame %d: coefficient blocks differ" % i
        assert (sim.probs() == host.probs()).all()
        if hh["segmentation_enabled"]:
            assert (sim.segmap() == host.segmentation()["map"].reshape(-1)).all(), "frame %d: segment map" % i
        assert steps != 0xFFFFFFFF          # (the step bound of the frame size was not hit)


FORMATS = pytest.mark.parametrize("packed", [False, True], ids=["dense", "packed"])


@FORMATS
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_device_algorithm_matches_host_parser_on_goldens(name, packed):
    check_stream(*golden_frames(name), packed=packed)


@FORMATS
@pytest.mark.parametrize("seed", list(range(200, 264)))       # (200..223 are the seeds the GPU runs too)
def test_device_algorithm_matches_host_parser_on_synthetic_feature_streams(seed, packed):
    import vp8_synth
    sizes = [(96, 80), (33, 17), (64, 64), (175, 143), (16, 16), (200, 48), (320, 176), (48, 256)]
    w, h = sizes[seed % len(sizes)]
    check_stream(w, h, vp8_synth.feature_stream(w, h, seed, 8).frames, packed=packed)


@FORMATS
@pytest.mark.parametrize("name", ["qcif_q30_lf24", "w200_q40_lf63s7", "qcif_allkey_q20"])
def test_device_algorithm_matches_host_parser_on_truncated_frames(name, packed):
    from test_parser_vs_oracle import truncated
    w, h, frames = golden_frames(name)
    check_stream(w, h, truncated(frames), packed=packed)


@FORMATS
def test_device_algorithm_on_long_runs_of_skipped_macroblocks(packed):
    """A token lane learns about macroblocks through a 64-entry flag ring topped up every 64 steps; skipped macroblocks take no
    steps, so long runs of them outrun the ring and the lane has to wait for it (tok::macroblock_boundary)."""
    import vp8_synth
    for w, h, seed, density in ((1920, 48, 31, 0.0), (1920, 48, 32, 0.004), (640, 360, 33, 0.002), (4096, 16, 34, 0.0)):
        s = vp8_synth.SynthStream(w, h, seed)
        s.frame(key=True, q_index=30, skip_prob=3, density=density, skip_rate=1.0, intra_bpred=0.2)
        for k in range(3):
            s.frame(key=False, q_index=30, skip_prob=2 + k, density=density, skip_rate=1.0, log2_parts=k % 3, lf_level=8)
        check_stream(w, h, s.frames, packed=packed)


@FORMATS
def test_device_algorithm_on_extreme_geometries(packed):
    import vp8_synth
    for w, h, seed in ((16, 4096, 901), (4096, 16, 902), (24, 1000, 903), (2000, 32, 904)):
        check_stream(w, h, vp8_synth.feature_stream(w, h, seed, 3).frames, packed=packed)


def test_bool_decoder_hand_over_between_window_widths():
    """Parser::parse_header hands its 64-bit-window decoder over to a GPU lane's 32-bit one wherever the frame header happens
    to end -- including right when the host reader owes itself a refill, and past the end of the data."""
    rng = np.random.default_rng(7)
    L = sim_lib()
    for size in (1, 2, 3, 5, 9, 40, 300):
        data = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
        assert L.fsm_sim_handover_check(data, size, min(8 * size + 40, 1500), 64) == 0, size


@FORMATS
def test_device_algorithm_on_a_1080p_bench_stream(packed):
    """A stream of the benchmark workload (the reference encoder's output at 1920x1080)."""
    import workload
    if not workload.have_reference_tools():
        pytest.skip("oracle/_ref (stream generator) not built")
    check_stream(*aa.read_ivf(workload.make_stream("1080p_inter_lf", 3, 105)), packed=packed)


def test_coefficient_pool_runs_dry_and_the_frame_is_handed_back():
    """A token lane draws 64-KB chunks of the coefficient heap as it goes.  With fewer chunks in the pool than a frame needs the
    lane waits, then hands the frame back (TOK_NO_MEMORY: fsm_sim_frame -> 202) instead of hanging; with exactly as many as the
    frame took the parse goes through, and a frame takes only what its non-zero blocks need (far less than 25 per macroblock)."""
    w, h, frames = golden_frames("cif_q60_lf40s5")
    host, sim = aa.Parser(w, h), Sim(w, h)
    hh, hmb, hcf = host.parse(frames[0])
    sh, smb, scf, _ = sim.frame(frames[0])
    took = sim.L.fsm_sim_last_chunks(sim.h)
    nmb = sim.mbw * sim.mbh
    assert took >= 1 and took == -(-hh["num_coeff_blocks"] // 2048) or took == -(-hh["num_coeff_blocks"] // 2048) + 1, (took, hh["num_coeff_blocks"])
    assert took * 2048 < 25 * nmb or hh["num_coeff_blocks"] > 20 * nmb          # demand-sized, not worst-case sized
    sim2 = Sim(w, h)
    sim2.L.fsm_sim_set_pool_chunks(sim2.h, max(0, took - 1) or 10 ** 6)
    if took > 1:
        sim2.frame(frames[0], expect=202)
    sim3 = Sim(w, h)
    sim3.L.fsm_sim_set_pool_chunks(sim3.h, took)
    h3, mb3, cf3, _ = sim3.frame(frames[0])
    assert h3 == hh and (cf3 == hcf.reshape(-1)).all()


def body_check(w, h, frames):
    """-> frames checked (streams that switch segmentation on are not eligible for host lanes from there on)"""
    L = sim_lib()
    hnd = L.fsm_sim_body_create(w, h)
    checked = 0
    try:
        for i, fr in enumerate(frames):
            rc = L.fsm_sim_body_frame(hnd, fr, len(fr))
            assert rc in (0, 4, 5), (i, rc)
            checked += rc == 0
    finally:
        L.fsm_sim_body_destroy(hnd)
    return checked


def test_host_lanes_parse_a_frame_from_its_header_prepass_alone():
    """runtime.cpp's host lanes (AA_SUBMIT_HOST on a big call: host cores take frames the way GPU token lanes do) run
    Parser::parse_header at submit time and aa::parse_frame_body on a worker thread: together they must give Parser::parse's
    records -- goldens, the synthetic feature streams (multi-partition, SPLITMV, golden / altref), truncated frames."""
    import vp8_synth
    from test_parser_vs_oracle import truncated
    total = 0
    for name in sorted(GOLDEN):
        w, h, frames = golden_frames(name)
        total += body_check(w, h, frames)
        total += body_check(w, h, truncated(frames))
    sizes = [(96, 80), (33, 17), (64, 64), (175, 143), (16, 16), (200, 48), (320, 176), (48, 256)]
    for seed in range(200, 232):
        w, h = sizes[seed % len(sizes)]
        total += body_check(w, h, vp8_synth.feature_stream(w, h, seed, 8).frames)
    assert total >= 150, total


def test_a_decoder_that_has_run_out_of_partition_reads_no_memory():
    """BoolReader32::next_byte loaded the aligned word at its read position even when that position
    was past the partition's end -- a frame cut inside its first partition (accepted with error concealment on) keeps the
    header decoder consuming zeros for every macroblock that follows, and the loads walked off the end of the buffer: on the
    host-lane path off the end of the pinned arena.  Here the frame's last byte is the last byte in front of an inaccessible
    page; the parse runs in a forked child (100 = it died).  Frames cut inside the first partition, at its end, inside the DCT
    partitions, and whole."""
    L = sim_lib()
    ran = 0
    for name in ("qcif_q30_lf24", "qcif_allkey_q20", "w200_q40_lf63s7"):
        w, h, frames = golden_frames(name)
        for fr in frames[:3]:
            tag = fr[0] | (fr[1] << 8) | (fr[2] << 16)
            hdr = 3 if tag & 1 else 10
            first = (tag >> 5) & 0x7FFFF
            for cut, conceal in ((len(fr), 0), (hdr + first, 1), (hdr + first // 2, 1), (hdr + max(1, first // 8), 1), (hdr + first + 1, 0),
                                 (hdr + first + (len(fr) - hdr - first) // 2, 0)):
                rc = L.fsm_sim_body_guarded(w, h, fr[:cut], cut, conceal)
                assert rc in (0, 4, 5), (name, cut, len(fr), rc)
                ran += rc == 0
    assert ran >= 30, ran
