"""Per-stage parity on the GPU: the device functions the reconstruction kernels are assembled from (alfalfa_amd/csrc/vp8_math.hh,
recon_inl.hh), run one stage at a time by the test kernels of tests/cpp/stage_kernels.hip on random inputs, against the ORACLE's
function for that stage (oracle/vp8_oracle.c vp8o_stage_*: thin wrappers around the functions the whole-frame oracle uses, each
citing the reference lines it follows).  The raster tests say WHICH macroblock differs; these say which stage:

    dequantise -> 4x4 inverse DCT -> add to the prediction     quantization.cc:95-126, transform.cc:100-137  (dense and packed blocks)
    dequantise -> inverse Walsh-Hadamard of Y2                 transform.cc:47-88
    the ten 4x4 intra predictors (table and switch form)       prediction.cc:469-618
    the 16x16 / 8x8 intra predictors                           prediction.cc:385-431
    one six-tap pass on packed bytes, all fractions / offsets  prediction.cc:645-653, 861-915
    the normal loop filter, macroblock and sub-block edges     loopfilter_filters.hh:50-183, loopfilter.cc:81-125

Inputs include the extremes (coefficients and quantisers whose product wraps int16 -- quirk Q4 --, saturating residuals, flat and
noisy pixel rows)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import vp8_oracle as vo
from conftest import ROOT

pytestmark = pytest.mark.gpu

BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
LIB = os.path.join(BUILD, "libstage_test.so")
SRC = os.path.join(ROOT, "tests", "cpp", "stage_kernels.hip")
CSRC = os.path.join(ROOT, "alfalfa_amd", "csrc")


def build_stage_lib():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("recon_inl.hh", "vp8_math.hh", "coeff_pack.hh", "tok_fsm.hh", "parse_common.hh")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", SRC, "-o", tmp], check=True)
        os.replace(tmp, LIB)
    return LIB


@pytest.fixture(scope="module")
def stage():
    L = C.CDLL(build_stage_lib())
    assert L.stage_device_count() > 0, "no HIP device: the stage kernels have nothing to run on"
    return L


@pytest.fixture(scope="module")
def ora():
    L = vo.lib()
    L.vp8o_stage_sixtap.restype = C.c_int
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def random_coefficients(rng, n):
    """Blocks of quantised coefficients like a stream's (few, small) mixed with blocks that are full and large, and quantisers
    up to 157 * 155 / 100 -- products that leave int16 included (the reference wraps: quantization.cc:110-121)."""
    coeff = np.zeros((n, 16), np.int16)
    for i in range(n):
        kind = i % 4
        if kind == 0:
            k = rng.integers(1, 5)
            coeff[i, rng.choice(16, k, replace=False)] = rng.integers(-12, 13, k)
        elif kind == 1:
            coeff[i] = rng.integers(-300, 301, 16)
        elif kind == 2:
            coeff[i] = rng.integers(-2048, 2048, 16)
        else:
            coeff[i, 0] = rng.integers(-2048, 2048)
    q = np.stack([rng.integers(4, 158, n), rng.integers(4, 244, n)], axis=1).astype(np.int32)
    q[::7] = (157, 243)
    return coeff, q


def test_dequantise_idct_add_dense_and_packed(stage, ora):
    rng = np.random.default_rng(61)
    n = 4096
    coeff, q = random_coefficients(rng, n)
    pred = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    pred[::5] = 255; pred[1::5] = 0
    want = pred.copy()
    for i in range(n):
        ora.vp8o_stage_residual(ptr(coeff[i]), int(q[i, 0]), int(q[i, 1]), ptr(want[i]))
    got = np.zeros_like(pred)
    assert stage.stage_residual(n, ptr(coeff), ptr(q), ptr(pred), ptr(got)) == 0
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "dequant + IDCT + add: block %d coeff %r q %r pred %r: got %r want %r" % (bad[0], coeff[bad[0]], q[bad[0]], pred[bad[0]], got[bad[0]], want[bad[0]])
    # the same blocks from packed storage: a mask word (bit k: zigzag position k) + the values in zigzag order (coeff_pack.hh)
    zigzag = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
    masks, firsts, values = np.zeros(n, np.uint32), np.zeros(n, np.uint32), []
    for i in range(n):
        firsts[i] = len(values)
        for k in range(16):
            if coeff[i, zigzag[k]]:
                masks[i] |= 1 << k
                values.append(coeff[i, zigzag[k]])
    values = np.array(values + [0] * 16, np.int16)
    got = np.zeros_like(pred)
    assert stage.stage_residual_packed(n, ptr(masks), ptr(firsts), ptr(values), len(values), ptr(q), ptr(pred), ptr(got)) == 0
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "packed block read + IDCT: block %d mask %x: got %r want %r" % (bad[0], masks[bad[0]], got[bad[0]], want[bad[0]])


def test_dequantise_inverse_walsh_hadamard(stage, ora):
    rng = np.random.default_rng(62)
    n = 4096
    coeff, q = random_coefficients(rng, n)
    q[:, 0] = np.minimum(q[:, 0] * 2, 314)                       # y2_dc = 2 x, y2_ac = 155 / 100 x (quantization.cc:83-93)
    want = np.zeros((n, 16), np.int16)
    for i in range(n):
        ora.vp8o_stage_iwht(ptr(coeff[i]), int(q[i, 0]), int(q[i, 1]), ptr(want[i]))
    got = np.zeros_like(want)
    assert stage.stage_iwht(n, ptr(coeff), ptr(q), ptr(got)) == 0
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "dequant + iWHT: block %d: got %r want %r" % (bad[0], got[bad[0]], want[bad[0]])


def test_the_ten_4x4_intra_predictors(stage, ora):
    """E[0..3] = left[3..0], E[4] = above-left, E[5..12] = above[0..7] (vp8_math.hh); the oracle predicts the block at (4, 4) of a
    16-wide plane, whose above-right pixels are the four to the right of the row above (prediction.cc:140-164, the plain case)."""
    rng = np.random.default_rng(63)
    n = 10 * 200
    mode = np.repeat(np.arange(10, dtype=np.uint8), 200)
    E = np.zeros((n, 13), np.uint8)
    want = np.zeros((n, 16), np.uint8)
    for i in range(n):
        plane = rng.integers(0, 256, (16, 16), dtype=np.uint8)
        if i % 9 == 0:
            plane[:] = rng.integers(0, 256)
        for k in range(4):
            E[i, k] = plane[4 + 3 - k, 3]
        E[i, 4] = plane[3, 3]
        E[i, 5:13] = plane[3, 4:12]
        ora.vp8o_stage_predict(ptr(plane), 16, 4, 4, 4, int(mode[i]))
        want[i] = plane[4:8, 4:8].reshape(-1)
    table, switch = np.zeros_like(want), np.zeros_like(want)
    assert stage.stage_bpred(n, ptr(mode), ptr(E), ptr(table), ptr(switch)) == 0
    for name, got in (("table form (bpred_entry / bpred_eval)", table), ("switch form (bpred_pixel)", switch)):
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, "4x4 predictor %s, mode %d: E %r: got %r want %r" % (name, mode[bad[0]], E[bad[0]], got[bad[0]], want[bad[0]])


@pytest.mark.parametrize("size", [16, 8])
def test_the_16x16_and_8x8_intra_predictors(stage, ora, size):
    rng = np.random.default_rng(64 + size)
    n = 4 * 60
    mode = np.repeat(np.arange(4, dtype=np.uint8), 60)
    above, left, corner = np.zeros((n, size), np.uint8), np.zeros((n, size), np.uint8), np.zeros(n, np.uint8)
    want = np.zeros((n, size * size), np.uint8)
    for i in range(n):
        plane = rng.integers(0, 256, (48, 48), dtype=np.uint8)
        if i % 7 == 0:
            plane[:] = rng.integers(0, 256)
        above[i], left[i], corner[i] = plane[15, 16:16 + size], plane[16:16 + size, 15], plane[15, 15]
        ora.vp8o_stage_predict(ptr(plane), 48, 16, 16, size, int(mode[i]))
        want[i] = plane[16:16 + size, 16:16 + size].reshape(-1)
    got = np.zeros_like(want)
    assert stage.stage_bigpred(n, size, ptr(mode), ptr(above), ptr(left), ptr(corner), ptr(got)) == 0
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "%dx%d predictor mode %d: first differing block %d" % (size, size, mode[bad[0]], bad[0])


def test_one_six_tap_pass_on_packed_bytes(stage, ora):
    """sixtap_x4_lane: four outputs from twelve source bytes, outputs k = bytes o+k .. o+k+5, every fraction (0 = the identity the
    reference also runs, prediction.cc:875-881) and every offset 0..3; extremes 0 / 255 included (the clamp after each pass: Q6)."""
    rng = np.random.default_rng(66)
    n = 8 * 4 * 400
    frac = np.tile(np.repeat(np.arange(8, dtype=np.uint8), 4), 400)
    off = np.tile(np.arange(4, dtype=np.uint8), 8 * 400)
    src = rng.integers(0, 256, (n, 12), dtype=np.uint8)
    src[::3] = rng.choice(np.array([0, 255], np.uint8), (len(src[::3]), 12))
    want = np.zeros((n, 4), np.uint8)
    for i in range(n):
        for k in range(4):
            want[i, k] = ora.vp8o_stage_sixtap(ptr(np.ascontiguousarray(src[i, off[i] + k:off[i] + k + 6])), int(frac[i]))
    got = np.zeros(n, np.uint32)
    assert stage.stage_sixtap(n, ptr(np.ascontiguousarray(src).view(np.uint32)), ptr(off), ptr(frac), ptr(got)) == 0
    got = got.view(np.uint8).reshape(n, 4)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, "six-tap pass frac %d offset %d src %r: got %r want %r" % (frac[bad[0]], off[bad[0]], src[bad[0]], got[bad[0]], want[bad[0]])


def test_the_loop_filter_edges(stage, ora):
    """lf_edge_pk (two positions per lane in packed int16) with the limits lf_params derives from (level, sharpness, key frame),
    against filter_edge with the limits of the oracle's NormalLoopFilter -- macroblock edges and sub-block edges."""
    rng = np.random.default_rng(67)
    cases = [(lv, sh, key, mb) for lv in (1, 2, 5, 14, 15, 19, 20, 24, 39, 40, 63) for sh in (0, 1, 4, 5, 7) for key in (0, 1) for mb in (0, 1)]
    per = 60
    n = len(cases) * per
    level, sharp, key, mb = (np.repeat(np.array([c[k] for c in cases], np.uint8), per) for k in range(4))
    px = np.zeros((n, 2, 8), np.uint8)
    for i in range(n):
        for h in range(2):
            spread = 1 << rng.integers(0, 9)
            base = rng.integers(0, 256)
            px[i, h] = np.clip(base + rng.integers(-spread, spread + 1, 8), 0, 255)
    want = px.copy()
    lim = (C.c_int * 4)()
    for i in range(n):
        ora.vp8o_stage_filter_limits(int(level[i]), int(sharp[i]), int(key[i]), lim)
        for h in range(2):
            ora.vp8o_stage_filter_edge(ptr(want[i, h]), int(mb[i]), lim[0], lim[1] if mb[i] else lim[2], lim[3])
    got = np.zeros_like(px)
    assert stage.stage_lf_edge(n, ptr(px), ptr(level), ptr(sharp), ptr(key), ptr(mb), ptr(got)) == 0
    bad = np.nonzero((got != want).reshape(n, -1).any(axis=1))[0]
    assert len(bad) == 0, "loop filter level %d sharpness %d key %d mb_edge %d px %r: got %r want %r" % (
        level[bad[0]], sharp[bad[0]], key[bad[0]], mb[bad[0]], px[bad[0]], got[bad[0]], want[bad[0]])
