"""Encoder feedback: aa_stream_lf_search = Encoder::apply_best_loopfilter_settings (encoder.cc:459-516) as one
batch.  For every candidate level L the expected raster is what the ORACLE decodes from the same frame written with level L and
zeroed filter adjustments (what the reference's search applies, encoder.cc:464-470); the expected score is the oracle's
restatement of x264's SSIM (oracle/ssim_x264.c; parity unpinned: libx264 is not in this image) within 1e-6; the choice follows
the reference's rule (ascending, stop at the first level that does not improve)."""
import numpy as np
import pytest

import alfalfa_amd as aa
import vp8_oracle as vo

pytestmark = pytest.mark.gpu

ZERO_DELTAS = dict(update=True, ref=[0, 0, 0, 0], mode=[0, 0, 0, 0])


def variant(w, h, seed, last_level, last_deltas, segmentation, sharpness):
    """key + 2 inter frames; only the LAST frame's loop-filter level / adjustments differ between variants."""
    import vp8_synth
    s = vp8_synth.SynthStream(w, h, seed)
    s.frame(key=True, lf_level=12, q_index=34, density=0.4, intra_bpred=0.4,
            lf_deltas=dict(update=True, ref=[2, None, -3, 1], mode=[4, -2, None, 3]))     # the state carries non-zero adjustments
    s.frame(key=False, lf_level=9, q_index=40, density=0.35, log2_parts=1, segmentation=segmentation)
    s.frame(key=False, lf_level=last_level, sharpness=sharpness, q_index=44, density=0.35, lf_deltas=last_deltas, refresh_golden=True,
            segmentation=dict(update_map=False) if segmentation else None)         # (enabled, map and levels persist)
    return s.frames


@pytest.mark.parametrize("w,h,seed,lo,hi,provisional,sharp,seg", [
    (175, 143, 61, 0, 5, 3, 0, None),
    (320, 176, 62, 22, 26, 40, 3, None),
    (96, 80, 63, 58, 63, 0, 7, dict(update_map=True, data=dict(absolute=False, quant=[0, 3, -4, 7], lf=[0, 5, -9, 14]), tree_probs=[120, 80, 200])),
    (64, 64, 64, 0, 63, 20, 0, None),
])
def test_lf_search_matches_the_oracle(gpu_ctx, w, h, seed, lo, hi, provisional, sharp, seg):
    given = variant(w, h, seed, provisional, dict(update=True, ref=[5, 5, None, -7], mode=[1, None, 2, -1]), seg, sharp)
    wants = {L: variant(w, h, seed, L, ZERO_DELTAS, seg, sharp) for L in range(lo, hi + 1)}
    for L in wants:
        assert wants[L][:2] == given[:2]                                    # same history, only the last frame differs
    dec = aa.Decoder(gpu_ctx, w, h)
    for fr in given[:2]:
        dec.get_frame_output(fr)
    state_before = dec.export_state()
    pw, ph = dec.padded_width, dec.padded_height
    rng = np.random.default_rng(seed)
    prev = np.frombuffer(dec.raster_bytes(1), np.uint8)[:pw * ph].reshape(ph, pw)
    original = np.clip(prev.astype(int) + rng.integers(-6, 7, prev.shape), 0, 255).astype(np.uint8)      # "the frame being encoded"

    best, best_q, qs, rasters = dec.lf_search(given[2], original, lo, hi, want_rasters=True)

    want_q = []
    for L in range(lo, hi + 1):
        ora = vo.OracleDecoder(w, h)
        for fr in wants[L]:
            ora.decode(fr)
        assert rasters[L - lo] == ora.raster_bytes(), "candidate level %d: %s" % (L, ora.frame_info())
        want_q.append(vo.ssim_plane(ora.raster_bytes()[:pw * ph], original.tobytes(), pw, ph))
    assert max(abs(a - b) for a, b in zip(qs, want_q)) <= 1e-6, (qs, want_q)
    exp_best, exp_q = lo, -1.0
    for L, q in zip(range(lo, hi + 1), qs):
        if q > exp_q:
            exp_best, exp_q = L, q
        else:
            break
    assert (best, best_q) == (exp_best, exp_q)
    # the decoder itself has not moved: it takes the frame re-written with the chosen level like any other
    assert dec.export_state() == state_before
    _, fi = dec.get_frame_output(wants[best][2])
    assert dec.raster_bytes(fi) == rasters[best - lo]


def test_lf_search_with_the_diagonal_schedule(gpu_ctx):
    """Without the row-pipelined filter every candidate is decoded by a scratch decoder of its own: same results."""
    gpu_ctx.set_schedule("diagonal")
    try:
        test_lf_search_matches_the_oracle(gpu_ctx, 175, 143, 65, 30, 33, 12, 2, None)
    finally:
        gpu_ctx.set_schedule("rows")


def test_lf_search_argument_errors(gpu_ctx):
    w, h = 64, 64
    frames = variant(w, h, 70, 10, None, None, 0)
    dec = aa.Decoder(gpu_ctx, w, h)
    dec.get_frame_output(frames[0])
    orig = np.zeros((dec.padded_height, dec.padded_width), np.uint8)
    with pytest.raises(aa.AlfalfaError):
        dec.lf_search(frames[1], orig, 5, 4)
    with pytest.raises(aa.AlfalfaError):
        dec.lf_search(frames[1], orig, 0, 64)
    with pytest.raises(aa.AlfalfaError) as e:
        dec.lf_search(frames[1][:7], orig, 0, 3)
    assert e.value.kind in ("Invalid", "Unsupported")
