"""Feature coverage the reference encoder cannot provide (SPLITMV, golden/altref + sign bias, segmentation, 1..8 DCT
partitions, loop-filter deltas, hidden frames, probability updates, MVs far outside the frame, odd sizes) through
synthetic streams written by tools/vp8_synth.py.  CPU part: the writer's intent == what the oracle parses, the product
parser == the oracle, and (when oracle/_ref exists) the oracle == the live reference byte for byte."""
import os

import numpy as np
import pytest

import alfalfa_amd as aa
import vp8_oracle as vo
import vp8_synth
from parser_compare import compare

SIZES = [(96, 80), (33, 17), (64, 64), (175, 143), (16, 16), (200, 48)]
SEEDS = list(range(100, 118))


def check_intent(plans, om):
    mbh, mbw = om.shape
    for r in range(mbh):
        for c in range(mbw):
            pl, o = plans[r * mbw + c], om[r, c]
            assert o["y_mode"] == pl.y_mode
            if pl.inter:
                assert o["ref_frame"] == pl.ref
                assert [tuple(x) for x in o["mv"]] == [tuple(m) for m in pl.mvs]
            else:
                assert list(o["b_mode"]) == list(pl.b_modes)


# the streams the GPU suite decodes (tests/test_gpu_parity.py::test_hip_matches_oracle_on_synthetic_feature_streams: same seeds,
# sizes and frame counts): oracle == live reference on exactly those, so that HIP == oracle there is HIP == reference
GPU_SIZES = [(96, 80), (33, 17), (64, 64), (175, 143), (16, 16), (200, 48), (320, 176), (48, 256)]
GPU_SEEDS = list(range(200, 224))


def stream_of(seed):
    if seed >= 200:
        w, h = GPU_SIZES[seed % len(GPU_SIZES)]
        return w, h, vp8_synth.feature_stream(w, h, seed, 8)
    w, h = SIZES[seed % len(SIZES)]
    return w, h, vp8_synth.feature_stream(w, h, seed, 7)


@pytest.mark.parametrize("seed", SEEDS + GPU_SEEDS)
def test_synth_stream_cpu(seed, tmp_path):
    w, h, st = stream_of(seed)
    ref = None
    if vo.ref_available():
        path = str(tmp_path / "s.ivf"); vo.write_ivf(path, w, h, st.frames)
        vo.ref_decode(path, str(tmp_path / "s.raw"))
        ref = open(str(tmp_path / "s.raw"), "rb").read()
    pw, ph = (w + 15) // 16 * 16, (h + 15) // 16 * 16
    fs = pw * ph * 3 // 2
    ora, par = vo.OracleDecoder(w, h), aa.Parser(w, h)
    for i, fr in enumerate(st.frames):
        ora.decode(fr)
        om = ora.macroblocks()
        check_intent(st.intent[i], om)
        hdr, mb, cf = par.parse(fr)
        compare(hdr, mb, cf, om, ora.frame_info())
        assert (par.probs() == ora.probs()).all()
        if ref is not None:
            assert ora.raster_bytes() == ref[i * fs:(i + 1) * fs], "oracle differs from the reference at frame %d" % i


def test_synth_exercises_the_features():
    """The generator really produces what the docstring claims (guards against a silently narrow fuzzer)."""
    seen = {"split": set(), "refs": set(), "parts": set(), "hidden": 0, "seg": 0, "fadj": 0, "bpred_inter": 0, "far_mv": 0}
    for seed in SEEDS:
        w, h = SIZES[seed % len(SIZES)]
        st = vp8_synth.feature_stream(w, h, seed, 7)
        ora = vo.OracleDecoder(w, h)
        for fr in st.frames:
            shown = ora.decode(fr)
            info, om = ora.frame_info(), ora.macroblocks()
            seen["hidden"] += (not shown); seen["seg"] += info["segmentation_enabled"]; seen["fadj"] += info["filter_adjustments_enabled"]
            seen["parts"].add(info["num_partitions"])
            sp = om["y_mode"] == 9
            seen["split"].update(int(x) for x in om["split_partition"][sp])
            seen["refs"].update(int(x) for x in np.unique(om["ref_frame"]))
            if not info["key_frame"]:
                seen["bpred_inter"] += int(((om["y_mode"] == 4)).sum())
            seen["far_mv"] += int((np.abs(om["mv"].astype(int)) > 1000).sum())
    assert seen["split"] == {0, 1, 2, 3} and seen["refs"] == {0, 1, 2, 3} and seen["parts"] == {1, 2, 4, 8}
    assert seen["hidden"] and seen["seg"] and seen["fadj"] and seen["bpred_inter"] and seen["far_mv"]
