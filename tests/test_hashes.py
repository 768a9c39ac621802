"""DecoderState::hash / raster hashes / Decoder::minihash against values computed by the reference itself
(tests/golden/hash_golden.json, written by tests/golden/make_hash_golden.py from oracle/_ref/ref_hash).
Parity caveat: oracle/_ref hashes with the pre-1.81 boost::hash_combine formula (boost itself is not in the image)."""
import json
import os

import pytest

import alfalfa_amd as aa
from conftest import GOLDEN_DIR, golden_frames

HASHES = json.load(open(os.path.join(GOLDEN_DIR, "hash_golden.json")))


@pytest.mark.parametrize("name", sorted(HASHES))
def test_decoder_state_hash_matches_the_reference(name):
    """CPU: the persistent state after every frame (probability tables, segmentation incl. its pixel-sized map, filter
    adjustments incl. the reference's empty second range) hashes to the reference's value."""
    w, h, frames = golden_frames(name)
    p = aa.Parser(w, h)
    for i, fr in enumerate(frames):
        p.parse(fr)
        assert p.state_hash() == HASHES[name]["state"][i], (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(HASHES))
def test_decoder_hash_and_minihash_match_the_reference(gpu_ctx, name):
    w, h, frames = golden_frames(name)
    g = HASHES[name]
    a, b = aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h)
    for i, fr in enumerate(frames):
        a.get_frame_output(fr)                                   # host parser
        fi = gpu_ctx.submit_frames([(b, fr)], route="device")[0]; gpu_ctx.decode_batch([b], [fi])      # GPU parser
        for d in (a, b):
            parts, whole = d.decoder_hash()
            assert parts == [g["state"][i], g["last"][i], g["golden"][i], g["alternative"][i]], (name, i)
            assert whole == g["hash"][i] and d.minihash() == g["minihash"][i], (name, i)
