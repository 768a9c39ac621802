"""The oracle (oracle/vp8_oracle.c, our C restatement) against the reference's outputs.

golden.json holds, for every frame of every fixture stream, the SHA-256 of the three padded planes produced
by the REFERENCE decoder (oracle/_ref/ref_decode, see tests/golden/make_golden.py) and the SHA-1 of its
decode-to-stdout dump (the form src/tests/decoding.test:22-74 pins).  When oracle/_ref is present the
comparison is also done live, byte for byte."""
import hashlib
import os

import numpy as np
import pytest

import vp8_oracle as vo
from conftest import GOLDEN, GOLDEN_DIR, golden_frames, sha256


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_matches_reference_hashes(name):
    g = GOLDEN[name]
    w, h, frames = golden_frames(name)
    assert (w, h, len(frames)) == (g["width"], g["height"], g["frames"])
    d = vo.OracleDecoder(w, h)
    display = hashlib.sha1()
    for i, fr in enumerate(frames):
        shown = d.decode(fr)
        assert shown == bool(g["shown"][i])
        assert d.frame_info()["key_frame"] == g["key"][i]
        assert sha256(d.raster_bytes()) == g["raster_sha256"][i], "frame %d differs from the reference" % i
        if shown:
            y, u, v = d.planes()
            display.update(y[:h, :w].tobytes() + u[:(h + 1) // 2, :(w + 1) // 2].tobytes() + v[:(h + 1) // 2, :(w + 1) // 2].tobytes())
    assert display.hexdigest() == g["display_sha1"]


@pytest.mark.skipif(not vo.ref_available(), reason="oracle/_ref not built (reference sources absent)")
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_matches_reference_live(name, tmp_path):
    w, h, frames = golden_frames(name)
    raw = str(tmp_path / "ref.raw")
    vo.ref_decode(os.path.join(GOLDEN_DIR, name + ".ivf"), raw)
    ref = open(raw, "rb").read()
    pw, ph = (w + 15) // 16 * 16, (h + 15) // 16 * 16
    fs = pw * ph * 3 // 2
    d = vo.OracleDecoder(w, h)
    for i, fr in enumerate(frames):
        d.decode(fr)
        assert d.raster_bytes() == ref[i * fs:(i + 1) * fs], "frame %d" % i


def test_oracle_errors():
    d = vo.OracleDecoder(176, 144)
    with pytest.raises(vo.OracleError) as e:
        d.decode(b"\x00\x00")
    assert e.value.code == -1
    w, h, frames = golden_frames("qcif_q30")
    bad = bytearray(frames[0]); bad[3] = 0   # break the key-frame start code
    with pytest.raises(vo.OracleError):
        d.decode(bytes(bad))
    d2 = vo.OracleDecoder(160, 144)            # size mismatch -> Unsupported (uncompressed_chunk.cc:112-115)
    with pytest.raises(vo.OracleError) as e:
        d2.decode(frames[0])
    assert e.value.code == -2
