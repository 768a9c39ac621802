"""Host logic on CPU (no GPU): the product's bitstream parser against the oracle, record by record,
and its persistent DecoderState (probability tables, segmentation, filter adjustments) frame by frame."""
import numpy as np
import pytest

import alfalfa_amd as aa
import vp8_oracle as vo
from conftest import GOLDEN, golden_frames
from parser_compare import compare


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_parser_matches_oracle(name):
    w, h, frames = golden_frames(name)
    p = aa.Parser(w, h)
    d = vo.OracleDecoder(w, h)
    d.set_phases(0)   # parse only
    for fr in frames:
        hdr, mb, cf = p.parse(fr)
        d.decode(fr)
        compare(hdr, mb, cf, d.macroblocks(), d.frame_info())
        assert (p.probs() == d.probs()).all()


def test_parser_error_types():
    w, h, frames = golden_frames("qcif_q30")
    p = aa.Parser(w, h)
    with pytest.raises(aa.AlfalfaError) as e:
        p.parse(b"\x00\x00")
    assert e.value.kind == "Invalid"
    bad = bytearray(frames[0]); bad[3] = 0
    with pytest.raises(aa.AlfalfaError) as e:
        p.parse(bytes(bad))
    assert e.value.kind == "Invalid" and "start code" in e.value.message
    with pytest.raises(aa.AlfalfaError) as e:
        aa.Parser(160, 144).parse(frames[0])
    assert e.value.kind == "Unsupported"
    v = bytearray(frames[1]); v[0] |= (2 << 1)       # version 2: only profile 0 is supported (uncompressed_chunk.cc:56-74)
    with pytest.raises(aa.AlfalfaError) as e:
        p.parse(bytes(v))
    assert e.value.kind == "Unsupported"
    # a failed parse must not have consumed decoder state: the real stream still parses
    for fr in frames:
        p.parse(fr)
