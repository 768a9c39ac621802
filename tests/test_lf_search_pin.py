"""Pins the ALGORITHM of aa_stream_lf_search against the reference ENCODER itself, on the CPU.

oracle/_ref/xc-enc-ssim is the reference encoder compiled in place with ssim() answered by our restatement of x264's SSIM
(the only piece of Encoder::apply_best_loopfilter_settings that lives outside /root/reference).  It runs the reference's own
search (encoder.cc:459-516) and writes the level it chose into each frame header.  Here the same choice is recomputed the way
aa_stream_lf_search defines it -- candidates = the frame decoded with level L and zero adjustments (the ORACLE decodes variants
re-serialised by the reference's own serialiser, oracle/_ref/ref_rewrite), scored by the restated SSIM of the padded luma plane
against the edge-extended original (input/yuv4mpeg.cc:231-271), ascending levels, stop at the first that does not improve --
and must land on the encoder's level for every frame.  tests/test_gpu_lf_search.py then ties the GPU implementation to exactly
these oracle computations."""
import os
import subprocess

import numpy as np
import pytest

import alfalfa_amd as aa
import vp8_oracle as vo
from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref")
ENC, REWRITE = os.path.join(REF, "xc-enc-ssim"), os.path.join(REF, "ref_rewrite")
pytestmark = pytest.mark.skipif(not (os.path.exists(ENC) and os.path.exists(REWRITE)), reason="oracle/_ref (the reference built in place) is not here")


def padded_luma(y, pw, ph):
    """edge_extend_component (input/yuv4mpeg.cc:231-264): right, bottom, corner."""
    h, w = y.shape
    out = np.empty((ph, pw), np.uint8)
    out[:h, :w] = y
    out[:h, w:] = y[:, w - 1:w]
    out[h:, :w] = y[h - 1:h, :]
    out[h:, w:] = y[h - 1, w - 1]
    return out


@pytest.mark.parametrize("w,h,entropy,qi,seed,nframes", [(176, 144, "low", 100, 7, 4), (176, 144, "low", 127, 8, 3), (176, 144, "high", 127, 9, 4),
                                                        (175, 143, "low", 100, 10, 1), (64, 48, "low", 110, 11, 5)])
def test_search_as_defined_here_reproduces_the_reference_encoders_choice(tmp_path, w, h, entropy, qi, seed, nframes):
    import make_y4m
    y4m, ivf = str(tmp_path / "in.y4m"), str(tmp_path / "enc.ivf")
    make_y4m.write_y4m(y4m, w, h, nframes, seed, entropy)
    subprocess.run([ENC, "-i", "y4m", "-y", str(qi), "-o", ivf, y4m], check=True, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    _, _, frames = aa.read_ivf(ivf)
    assert len(frames) >= 1
    originals = [planes[0] for planes in make_y4m.synth_frames(w, h, nframes, seed, entropy)]
    pw, ph = (w + 15) // 16 * 16, (h + 15) // 16 * 16
    parser = aa.Parser(w, h)
    chosen = []
    for fr in frames:
        hdr, _, _ = parser.parse(fr)
        chosen.append(hdr["loop_filter_level"])
        assert hdr["filter_adjustments_enabled"] and parser.filter_adjustments()["ref"] == [0] * 4 and parser.filter_adjustments()["mode"] == [0] * 4
        assert not hdr["segmentation_enabled"]
    nonzero = 0
    for k, want in enumerate(chosen):
        original = padded_luma(np.asarray(originals[k]).reshape(h, w), pw, ph).tobytes()
        best, best_q = 0, -1.0
        for level in range(64):                          # Encoder::loop_filter_level_ is only set in real-time mode: 0..63 every frame
            var = str(tmp_path / "var.ivf")
            subprocess.run([REWRITE, ivf, var, str(level), "-1", str(k)], check=True)
            ora = vo.OracleDecoder(w, h)
            for fr in aa.read_ivf(var)[2][:k + 1]:
                ora.decode(fr)
            q = vo.ssim_plane(ora.raster_bytes()[:pw * ph], original, pw, ph)
            if q > best_q:
                best, best_q = level, q
            else:
                break
        assert best == want, "frame %d: the reference encoder wrote level %d, the search as defined here finds %d" % (k, want, best)
        nonzero += want != 0
    assert nonzero or entropy == "high", chosen       # (the case is only interesting if the encoder really filtered)
