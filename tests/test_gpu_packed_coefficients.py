"""Packed coefficient storage (aa_ctx_set_packed_coefficients; tok_fsm.hh "Packed coefficients", coeff_pack.hh) on a real
MI355X: a context whose token lanes store a mask word + the non-zero values of every block, read in that form by the
reconstruction kernels (round 6; rounds 3-5 expanded them into a transient dense array first), must give byte for byte what the dense path gives --
records against the host parser, rasters against the oracle and the committed reference hashes.  The statements are
tools/check_packed.py's (also runnable on its own); the host-side simulation of the same lanes is tests/test_fsm_sim.py."""
import os
import sys

import pytest

import alfalfa_amd as aa
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def device_route(monkeypatch):
    monkeypatch.setenv("ALFALFA_AMD_ROUTE", "device")       # (small calls would otherwise be parsed by host workers: dense)


@pytest.fixture(scope="module")
def packed_ctx():
    ctx = aa.Context(0)
    ctx.set_packed_coefficients(True)
    assert ctx.info()["packed_coefficients"] == 1
    return ctx


@pytest.mark.parametrize("name", ["qcif_q30_lf24", "w200_q40_lf63s7", "cif_q60_lf40s5", "qcif_allkey_q20", "synth_175x143_s3"])
def test_packed_records_and_rasters_match_host_parser_and_reference(packed_ctx, name):
    import check_packed
    assert check_packed.one_stream(packed_ctx, name) > 0
    st = packed_ctx.kernel_stats()
    # (a macroblock that stores blocks takes 25 mask slots + a word per coefficient: 41 words for a lone full block at worst)
    assert st["packed_frames"] > 0 and 0 < st["packed_words"] <= 41 * st["packed_blocks"]


def test_packed_and_host_parsed_frames_in_one_call_and_frames_decoded_twice(packed_ctx):
    """24 streams in lock step, every second one parsed on the host (dense records in the same reconstruction call), every
    third step reconstructed twice."""
    import check_packed
    assert check_packed.lock_step(packed_ctx, ["qcif_q30", "synth_96x80_s1", "w200_q40_lf63s7", "qvga_q100"], 6) > 0


def test_the_format_is_fixed_once_frames_were_submitted(packed_ctx, gpu_ctx):
    from conftest import golden_frames
    w, h, frames = golden_frames("qcif_q30")
    d = aa.Decoder(packed_ctx, w, h)
    packed_ctx.submit_frames([(d, frames[0])])
    with pytest.raises(aa.AlfalfaError) as e:
        packed_ctx.set_packed_coefficients(False)
    assert e.value.kind == "LogicError" or "submitted" in str(e.value)
    packed_ctx.set_packed_coefficients(True)                 # (saying what already holds is fine)
    packed_ctx.decode_batch([d], [0])
    from conftest import GOLDEN, sha256
    assert sha256(d.raster_bytes(0)) == GOLDEN["qcif_q30"]["raster_sha256"][0]
    assert gpu_ctx.info()["packed_coefficients"] in (0, 1)   # (the session's shared context: one format per run of the suite)
