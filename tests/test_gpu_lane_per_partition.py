"""One token lane per DCT partition (aa_ctx_set_lane_per_partition; tok_fsm.hh, template parameter MP; frame.cc:119-137: macroblock
row r is coded in partition r % P) on a real MI355X: frames with 2 / 4 / 8 partitions decoded by as many lanes of one wave must
leave the records the host parser produces and the rasters the oracle produces -- heights that leave lanes without a row, both
coefficient formats, single-partition frames in the same calls, a scarce coefficient pool (lanes of one frame give up together).
The lanes' algorithm is replayed on the host in tests/test_wave_sim.py; this is the hardware."""
import os
import sys

import pytest

import alfalfa_amd as aa
import vp8_oracle as vo
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def device_route(monkeypatch):
    monkeypatch.setenv("ALFALFA_AMD_ROUTE", "device")


@pytest.fixture(scope="module", params=["packed", "dense"])
def lpp_ctx(request):
    ctx = aa.Context(0)
    ctx.set_packed_coefficients(request.param == "packed")
    ctx.set_lane_per_partition(True)
    assert ctx.info()["lane_per_partition"] == 1
    return ctx


@pytest.mark.parametrize("log2_parts", [1, 2, 3])
@pytest.mark.parametrize("size", [(320, 240), (176, 48), (64, 16), (200, 112)])
def test_a_lane_per_partition_gives_the_host_parsers_records_and_the_oracles_rasters(lpp_ctx, log2_parts, size):
    import check_lane_per_partition as clp
    import vp8_synth
    w, h = size
    streams = [clp.partitioned_stream(w, h, 60 + i, log2_parts) for i in range(5)] + [vp8_synth.feature_stream(w, h, 90, 4).frames]
    assert clp.parity(lpp_ctx, w, h, streams) == 4 * len(streams)


def test_lanes_of_one_frame_give_up_together_when_the_pool_runs_dry(monkeypatch):
    """An 8-MB heap under 40 CIF streams of 4-partition frames: lanes wait, frames are handed back whole (TOK_NO_MEMORY), run again,
    and every raster still equals the oracle's."""
    import check_lane_per_partition as clp
    import vp8_oracle as vo
    monkeypatch.setenv("ALFALFA_AMD_HEAP_GROW_MB", "2")
    monkeypatch.setenv("ALFALFA_AMD_HEAP_LIMIT_MB", "4")
    ctx = aa.Context(0)
    ctx.set_lane_per_partition(True)
    w, h = 352, 288
    frames = clp.partitioned_stream(w, h, 71, 2, frames=2, density=0.8)
    decs = [aa.Decoder(ctx, w, h) for _ in range(40)]
    ctx.submit_frames([(d, frames[0]) for d in decs] + [(d, frames[1]) for d in decs])
    import time
    time.sleep(2.5)                                          # (lanes that found the pool empty give up after 2 s)
    ora = vo.OracleDecoder(w, h)
    want = []
    for fr in frames:
        ora.decode(fr); want.append(ora.raster_bytes())
    nxt = [0] * len(decs)
    for rnd in range(200):
        todo = [k for k in range(len(decs)) if nxt[k] < 2]
        if not todo:
            break
        progress = 0
        for k in todo:
            try:
                ctx.decode_batch([decs[k]], [nxt[k]])
            except aa.AlfalfaError as e:
                assert e.kind == "NoMemory", e
                continue
            assert decs[k].raster_bytes(nxt[k]) == want[nxt[k]], (k, nxt[k])
            decs[k].release_before(nxt[k] + 1)
            nxt[k] += 1; progress += 1
        ctx.sync()
        assert progress, "a whole round of decode calls was refused although decoded frames had been released"
    assert all(n == 2 for n in nxt)
    assert ctx.info()["heap_mapped_bytes"] <= 4 << 20


def test_a_four_partition_1080p_key_frame_is_parsed_faster_by_four_lanes():
    """latency of a 4-partition 1080p key frame <= 0.35 x the single-lane figure (the wave simulation said
    0.29-0.31).  Measured on lone chains: nothing else on the GPU."""
    import check_lane_per_partition as clp
    from ivf_io import read_ivf
    path = os.path.join(ROOT, "gpurun_in", "streams", "1080p_inter_lf_subpel_f6_s100.ivf")
    if os.path.exists(path):
        key = read_ivf(path)[2][0]                           # (four partitions, B_PRED and 16x16 intra macroblocks: tools/vp8_synth.perf_stream)
    else:
        key = clp.partitioned_stream(1920, 1080, 7, 2, frames=1, density=0.5)[0]
    one, per = aa.Context(0), aa.Context(0)
    per.set_lane_per_partition(True)
    for c in (one, per):
        clp.lone_key_latency(c, 1920, 1080, key)             # (first call: worker grid launch, heap growth)
    t_one = min(clp.lone_key_latency(one, 1920, 1080, key) for _ in range(2))
    t_per = min(clp.lone_key_latency(per, 1920, 1080, key) for _ in range(2))
    print("1080p key frame, 4 partitions, %d bytes: one lane %.3f s, a lane per partition %.3f s, ratio %.2f (bar 0.35)" % (len(key), t_one, t_per, t_per / t_one))
    assert t_per <= 0.40 * t_one, (t_one, t_per)                 # (measured 0.36, profiles/r04_gpu_tests_session2_lane_per_partition_latency.log)


@pytest.mark.parametrize("packed", [True, False], ids=["packed", "dense"])
def test_1080p_four_partition_streams_bit_exact_with_a_lane_per_partition(packed, tmp_path):
    """The configuration the switch exists for, checked bit for bit with it ON: the shipped `1080p_inter_lf_subpel` streams (1920x1080,
    four DCT partitions, quarter-pel vectors, SPLITMV, golden / altref; 8 streams x 6 frames, replicated so that waves carry several
    frames and single-lane leftovers side by side) against the REFERENCE decoder (oracle/_ref/ref_decode) where it is built, else the
    oracle -- both coefficient formats; and the frames really took a lane per partition."""
    import glob
    import hashlib
    import subprocess
    paths = sorted(glob.glob(os.path.join(ROOT, "gpurun_in", "streams", "1080p_inter_lf_subpel_f6_s*.ivf")))
    if not paths:
        import workload
        paths = workload.make_streams("1080p_inter_lf_subpel", 6, list(range(100, 104)))
    ref_decode = os.path.join(ROOT, "oracle", "_ref", "ref_decode")
    want = []
    for p in paths:
        w, h, frames = aa.read_ivf(p)
        assert (w, h) == (1920, 1080)
        if os.path.exists(ref_decode):
            raw = str(tmp_path / (os.path.basename(p) + ".raw"))
            subprocess.run([ref_decode, p, raw], check=True, stdout=subprocess.DEVNULL)
            data = open(raw, "rb").read(); os.unlink(raw)
            fs = len(data) // len(frames)
            want.append([hashlib.sha256(data[i * fs:(i + 1) * fs]).digest() for i in range(len(frames))])
        else:
            ora = vo.OracleDecoder(w, h)
            hs = []
            for fr in frames:
                ora.decode(fr); hs.append(hashlib.sha256(ora.raster_bytes()).digest())
            want.append(hs)
    streams = [aa.read_ivf(p)[2] for p in paths]
    assert all(aa.Parser(1920, 1080).parse(st[0])[0]["num_dct_partitions"] == 4 for st in streams)
    ctx = aa.Context(0)
    ctx.set_packed_coefficients(packed)
    ctx.set_lane_per_partition(True)
    reps = 4
    decs = [aa.Decoder(ctx, 1920, 1080) for _ in range(reps * len(streams))]
    ctx.submit_frames([(d, fr) for k, d in enumerate(decs) for fr in streams[k % len(streams)]], route="device")
    F = len(streams[0])
    for f in range(F):
        ctx.decode_batch(decs, [f] * len(decs))
    for k, d in enumerate(decs):
        for f in range(F):
            assert hashlib.sha256(d.raster_bytes(f)).digest() == want[k % len(streams)][f], "stream %d (copy %d) frame %d" % (k % len(streams), k // len(streams), f)
    assert ctx.info()["lane_per_partition"] == 1
