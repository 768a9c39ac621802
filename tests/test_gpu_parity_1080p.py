"""Parity at the BASELINE geometry (1920x1080: 120x68 macroblocks, 1080 -> 1088 rows of padding), on a real MI355X:
  * every distinct stream of the benchmark workload (120 synthetic videos x 12 frames, the reference encoder's output)
    decoded through the GPU entropy decoder, every frame against the REFERENCE decoder itself (oracle/_ref/ref_decode);
    a sample of them also through the host parser;
  * synthetic feature streams at 1920x1080 (SPLITMV, golden/altref, segmentation, up to 8 partitions, far vectors) against
    the oracle, both parse paths;
  * 1280x720 all-intra and inter streams (BASELINE configs 1 and 2), every frame, against the oracle."""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

import alfalfa_amd as aa
import vp8_oracle as vo
from conftest import ROOT

pytestmark = pytest.mark.gpu

REF_DECODE = os.path.join(ROOT, "oracle", "_ref", "ref_decode")


def reference_hashes(path, nframes, tmpdir):
    """SHA-256 of every frame's three padded planes as the reference decoder writes them."""
    raw = os.path.join(tmpdir, "ref_%d_%s.raw" % (os.getpid(), os.path.basename(path)))
    subprocess.run([REF_DECODE, path, raw], check=True, stdout=subprocess.DEVNULL)
    with open(raw, "rb") as fh:
        data = fh.read()
    os.unlink(raw)
    fs = len(data) // nframes
    assert fs * nframes == len(data)
    return [hashlib.sha256(data[i * fs:(i + 1) * fs]).digest() for i in range(nframes)]


@pytest.mark.skipif(not os.path.exists(REF_DECODE), reason="oracle/_ref (the reference built in place) is not here")
def test_every_bench_stream_matches_the_reference_decoder(gpu_ctx, tmp_path):
    import workload
    F, seeds = 12, list(range(100, 220))
    paths = workload.make_streams("1080p_inter_lf", F, seeds)
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        want = list(ex.map(lambda p: reference_hashes(p, F, str(tmp_path)), paths))
    streams = [aa.read_ivf(p)[2] for p in paths]
    for base in range(0, len(seeds), 40):                       # 40 streams at a time: bounded memory, still a batch
        part = list(range(base, min(base + 40, len(seeds))))
        decs = [aa.Decoder(gpu_ctx, 1920, 1080) for _ in part]
        gpu_ctx.submit_frames([(d, fr) for d, i in zip(decs, part) for fr in streams[i]], route="device")        # GPU entropy decode
        for f in range(F):
            gpu_ctx.decode_batch(decs, [f] * len(decs))
        for d, i in zip(decs, part):
            for f in range(F):
                assert hashlib.sha256(d.raster_bytes(f)).digest() == want[i][f], "seed %d frame %d (GPU parser)" % (seeds[i], f)
        del decs
    for i in range(0, len(seeds), 15):                          # and the host parser on a sample
        d = aa.Decoder(gpu_ctx, 1920, 1080)
        for f, fr in enumerate(streams[i]):
            _, fi = d.get_frame_output(fr)
            assert hashlib.sha256(d.raster_bytes(fi)).digest() == want[i][f], "seed %d frame %d (host parser)" % (seeds[i], f)


@pytest.mark.parametrize("seed", [7701, 7702, 7703])
def test_feature_streams_at_1080p(gpu_ctx, seed):
    import vp8_synth
    frames = vp8_synth.feature_stream(1920, 1080, seed, 3).frames
    ora = vo.OracleDecoder(1920, 1080)
    want = []
    for fr in frames:
        ora.decode(fr)
        want.append(ora.raster_bytes())
    a, b = aa.Decoder(gpu_ctx, 1920, 1080), aa.Decoder(gpu_ctx, 1920, 1080)
    idx = gpu_ctx.submit_frames([(a, fr) for fr in frames], route="device")
    for f, fr in enumerate(frames):
        gpu_ctx.decode_batch([a], [idx[f]])
        _, fi = b.get_frame_output(fr)
        assert a.raster_bytes(idx[f]) == want[f], "seed %d frame %d (GPU parser): %s" % (seed, f, ora.frame_info())
        assert b.raster_bytes(fi) == want[f], "seed %d frame %d (host parser)" % (seed, f)


@pytest.mark.parametrize("config,frames", [("720p_intra", 6), ("720p_inter", 12)])
def test_720p_configs_every_frame(gpu_ctx, config, frames):
    import workload
    for seed in (300, 301, 302):
        w, h, fr = aa.read_ivf(workload.make_stream(config, frames, seed))
        ora, dec = vo.OracleDecoder(w, h), aa.Decoder(gpu_ctx, w, h)
        idx = gpu_ctx.submit_frames([(dec, f) for f in fr], route="device")
        for i, f in enumerate(fr):
            gpu_ctx.decode_batch([dec], [idx[i]])
            ora.decode(f)
            assert dec.raster_bytes(idx[i]) == ora.raster_bytes(), (config, seed, i)


def test_realistic_inter_workload_streams(gpu_ctx):
    """The "realistic inter" workload of bench.py --config 1080p_inter_lf_subpel (quarter-pel vectors, ~15 % SPLITMV, golden /
    altref, four partitions) at CIF size, every frame, both parse paths, against the oracle."""
    import workload
    for seed in (100, 101, 102, 103):
        w, h, fr = aa.read_ivf(workload.make_stream("cif_inter_lf_subpel", 8, seed))
        ora, a, b = vo.OracleDecoder(w, h), aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h)
        idx = gpu_ctx.submit_frames([(a, f) for f in fr], route="device")
        for i, f in enumerate(fr):
            gpu_ctx.decode_batch([a], [idx[i]])
            _, fi = b.get_frame_output(f)
            ora.decode(f)
            assert a.raster_bytes(idx[i]) == ora.raster_bytes(), (seed, i, "GPU parser")
            assert b.raster_bytes(fi) == ora.raster_bytes(), (seed, i, "host parser")
