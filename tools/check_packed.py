"""Packed coefficient storage on a real MI355X, in one short run (the pytest form is tests/test_gpu_packed_coefficients.py):
records and rasters of device-parsed frames of a context with aa_ctx_set_packed_coefficients(1) against the host parser, the
oracle and a dense context; several frames per call, frames decoded twice, a call mixing host- and device-parsed frames.

    python tools/check_packed.py [--big]      # --big: also 1080p streams of the benchmark workload
"""
import os
import sys
import time

os.environ.setdefault("ALFALFA_AMD_ROUTE", "device")        # (small calls would otherwise go to host workers)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np

import alfalfa_amd as aa
import vp8_oracle as vo
from conftest import GOLDEN, golden_frames, sha256


def records_equal(got, want):
    gh, gmb, gcf = got
    wh, wmb, wcf = want
    return gh == wh and (gmb.reshape(-1).view(np.uint8) == wmb.reshape(-1).view(np.uint8)).all() and gcf.shape == wcf.shape and (gcf == wcf).all()


def one_stream(ctx, name):
    w, h, frames = golden_frames(name)
    dec, host, ora = aa.Decoder(ctx, w, h), aa.Parser(w, h), vo.OracleDecoder(w, h)
    idx = ctx.submit_frames([(dec, fr) for fr in frames])
    assert idx == list(range(len(frames)))
    for fi, fr in enumerate(frames):
        assert records_equal(dec.read_records(fi), host.parse(fr)), (name, fi, "records")
        ctx.decode_batch([dec], [fi])
        ora.decode(fr)
        got = dec.raster_bytes(fi)
        assert got == ora.raster_bytes(), (name, fi, "raster")
        assert sha256(got) == GOLDEN[name]["raster_sha256"][fi]
    return len(frames)


def lock_step(ctx, names, copies):
    """copies x len(names) streams, one frame of every stream per call; every second stream parsed on the host"""
    streams = []
    for c in range(copies):
        for name in names:
            w, h, frames = golden_frames(name)
            streams.append((name, aa.Decoder(ctx, w, h), frames))
    steps = min(len(f) for _, _, f in streams)
    for t in range(steps):
        dev = [(d, f[t]) for k, (_, d, f) in enumerate(streams) if k % 2 == 0]
        ctx.submit_frames(dev)
        for k, (_, d, f) in enumerate(streams):
            if k % 2:
                d.parse_frame(f[t])
        ctx.decode_batch([d for _, d, _ in streams], [t] * len(streams))
        if t % 3 == 2:                       # the same frames once more (the dense blocks are made again)
            for _, d, _ in streams:
                d.rewind_to(t)
            ctx.decode_batch([d for _, d, _ in streams], [t] * len(streams))
        for name, d, _ in streams:
            assert sha256(d.raster_bytes(t)) == GOLDEN[name]["raster_sha256"][t], (name, t)
    return steps * len(streams)


def big(ctx_packed, ctx_dense, n_streams, n_frames):
    import workload
    one = aa.read_ivf(workload.make_stream("1080p_inter_lf", n_frames, 105))
    streams = [one] * n_streams
    out = []
    for ctx in (ctx_packed, ctx_dense):
        decs = [aa.Decoder(ctx, w, h) for w, h, _ in streams]
        for t in range(n_frames):
            ctx.submit_frames([(d, s[2][t]) for d, s in zip(decs, streams)])
        hashes = []
        for t in range(n_frames):
            ctx.decode_batch(decs, [t] * len(decs))
            hashes.append([d.raster_hash(t) for d in decs])
        out.append(hashes)
    assert out[0] == out[1], "1080p: packed and dense contexts disagree"
    w, h, frames = streams[0]
    ora = vo.OracleDecoder(w, h)
    d = aa.Decoder(ctx_packed, w, h)
    ctx_packed.submit_frames([(d, fr) for fr in frames[:2]])
    for t in range(2):
        ctx_packed.decode_batch([d], [t])
        ora.decode(frames[t])
        assert d.raster_bytes(t) == ora.raster_bytes(), ("1080p oracle", t)
    return n_streams * n_frames


def main():
    t0 = time.time()
    ctx = aa.Context(0)
    ctx.set_packed_coefficients(True)
    assert ctx.info()["packed_coefficients"] == 1
    n = 0
    for name in ("qcif_q30_lf24", "w200_q40_lf63s7", "cif_q60_lf40s5", "qcif_allkey_q20", "synth_175x143_s3"):
        n += one_stream(ctx, name)
        print("ok", name, flush=True)
    n += lock_step(ctx, ["qcif_q30", "synth_96x80_s1", "w200_q40_lf63s7", "qvga_q100"], 6)
    print("ok lock step", flush=True)
    if "--big" in sys.argv:
        dense = aa.Context(0)
        dense.set_packed_coefficients(False)
        n += big(ctx, dense, 6, 3)
        print("ok 1080p", flush=True)
    st = ctx.kernel_stats()
    assert st["packed_frames"] > 0 and st["packed_words"] > 0
    print("PACKED OK: %d frames; packed frames %d, %.2f words per dense block (dense: 16), %.1f s"
          % (n, st["packed_frames"], st["packed_words"] / max(1, st["packed_blocks"]), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
