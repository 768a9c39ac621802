"""One token lane per DCT partition (aa_ctx_set_lane_per_partition; tok_fsm.hh, template parameter MP) on a real MI355X -- the
stand-alone form of tests/test_gpu_lane_per_partition.py (run on the GPU since round 4: profiles/r04_gpu_tests_session2_lane_per_partition_latency.log).

    python tools/check_lane_per_partition.py            # parity: records vs the host parser, rasters vs the oracle
    python tools/check_lane_per_partition.py --latency  # + a 1080p 4-partition key frame: parse latency vs one lane (measured 0.36)
"""
import os
import sys
import time

os.environ.setdefault("ALFALFA_AMD_ROUTE", "device")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np

import alfalfa_amd as aa
import vp8_oracle as vo
import vp8_synth


def partitioned_stream(w, h, seed, log2_parts, frames=4, density=0.4):
    s = vp8_synth.SynthStream(w, h, seed)
    s.frame(key=True, q_index=20, skip_prob=200, density=density, log2_parts=log2_parts, lf_level=10)
    for k in range(frames - 1):
        s.frame(key=False, q_index=30, skip_prob=100 + 20 * k, density=density * 0.7, log2_parts=log2_parts, lf_level=8, skip_rate=0.3 * k)
    return s.frames


def records_equal(got, want):
    gh, gmb, gcf = got
    wh, wmb, wcf = want
    return gh == wh and (gmb.reshape(-1).view(np.uint8) == wmb.reshape(-1).view(np.uint8)).all() and gcf.shape == wcf.shape and (gcf == wcf).all()


def parity(ctx, w, h, streams):
    decs = [aa.Decoder(ctx, w, h) for _ in streams]
    hosts = [aa.Parser(w, h) for _ in streams]
    oras = [vo.OracleDecoder(w, h) for _ in streams]
    n = min(len(s) for s in streams)
    ctx.submit_frames([(d, s[f]) for d, s in zip(decs, streams) for f in range(n)])
    for f in range(n):
        for d, hp, s in zip(decs, hosts, streams):
            assert records_equal(d.read_records(f), hp.parse(s[f])), ("records", w, h, f)
        ctx.decode_batch(decs, [f] * len(decs))
        for d, o, s in zip(decs, oras, streams):
            o.decode(s[f])
            assert d.raster_bytes(f) == o.raster_bytes(), ("raster", w, h, f)
    return n * len(streams)


def lone_key_latency(ctx, w, h, frame):
    d = aa.Decoder(ctx, w, h)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.submit_frames([(d, frame)])
    d.frame_header(0)                   # waits for the parse
    return time.perf_counter() - t0


def main():
    frames = 0
    for packed in (False, True):
        ctx = aa.Context(0)
        ctx.set_lane_per_partition(True)
        ctx.set_packed_coefficients(packed)
        assert ctx.info()["lane_per_partition"] == 1
        for log2_parts in (1, 2, 3):
            for w, h in ((320, 240), (176, 48), (64, 16)):
                streams = [partitioned_stream(w, h, 60 + i, log2_parts) for i in range(5)] + [vp8_synth.feature_stream(w, h, 90, 4).frames]
                frames += parity(ctx, w, h, streams)
            print("ok %d partitions%s" % (1 << log2_parts, " (packed)" if packed else ""), flush=True)
    print("LANE PER PARTITION OK: %d frames" % frames, flush=True)
    if "--latency" in sys.argv:
        w, h = 1920, 1080
        key = partitioned_stream(w, h, 7, 2, frames=1, density=0.5)[0]
        one, per = aa.Context(0), aa.Context(0)
        per.set_lane_per_partition(True)
        for c in (one, per):
            lone_key_latency(c, w, h, key)                 # (first call: worker grid launch, heap growth)
        t_one = min(lone_key_latency(one, w, h, key) for _ in range(3))
        t_per = min(lone_key_latency(per, w, h, key) for _ in range(3))
        print("1080p key frame, 4 partitions, %d bytes: one lane %.3f s, a lane per partition %.3f s, ratio %.2f (bar 0.35)"
              % (len(key), t_one, t_per, t_per / t_one), flush=True)


if __name__ == "__main__":
    main()
