"""Device-parser probe: S streams x F frames of a bench config through aa_submit_frames, nothing else on the GPU.
Prints wall time of the parse and the kernels' HIP-event times; meant to be run under rocprofv3 (--pmc ...) too.
python tools/parse_probe.py [--config 1080p_inter_lf] [--streams 32] [--frames 12] [--reps 2]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import alfalfa_amd as aa  # noqa: E402
import workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="1080p_inter_lf"); ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--frames", type=int, default=12); ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--first", type=int, default=0, help="first frame of every stream to submit (skip the key frame with 1)")
    a = ap.parse_args()
    w, h = workload.CONFIGS[a.config][:2]
    paths = workload.make_streams(a.config, a.frames, [100 + i % 120 for i in range(a.streams)])
    streams = [aa.read_ivf(p)[2] for p in paths]
    ctx = aa.Context(0)
    out = []
    for rep in range(a.reps):
        decs = [aa.Decoder(ctx, w, h) for _ in streams]
        if a.first:     # state up to the first submitted frame comes from the host parser
            for d, st in zip(decs, streams):
                for fr in st[:a.first]:
                    d.parse_frame(fr)
        ctx.profile(True); ctx.kernel_stats(reset=True)
        pairs = [(d, fr) for d, st in zip(decs, streams) for fr in st[a.first:]]
        t0 = time.perf_counter()
        ctx.submit_frames(pairs)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        ks = ctx.kernel_stats(reset=True); ctx.profile(False)
        mbs = len(pairs) * ((w + 15) // 16) * ((h + 15) // 16)
        out.append({"chains": len(pairs), "host_s": round(t1 - t0, 4), "parse_wall_s": round(t2 - t0, 4), "mb_per_s": round(mbs / (t2 - t0)),
                    "headers_ms": round(ks["parse_headers_ms"], 2), "tokens_ms": round(ks["parse_tokens_ms"], 2)})
        del decs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
