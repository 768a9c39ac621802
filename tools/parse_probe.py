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
    prev = [0] * 8
    for rep in range(a.reps):
        decs = [aa.Decoder(ctx, w, h) for _ in streams]
        if a.first:     # state up to the first submitted frame comes from the host parser
            for d, st in zip(decs, streams):
                for fr in st[:a.first]:
                    d.parse_frame(fr)
        ctx.profile(True); ctx.kernel_stats(reset=True)
        pairs = [(d, fr) for d, st in zip(decs, streams) for fr in st[a.first:]]
        t0 = time.perf_counter()
        ctx.submit_frames(pairs, route="device")
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        for d, st in zip(decs, streams):
            for i in range(a.first, len(st)):
                d.frame_header(i)                         # (the parse summaries: decode steps per frame)
        ks = ctx.kernel_stats(reset=True); ctx.profile(False)
        info = ctx.info()
        mbs = len(pairs) * ((w + 15) // 16) * ((h + 15) // 16)
        lanes = info["token_lanes_per_workgroup"] * info["token_workgroups_capacity"]
        rec = {"chains": len(pairs), "host_s": round(t1 - t0, 4), "parse_wall_s": round(t2 - t0, 4), "mb_per_s": round(mbs / (t2 - t0)),
               "headers_ms": round(ks["parse_headers_ms"], 2), "worker_grid_ms_sum": round(ks["parse_tokens_ms"], 2), "worker_grids": ks["worker_launches"],
               "worker_wgs": ks["worker_wgs"], "lane_steps": ks["token_steps"], "lanes": lanes, "lanes_per_wg": info["token_lanes_per_workgroup"],
               "us_per_step_if_all_lanes_busy": round((t2 - t0) * 1e6 * min(lanes, len(pairs)) / max(1, ks["token_steps"]), 4),
               "heap_mapped_gb": round(info["heap_mapped_bytes"] / 1e9, 1), "starved": info["lanes_starved"]}
        p = info["token_profile"]
        if p[7] and prev is not None:
            d8 = [x - y for x, y in zip(p, prev)]
            tot = d8[3] + d8[4] + d8[5]
            rec["profile"] = {"wave_seconds": round(tot / 1e8, 2), "frac_boundary": round(d8[0] / tot, 3), "frac_take_and_begin": round(d8[3] / tot, 3),
                              "frac_top_up": round(d8[4] / tot, 3), "frac_steps": round((d8[5] - d8[0]) / tot, 3),
                              "us_per_boundary_pass": round(d8[0] / max(1, d8[1]) / 100.0, 3), "us_per_wave_step": round((d8[5] - d8[0]) / max(1, d8[2]) / 100.0, 4),
                              "wave_steps": d8[2], "boundary_passes": d8[1], "lanes_with_frame_per_period": round(d8[6] / max(1, d8[7]), 2),
                              "steps_per_period": round(d8[2] / max(1, d8[7]), 2)}
        prev = p
        out.append(rec)
        del decs
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
