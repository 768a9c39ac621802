"""Reconstruction kernels alone, for counter passes (rocprofv3 --pmc serialises dispatches and needs every dispatch to END):
S streams x F frames of a bench config are parsed on the device with ALFALFA_AMD_WORKER_LINGER_MS=0 (worker waves leave when
the queue is empty), the host waits for every frame's parse, then the reconstruction of all frames is replayed `reps` times --
k_dense_index / k_expand_coeffs, k_recon_inter4, k_recon_intra4, k_loopfilter_rows4 with no worker grid resident.
    python tools/recon_replay.py [--config 1080p_inter_lf] [--streams 120] [--frames 4] [--reps 2]
Prints one JSON line: macroblocks per kind per replay, HIP-event times per kernel (what the counters are divided by)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("ALFALFA_AMD_WORKER_LINGER_MS", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import alfalfa_amd as aa  # noqa: E402
import workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="1080p_inter_lf"); ap.add_argument("--streams", type=int, default=120)
    ap.add_argument("--frames", type=int, default=4); ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--distinct", type=int, default=24)
    a = ap.parse_args()
    w, h = workload.CONFIGS[a.config][:2]
    paths = workload.make_streams(a.config, a.frames, [100 + i % a.distinct for i in range(a.streams)])
    streams = [aa.read_ivf(p)[2] for p in paths]
    ctx = aa.Context(0)
    decs = [aa.Decoder(ctx, w, h) for _ in streams]
    t0 = time.perf_counter()
    ctx.submit_frames([(d, fr) for d, st in zip(decs, streams) for fr in st], route="device")
    kinds = {"inter_whole": 0, "inter_split": 0, "intra": 0}
    for d, st in zip(decs, streams):
        for i in range(len(st)):
            d.frame_header(i)                              # waits for the frame's parse
    for d in decs[:a.distinct]:
        for i in range(a.frames):
            _, mb, _ = d.read_records(i)
            inter = (mb["flags"] & 4) != 0
            sp = int((inter & (mb["y_mode"] == 9)).sum())
            kinds["inter_split"] += sp; kinds["inter_whole"] += int(inter.sum()) - sp; kinds["intra"] += int((~inter).sum())
    scale = a.streams / min(a.distinct, a.streams)
    kinds = {k: int(v * scale) for k, v in kinds.items()}
    t_parse = time.perf_counter() - t0
    time.sleep(0.5)                                        # (worker waves with linger 0 are gone by now)

    def replay():
        for f in range(a.frames):
            ctx.decode_batch(decs, [f] * len(decs))
        for d in decs:
            d.rewind_to(0)
    replay(); ctx.sync()
    ctx.profile(True); ctx.kernel_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        replay()
    ctx.sync()
    dt = (time.perf_counter() - t0) / a.reps
    ks = ctx.kernel_stats(reset=True); ctx.profile(False)
    mbs = len(decs) * a.frames * ((w + 15) // 16) * ((h + 15) // 16)
    out = {"config": a.config, "streams": a.streams, "frames": a.frames, "reps": a.reps, "replays_incl_warmup": a.reps + 1, "macroblocks_per_replay": mbs,
           "macroblocks_by_kind_per_replay": kinds, "parse_s": round(t_parse, 2), "replay_ms": round(dt * 1e3, 2), "mb_per_s": round(mbs / dt),
           "kernel_ms_per_replay": {k: round(ks[k + "_ms"] / a.reps, 3) for k in ("recon_inter", "recon_split", "recon_intra", "loopfilter", "expand")},
           "launches_per_replay": {k: ks[k + "_launches"] // a.reps for k in ("recon_inter", "recon_split", "recon_intra", "loopfilter", "expand")},
           "packed": bool(ctx.info()["packed_coefficients"])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
