#!/usr/bin/env python3
"""bench.py -- 1080p macroblocks/s of the VP8 decode hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the device half of the hot path (reconstruct + loop filter + reference update) over one
batch: S independent synthetic 1920x1080 inter-frame streams (1 key + F-1 inter frames each, loop filter level 24),
i.e. S*F*8160 macroblocks, whose parsed records are ALREADY RESIDENT IN HBM when the timed region starts.  Host parse
and H2D rates are measured separately and reported beside it (they are never part of `value`).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU: one process per GPU, independent streams per rank (weak scaling, no data-path collective); the only RCCL
traffic is the one-shot entry-state hand-off (shared reference raster broadcast over xGMI) before the timed region.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per macroblock, SURVEY.md 8(d): descriptor 80 + dense coefficients 800 + reference read 384
# + reconstruction write 384 (k_recon_inter) ; loop filter read+write 768 (k_loopfilter)
BYTES_PER_MB = {"recon_inter": 80 + 800 + 384 + 384, "recon_intra": 80 + 800 + 384, "loopfilter": 768}
PATH_BYTES_PER_MB = 2416       # inter + deblock, whole path
# HBM bytes per macroblock measured with rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KiB -> bytes, per launch / MBs per launch),
# profiles/r01j_kernels.md; used for roofline.traffic (counters cannot be read from inside this process)
PMC_TRAFFIC_BYTES_PER_MB = {"recon_inter": 497 + 407, "loopfilter": 356 + 653, "recon_intra": None}
KERNEL_NAMES = {"rows": {"recon_inter": "k_recon_inter4", "recon_intra": "k_recon_intra4", "loopfilter": "k_loopfilter_rows4"},
                "diagonal": {"recon_inter": "k_recon_inter", "recon_intra": "k_recon_intra", "loopfilter": "k_loopfilter"}}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="1080p_inter_lf")
    ap.add_argument("--streams", type=int, default=480, help="independent streams per GPU")
    ap.add_argument("--frames", type=int, default=12, help="frames per stream")
    ap.add_argument("--schedule", default="rows", choices=["rows", "diagonal"])
    ap.add_argument("--queues", type=int, default=1, help="independent HIP queues (contexts) the streams are spread over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("AA_BENCH_FORCE_DIST"):     # the env switch lets a 1-GPU box exercise the RCCL path
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import alfalfa_amd as aa
    import workload

    from alfalfa_amd import sharding
    width, height = workload.CONFIGS[args.config][:2]
    S, F = args.streams, args.frames
    # stream ids are disjoint across ranks; their CONTENT is drawn from a pool of 120 synthetic videos so that the one-off
    # generation cost (reference encoder, cached on disk) stays bounded on an 8-GPU node
    seeds = [100 + (g - 100) % 120 for g in sharding.stream_ids(rank, world, S)]
    t0 = time.time()
    paths = workload.make_streams(args.config, F, seeds)
    t_gen = time.time() - t0
    streams = [aa.read_ivf(p)[2] for p in paths]
    mbs_per_frame = ((width + 15) // 16) * ((height + 15) // 16)
    mbs_per_step = S * F * mbs_per_frame

    Q = max(1, min(args.queues, S))
    ctxs = [aa.Context(local_rank) for _ in range(Q)]
    for c in ctxs:
        c.set_schedule(args.schedule)
    ctx = ctxs[0]
    decs = [aa.Decoder(ctxs[i % Q], width, height) for i in range(S)]
    batches = [(ctxs[q], [d for i, d in enumerate(decs) if i % Q == q]) for q in range(Q)]      # (context, decoders submitted together)

    def sync_all():
        for c in ctxs:
            c.sync()

    # ---- multi-GPU only: one-shot entry-state hand-off (outside the timed region).  Rank 0 decodes the head (key frame)
    # of a shared GOP; its DecoderState blob and reference raster are broadcast (RCCL over xGMI) and every rank continues
    # the GOP from the imported state; all ranks must end on the same raster as a straight decode. ----
    handoff = None
    if dist is not None:
        import torch
        shared = aa.read_ivf(workload.make_stream(args.config, min(F, 4), 99))[2]
        cont = aa.Decoder(ctx, width, height)
        ysz, usz, vsz = cont.plane_sizes()
        planes = torch.empty(ysz + usz + vsz, dtype=torch.uint8, device="cuda")
        base = planes.data_ptr()
        blob = b""
        if rank == 0:
            head = aa.Decoder(ctx, width, height)
            _, fi = head.get_frame_output(shared[0])
            head.export_raster_device(fi, base, base + ysz, base + ysz + usz)
            ctx.sync()
            blob = head.export_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(planes, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        blob = sharding.broadcast_bytes(dist, blob, 0, device="cuda")
        cont.import_state(blob)
        cont.import_reference_device(base, base + ysz, base + ysz + usz)
        last = None
        for fr in shared[1:]:
            _, last = cont.get_frame_output(fr)
        digest = sharding.sha256(cont.raster_bytes(last))
        agree = sharding.digests_agree(dist, digest, device="cuda")
        if rank == 0:
            straight = aa.Decoder(ctx, width, height)
            for fr in shared:
                _, fi = straight.get_frame_output(fr)
            agree = agree and sharding.sha256(straight.raster_bytes(fi)) == digest
        handoff = {"raster_bytes": ysz + usz + vsz, "state_bytes": len(blob), "broadcast_ms": round(t_bcast * 1e3, 3),
                   "continuations_agree": bool(agree)}
        if not agree:
            raise SystemExit("entry-state hand-off mismatch across ranks")

    # ---- host half: serial BoolDecoder parse into pinned staging, one host thread per stream ----
    def parse_stream(i):
        t = time.perf_counter()
        for fr in streams[i]:
            decs[i].parse_frame(fr)
        return time.perf_counter() - t
    nthreads = min(S, os.cpu_count() or 1, 128)
    compressed_bytes = sum(len(fr) for st in streams for fr in st)
    # waves of `nthreads` streams: parse (one host thread per stream) -> H2D on the copy stream -> give the pinned staging
    # back, so that pinned host memory stays bounded however many streams a GPU holds
    per_stream_parse_s = []
    t_parse_wall = t_h2d = 0.0
    with ThreadPoolExecutor(max_workers=nthreads) as ex:
        for base in range(0, S, nthreads):
            ids = range(base, min(S, base + nthreads))
            t0 = time.perf_counter()
            per_stream_parse_s += list(ex.map(parse_stream, ids))
            t_parse_wall += time.perf_counter() - t0
            t0 = time.perf_counter()
            for i in ids:
                decs[i].upload()
            sync_all()
            t_h2d += time.perf_counter() - t0
            for i in ids:
                decs[i].release_staging()
    # the parser alone (no pinned/device allocation, one thread): the serial BoolDecoder rate per host core
    pp = aa.Parser(width, height)
    t0 = time.perf_counter()
    for fr in streams[0]:
        pp.parse(fr)
    parser_only = len(streams[0]) * mbs_per_frame / (time.perf_counter() - t0)

    def one_pass():
        for f in range(F):
            for c, part in batches:
                c.decode_batch(part, [f] * len(part))

    def one_step():
        one_pass()
        for d in decs:
            d.rewind()

    def barrier():
        sync_all()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    ms_per_step = elapsed / args.steps * 1e3
    value = world * mbs_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernel: per-launch HIP events on the compute stream, extra identical pass ----
    roofline = None
    kstats = None
    if not args.no_profile_pass:
        for c in ctxs:
            c.profile(True); c.kernel_stats(reset=True)
        one_pass()
        kstats = None
        for c in ctxs:
            st = c.kernel_stats(reset=True); c.profile(False)
            kstats = st if kstats is None else {k: kstats[k] + st[k] for k in st}
        for d in decs:
            d.rewind()
        names = ("recon_inter", "recon_intra", "loopfilter")
        dom = max(names, key=lambda k: kstats[k + "_ms"])
        launches = max(1, kstats[dom + "_launches"])
        total_ms = kstats[dom + "_ms"]
        # units per launch: inter = inter frames' MBs of one frame-step; loop filter = every MB once per frame-step
        mbs_total = S * (F - 1) * mbs_per_frame if dom == "recon_inter" else S * F * mbs_per_frame
        bytes_per_launch = BYTES_PER_MB[dom] * mbs_total / launches
        avg_ms = total_ms / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = PMC_TRAFFIC_BYTES_PER_MB.get(dom) if args.schedule == "rows" else None
        roofline = {"bound": "hbm", "kernel": KERNEL_NAMES[args.schedule][dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": None if traffic is None else round(traffic * mbs_total / launches),
                    "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/r01j_kernels.md (bytes per launch)",
                    "avg_launch_us": round(avg_ms * 1e3, 3), "launches_per_step": launches,
                    "algorithmic_bytes_per_launch": round(bytes_per_launch),
                    "path_frac_of_hbm_peak": round(value / world * PATH_BYTES_PER_MB / (HBM_PEAK_GBS * 1e9), 5)}

    # ---- cpu_baseline leg (rank 0, 1 GPU only): the REFERENCE decoder (oracle/_ref, built without x86 asm) on one of
    # the streams, single thread; its output doubles as the parity gate for that stream ----
    cpu_baseline = None
    verified = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref_time = os.path.join(ROOT, "oracle", "_ref", "ref_time")
        if os.path.exists(ref_time):
            reps = max(1, int(round(15.0 / (F * mbs_per_frame / 90000.0))))
            out = subprocess.run([ref_time, paths[0], str(reps)], check=True, capture_output=True, text=True).stdout
            r = json.loads(out)
            cpu_baseline = {"value": round(r["mb_per_s"], 1), "unit": "macroblocks/s", "cores": 1, "kind": "reference",
                            "sample": "stream seed %d (%d frames %dx%d) decoded %d times by oracle/_ref/ref_time; "
                                      "reference built without x86 asm (no assembler in the image)" % (seeds[0], F, width, height, reps),
                            "parse_fraction": round(r["parse_s"] / r["seconds"], 3)}
        else:   # reference binary not built: time our C restatement instead ("port")
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import vp8_oracle as vo
            od = vo.OracleDecoder(width, height)
            t1 = time.perf_counter()
            for fr in streams[0]:
                od.decode(fr)
            dt = time.perf_counter() - t1
            cpu_baseline = {"value": round(F * mbs_per_frame / dt, 1), "unit": "macroblocks/s", "cores": 1, "kind": "port",
                            "sample": "stream seed %d (%d frames %dx%d) decoded once by oracle/liboracle.so" % (seeds[0], F, width, height)}
    if rank == 0 and not args.no_verify:
        ref_decode = os.path.join(ROOT, "oracle", "_ref", "ref_decode")
        if os.path.exists(ref_decode):
            one_pass()
            raw = os.path.join(workload.cache_dir(), "bench_verify_%d.raw" % os.getpid())
            subprocess.run([ref_decode, paths[0], raw], check=True, stdout=subprocess.DEVNULL)
            ref = open(raw, "rb").read(); os.unlink(raw)
            fs = len(ref) // F
            verified = all(decs[0].raster_bytes(f) == ref[f * fs:(f + 1) * fs] for f in range(F))
            if not verified:
                raise SystemExit("PARITY FAILURE: HIP output differs from the reference decoder on the bench stream")

    if rank == 0:
        line = {
            "metric": "1080p macroblocks/s decode (bit-exact vs reference)" if height == 1080 else "%dp macroblocks/s decode (bit-exact vs reference)" % height,
            "value": round(value, 1), "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d independent %dx%d streams per GPU x %d frames (1 key + %d inter, y_ac_qi %d, loop filter %d), "
                                   "parsed records resident in HBM" % (args.config, S, width, height, F, F - 1,
                                                                        workload.CONFIGS[args.config][3], workload.CONFIGS[args.config][4]),
                       "streams_per_gpu": S, "frames_per_stream": F, "macroblocks_per_step_per_gpu": mbs_per_step,
                       "compressed_bytes_per_mb": round(compressed_bytes / mbs_per_step, 2), "sharding": "streams, one shard per GPU, no data-path collective", "schedule": args.schedule, "queues": Q},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            "host": {"parser_mb_per_s_per_core": round(parser_only, 1),
                     "parse_into_pinned_staging": {"threads": nthreads, "wall_s": round(t_parse_wall, 3),
                                                   "mb_per_s_aggregate": round(mbs_per_step / t_parse_wall, 1),
                                                   "note": "includes first-touch hipHostMalloc/hipMalloc of the frame store and raster slots"},
                     "h2d_s": round(t_h2d, 3), "stream_generation_s": round(t_gen, 1)},
            "kernel_stats": kstats, "verified_bit_exact_vs_reference": verified, "entry_state_handoff": handoff,
        }
    else:
        line = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # the JSON line is the LAST thing on stdout: flush what C libraries (RCCL's version banner) still hold in stdio
        # buffers first
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
