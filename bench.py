#!/usr/bin/env python3
"""bench.py -- 1080p macroblocks/s of the VP8 decode hot path on MI355X (BASELINE.json metric), END TO END.

One "step" = one pass of the WHOLE hot path over one batch: S independent synthetic 1920x1080 inter-frame streams
(1 key + F-1 inter frames each, loop filter level 24) = S*F*8160 macroblocks, from COMPRESSED FRAMES IN HOST MEMORY to
filtered rasters in HBM:
    host    frame-header pre-pass (serial across the frames of a stream) + staging of the compressed bytes   [C++ workers]
    H2D     the compressed frames themselves (~36 B/macroblock instead of ~360 B of parsed records)
    GPU     BoolDecoder entropy decode: macroblock headers + tokens, one lane per (stream, frame)   [k_parse_*, k_token_workers]
    GPU     reconstruction + loop filter + reference update                                        [k_recon_*, k_loopfilter_*]
Steps are pipelined: a frame's entropy decode is one serial chain on one GPU lane (seconds for a 1080p frame), so the rate
comes from the number of chains in flight.  Key frames are handed to the GPU parser `--key-ahead` steps before their group
is reconstructed, inter frames `--depth` steps before -- as far as the HBM budget (`--hbm-gb`) holds what they store; frames
are released as they are consumed, so memory is a ring.  The timed region runs K steps from an EMPTY pipeline to an EMPTY
pipeline (it pays for filling and draining); `value` is that end-to-end rate, `steady_state` the rate between fill and drain.
The LAST step of the timed region is the one whose rasters are compared with the reference decoder.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU: one process per GPU, independent streams per rank (weak scaling, no data-path collective); the only RCCL
traffic is the one-shot entry-state hand-off (shared reference raster broadcast over xGMI) before the timed region.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

# The pipeline runs 15 HIP streams side by side; HIP multiplexes streams over GPU_MAX_HW_QUEUES (default 4) hardware queues and
# reads the variable when the runtime starts -- which, with torch.distributed, is before libalfalfa_amd.so is loaded.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# the worker waves account for their own time (a few clock reads per 32 decode steps): the entropy decode's own roof in the
# bench line comes from these counters
os.environ.setdefault("ALFALFA_AMD_TOKEN_PROFILE", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per macroblock, SURVEY.md 8(d): descriptor 80 + dense coefficients 800 + reference read 384
# + reconstruction write 384 (k_recon_inter) ; loop filter read+write 768 (k_loopfilter); the entropy decode reads the
# compressed macroblock and writes the descriptor + dense coefficients that the survey's model has the reconstruction read
BYTES_PER_MB = {"recon_inter": 80 + 800 + 384 + 384, "recon_split": 80 + 800 + 384 + 384, "recon_intra": 80 + 800 + 384, "loopfilter": 768,
                "parse_tokens": 80 + 800, "parse_headers": 80}
PATH_BYTES_PER_MB = 2416       # inter + deblock, whole path
DELIVER_RING = int(os.environ.get("AA_BENCH_DELIVER_RING") or 3)     # pinned slabs the delivered frames go to (one frame index of every stream each)
KERNEL_NAMES = {"recon_inter": "k_recon_inter4", "recon_split": "k_recon_inter", "recon_intra": "k_recon_intra4", "loopfilter": "k_loopfilter_rows4",
                "parse_tokens": "k_token_workers", "parse_headers": "k_parse_mb_headers"}


TRAFFIC_SOURCES = {}         # kernel key -> where its traffic figure was measured (profiles/pmc_traffic.json "per_kernel_source")


def pmc_traffic(config):
    """HBM bytes per macroblock per kernel from rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/profile_summary.py): {"round": ..., config: {kernel key: bytes per macroblock}}.  Only figures measured on the kernels
    of THIS round's tree are used (the file says which round measured them); else None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None, None
    if d.get("round") != "r06":
        return None, None
    global TRAFFIC_SOURCES
    TRAFFIC_SOURCES = d.get("per_kernel_source") or {}
    return d.get(config), d.get("source")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="1080p_inter_lf")
    ap.add_argument("--streams", type=int, default=480, help="independent streams per GPU")
    ap.add_argument("--frames", type=int, default=12, help="frames per stream per step")
    ap.add_argument("--key-ahead", type=int, default=16, help="steps by which KEY frames are handed to the GPU parser ahead of reconstruction (bounded by what the HBM budget holds)")
    ap.add_argument("--depth", type=int, default=8, help="steps by which inter frames are handed to the GPU parser ahead of reconstruction (bounded likewise)")
    ap.add_argument("--hbm-gb", type=float, default=240.0, help="HBM the decoder context may use on each GPU (aa_ctx_set_memory_limit); the look-ahead is planned inside it")
    ap.add_argument("--header-ahead", type=int, default=0, help="steps by which the macroblock-header pass of inter frames runs ahead of their token pass (two-phase submit)")
    ap.add_argument("--threads", type=int, default=0, help="host workers of the header pre-pass (0: cores / local ranks)")
    ap.add_argument("--schedule", default="rows", choices=["rows", "diagonal"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-device-half", action="store_true")
    ap.add_argument("--profile-timed", action="store_true", help="HIP-event timing of every kernel INSIDE the timed region too (diagnostics)")
    ap.add_argument("--overcommit", type=float, default=1.2, help="frames are admitted while what those in flight are EXPECTED to store stays below this x the room "
                    "(frames being parsed hold only part of it, and the oldest are released first; a lane that finds the pool empty waits)")
    ap.add_argument("--dense", action="store_true", help="device-parsed frames store DENSE 32-byte coefficient blocks (aa_ctx_set_packed_coefficients( 0 )) instead of the "
                    "default packed form (a mask word + the non-zero values per block, expanded on the device when a frame is reconstructed)")
    ap.add_argument("--packed", action="store_true", help="(the default since round 4; accepted for old command lines)")
    ap.add_argument("--deliver", action="store_true", help="every reconstructed frame is also DELIVERED: copied to pinned host memory (aa_download_batch_async: one "
                    "gather + one copy per frame index) while the next frames are decoded -- what vp8decode / xc-decode-bundle do with every shown frame; "
                    "the timed region then includes PCIe")
    ap.add_argument("--deliver-steps", type=int, default=12, help="without --deliver: this many steps with every frame delivered AFTER the timed region, reported as `delivery` (0 = skip)")
    ap.add_argument("--host-share-ms", type=float, default=None, help="aa_ctx_set_host_share_ms: key frames of a hand-over are parsed by host workers while that is "
                    "expected to take no longer than this on the rank's host threads (library default 80; 0: every frame on the GPU's token lanes)")
    ap.add_argument("--lanes-only-steps", type=int, default=8, help="after the main run: this many steps with host_share_ms = 0 (every frame, key frames too, "
                    "parsed by GPU lanes), reported as all_frames_on_gpu_lanes (0 = skip)")
    ap.add_argument("--urgent-host", action="store_true", help="(accepted for old command lines: the host lanes are the default for the groups a pipeline starts with)")
    ap.add_argument("--urgent-groups", type=int, default=2, help="groups of pictures, counted from the one an EMPTY pipeline starts with, whose key frames go to the context's host lanes "
                    "(AA_SUBMIT_HOST: host cores in the role of token lanes, the call does not wait)")
    ap.add_argument("--no-urgent-host", action="store_true", help="key frames of the group a pipeline STARTS with also take the default route (GPU lanes) instead of the "
                    "host route (AA_SUBMIT_HOST); they are the ones whose chain latency is the fill of the pipeline")
    ap.add_argument("--trace-memory", action="store_true", help="print the context's memory books after every step of the timed region (stderr)")
    ap.add_argument("--small-batches", default="1,8,64", help="stream counts for the small-batch end-to-end figures ('' = skip)")
    ap.add_argument("--secondary", default="720p_intra,720p_inter,1080p_inter_lf_subpel",
                    help="other BASELINE configs measured end to end after the main run, a few steps each, parity checked in the run ('' = skip)")
    ap.add_argument("--two-cpu-steps", type=int, default=8, help="rank 0 of a 1-GPU run: the same workload once more in a CHILD process confined to 2 CPUs (affinity, 2 host lanes, "
                    "2 pre-pass threads) -- what a rank gets when eight share a 16-CPU grant -- this many steps, reported as per_rank_on_2_cpus (0 = skip)")
    ap.add_argument("--secondary-streams", type=int, default=96)
    ap.add_argument("--secondary-steps", type=int, default=4)
    return ap.parse_args()


class Pipeline:
    """Group g = one group of pictures of every stream = S decoders created when its key frames are handed over (a new
    Decoder per chunk, as xc-decode-bundle does) and dropped when its last frame has been reconstructed: a decoder waiting
    for its turn holds no raster (its references point at the context's shared blank one).

    A step's S groups of pictures are S independent decode jobs (ExCamera chunks: one Decoder each, xc-decode-bundle).  The
    entropy decode of a frame is ONE serial chain on ONE GPU lane, so what matters is how many chains are in flight; and a key
    frame's chain is ~2.6x longer than an inter frame's.  The scheduler therefore hands key frames to the GPU `key_ahead`
    steps before their group is reconstructed and inter frames only `depth` steps before (they would otherwise sit in HBM
    waiting for their key frame)."""

    def __init__(self, env, stream_list, key_ahead, depth, header_ahead=0):
        import alfalfa_amd as aa
        self.aa, self.env = aa, env
        self.ctx, self.F = env["ctx"], env["F"]
        self.streams = stream_list
        self.n = len(stream_list)
        self.H = max(0, header_ahead)
        self.inter_h = 0
        self.K, self.D = max(1, key_ahead), max(1, min(depth, key_ahead))
        if env.get("keys_on_host"):
            # key frames are parsed by host workers inside the hand-over call (milliseconds, not a 2.4-s chain on a lane): nothing
            # is gained by handing them over earlier than the inter frames that follow them
            self.K = self.D
        self.groups = {}                                # g -> [Decoder]
        self.kept = {}                                  # g -> [Decoder] whose frames were not released (the step that is verified)
        self.keep_group = None                          # group whose distinct decoders keep every frame (verification of a TIMED step)
        n, F = self.n, self.F
        # argument blocks, reused: only the decoder handles change from group to group
        self.key_arr = (aa.capi.FrameIn * n)(); self.key_out = (C.c_int * n)()
        self.inter_arr = (aa.capi.FrameIn * (n * (F - 1)))(); self.inter_out = (C.c_int * (n * (F - 1)))()
        for i, st in enumerate(stream_list):
            self.key_arr[i].data, self.key_arr[i].size = st[0], len(st[0])
            for k, fr in enumerate(st[1:]):             # stream-major: the inter frames of one stream are consecutive (a host worker takes a whole stream)
                e = self.inter_arr[i * (F - 1) + k]
                e.data, e.size = fr, len(fr)
        self.keys = self.inters = self.decoded = 0      # groups handed to the GPU parser (key / inter frames), groups reconstructed
        self.frames_submitted = 0
        self.host_s = 0.0
        self.t_launch = self.t_decode = self.t_release = 0.0      # host time in aa_launch_tokens / aa_decode_batch / releases
        self.done_t = []
        self.t_room = 0.0                                # host time in aa_ctx_get_info (the planner's look at the books)
        self.t_decode_mark = 0.0
        self.step_series = None
        self.delivered_bytes = 0
        self.deliveries = 0
        self.refused = 0                                 # times a hand-over was put off because HBM had no room for it
        self.refused_by_the_library = 0                  # ... of which by aa_submit_frames itself (AA_ERR_NO_MEMORY at the context's limit)
        self.urgent_groups = 0                           # groups whose key frames took the host route because they were needed at once

    def _deliver(self, ds, f):
        """Frame f of every decoder of the group -> the pinned ring, behind its reconstruction, beside the next frame's: ONE
        gather kernel + ONE copy for the whole frame index (aa_download_batch_async)."""
        env = self.env
        ring = env["deliver_ring"]
        slab = ring[self.deliveries % len(ring)]; self.deliveries += 1
        self.ctx.download_wait(len(ring) - 1)       # the copy that used this slab of the ring before is through (not the one queued a frame index ago)
        self.ctx.download_batch_async(ds, [f] * len(ds), slab, env["raster_bytes"])
        self.delivered_bytes += len(ds) * env["raster_bytes"]

    def _submit_keys(self, g, urgent=False):
        """urgent: the group is the one the pipeline is about to reconstruct (it starts empty): its key frames are needed NOW, and a key
        frame is 35 ms on a host core but a 2.4-s chain on a GPU lane -- such a hand-over asks for the host route (AA_SUBMIT_HOST: every
        frame of the call parsed by host workers into shared arenas).  Key frames handed over steps ahead of their turn take the
        library's default route (the lanes, but for the share the host can take within host_share_ms)."""
        t = time.perf_counter()
        env = self.env
        ds = self.groups[g] = [self.aa.Decoder(self.ctx, env["width"], env["height"]) for _ in range(self.n)]
        for i, d in enumerate(ds):
            self.key_arr[i].stream = d.h.value
        try:
            self.ctx.submit_prepared((self.key_arr, self.key_out, None), env["threads"], False, "host" if urgent and env.get("urgent_keys_on_host") else "auto")
        except self.aa.AlfalfaError as e:
            # the context's memory limit leaves no room for this hand-over's arena (a hard limit since round 5; the call appends
            # nothing then and can be repeated): "not now" -- reconstruct and release first, like a hand-over _room() puts off
            if e.kind != "NoMemory" or self.keys == self.decoded:
                raise
            del self.groups[g], ds
            self.refused += 1; self.refused_by_the_library += 1
            self.host_s += time.perf_counter() - t
            return False
        if urgent and env.get("urgent_keys_on_host"):
            self.urgent_groups += 1
        self.frames_submitted += self.n
        self.host_s += time.perf_counter() - t
        return True

    def _submit_inters(self, g, defer_tokens=False):
        F = self.F
        if F < 2:
            return
        t = time.perf_counter()
        for i, d in enumerate(self.groups[g]):
            h = d.h.value
            for k in range(F - 1):
                self.inter_arr[i * (F - 1) + k].stream = h
        try:
            self.ctx.submit_prepared((self.inter_arr, self.inter_out, None), self.env["threads"], defer_tokens)
        except self.aa.AlfalfaError as e:
            if e.kind != "NoMemory" or g == self.decoded:          # (the group about to be reconstructed must go: nothing would free memory otherwise)
                raise
            self.refused += 1; self.refused_by_the_library += 1
            self.host_s += time.perf_counter() - t
            return False
        self.frames_submitted += self.n * (F - 1)
        self.host_s += time.perf_counter() - t
        return True

    def decode(self, release=True):
        g = self.decoded
        ds = self.groups[g]
        F, ctx, env = self.F, self.ctx, self.env
        keep = set(env["distinct"]) if (g == self.keep_group and release) else ()
        for f in range(F):
            t = time.perf_counter()
            ctx.decode_batch(ds, [f] * self.n)
            t1 = time.perf_counter(); self.t_decode += t1 - t
            if env.get("deliver_ring") and release:
                self._deliver(ds, f)
                t1 = time.perf_counter()
            if release:             # this frame is consumed: its records go back to the pool once the kernels queued so far
                for i, d in enumerate(ds):        # have run, its raster when nothing refers to it any more (RasterHandle semantics)
                    if i not in keep:
                        d.release_before(f + 1)
                self.t_release += time.perf_counter() - t1
        if release:
            t1 = time.perf_counter()
            if keep:
                self.kept[g] = ds
            del self.groups[g], ds  # the chunk is done: its decoders go (nothing waits for the GPU here)
            self.t_release += time.perf_counter() - t1
        self.decoded += 1
        self.done_t.append(time.perf_counter())
        if self.step_series is not None:          # what the host waited for, step by step (diagnostics of the timed region)
            # (aa_ctx_get_info only: it looks at the books without waiting for the GPU.  aa_ctx_kernel_stats synchronises with the compute
            # and parse streams -- called here, as the first version of this series did, it made every step wait for its own
            # reconstruction: the "plateau" of 450 ms per step in the first round-5 sessions was this line)
            t_info = time.perf_counter()
            i = ctx.info()
            self.t_room += time.perf_counter() - t_info
            self.step_series.append((round((self.t_decode - self.t_decode_mark) * 1e3), i["token_workgroups_alive"], i["jobs_waiting"],
                                     round(i["heap_used_bytes"] / 1e9, 1), round((i["pool_bytes"] - i["pool_free_bytes"]) / 1e9, 1),
                                     i["host_waited_parse_ms"] - self.wait_mark[0], i["host_waited_compute_ms"] - self.wait_mark[1]))
            self.t_decode_mark = self.t_decode; self.wait_mark = (i["host_waited_parse_ms"], i["host_waited_compute_ms"])
        if env["args"].trace_memory:
            i = ctx.info()
            print("step %d: pool %.1f GB (free %.1f, pending %.1f) heap mapped %.1f used %.1f GB free chunks %d starved %d alive wgs %d waiting %d refused %d"
                  % (self.decoded, i["pool_bytes"] / 1e9, i["pool_free_bytes"] / 1e9, i["pool_pending_bytes"] / 1e9, i["heap_mapped_bytes"] / 1e9,
                     i["heap_used_bytes"] / 1e9, i["heap_free_chunks"], i["lanes_starved"], i["token_workgroups_alive"], i["jobs_waiting"], self.refused), file=sys.stderr, flush=True)

    def _room(self, nframes, coeff_bytes, arena_bytes):
        """Is there room in HBM for `nframes` more frames in flight?  Both halves of the context's memory count: the POOL (batch
        arenas with the compressed frames and the macroblock records, rasters, the transient dense blocks of a reconstruction
        call) and the coefficient HEAP (frames in flight are on the context's books with what they are expected to store, parsed
        ones with what they took).  The pool's live pieces + this hand-over's arena + what the next reconstruction calls need,
        plus the heap's expected content (over-committed: frames being parsed hold only part of what they will, and the oldest
        are released first) must fit the limit.  Never refuses when nothing is in flight: waiting would free nothing."""
        if self.keys == self.decoded:
            return True
        env = self.env
        t_info = time.perf_counter()
        i = self.ctx.info()
        self.t_room += time.perf_counter() - t_info
        limit, oc = i["memory_limit_bytes"], env["args"].overcommit
        # the pool never gives memory back to the device and the heap never unmaps: what the HEAP can still get is what the pool has
        # not taken (pool_bytes, free lists included), what the POOL can still get is what the heap has not mapped
        heap_after = i["heap_used_bytes"] + coeff_bytes
        heap_cap = min(i["heap_limit_bytes"] or limit, limit - i["pool_bytes"])
        # (pieces released and waiting for kernels already queued -- "pending" -- are free by the time new frames need the room)
        pool_live = i["pool_bytes"] - i["pool_free_bytes"] - i["pool_pending_bytes"]
        pool_after = pool_live + arena_bytes + env["recon_reserve"]
        pool_cap = limit - max(i["heap_mapped_bytes"], heap_after / oc)
        ok = heap_after <= oc * heap_cap and pool_after <= pool_cap
        if not ok:
            self.refused += 1
        return ok

    def run(self, steps):
        """`steps` whole steps, from an empty pipeline to an empty pipeline.  Frames go to the GPU parser in the order they
        are needed, as far ahead as the look-ahead says AND the HBM budget holds: key frames up to K steps before their
        group is reconstructed (their chains are the long ones), inter frames up to D steps."""
        env, F = self.env, self.F
        target = self.decoded + steps
        if env.get("urgent_keys_on_host") and self.keys == self.decoded and self.decoded < target:
            # An EMPTY pipeline.  The key frames of the group it starts with are needed at once and are the longest chains there are
            # (2 s on a GPU lane, 35 ms on a host core): they go to the context's HOST LANES (AA_SUBMIT_HOST; the call does not wait
            # for them).  The loop below then hands that group's inter frames and the later groups' key frames to the GPU lanes as ever.
            # The SECOND group's too: the host's cores get through a group in ~1 s, a key frame handed to a GPU lane now is through in
            # ~2.9 s (header kernel + a 2.2-s chain) -- the second group is ready at 2.1 s instead of 3, and from the third on the lanes'
            # key frames arrive as fast as the host's would.
            g0 = self.decoded
            for g in range(g0, min(target, g0 + env.get("urgent_groups", env["args"].urgent_groups))):
                self._submit_keys(g, urgent=True)
                self.keys = g + 1
        while self.decoded < target:
            while True:
                can_inter = self.inter_h < min(target, self.decoded + self.D + self.H, self.keys)
                can_key = self.keys < min(target, self.decoded + self.K)
                # the group about to be reconstructed comes first; otherwise key frames lead (K - D steps ahead of the inter frames)
                if can_inter and (self.inter_h == self.decoded or not can_key or self.keys - self.inter_h > self.K - self.D):
                    if self.inter_h != self.decoded and not self._room(self.n * (F - 1), self.n * (F - 1) * env["inter_coeff_bytes"], self.n * (F - 1) * env["inter_arena_bytes"]):
                        break
                    if self._submit_inters(self.inter_h, defer_tokens=self.H > 0) is False:
                        break
                    self.inter_h += 1
                    if self.H == 0:
                        self.inters += 1
                elif can_key:
                    if env.get("keys_on_host"):      # (dense records in a pool piece, nothing in the coefficient heap)
                        fits = self._room(self.n, 0, self.n * (env["key_arena_bytes"] + env["key_dense_bytes"]))
                    else:
                        fits = self._room(self.n, self.n * env["key_coeff_bytes"], self.n * env["key_arena_bytes"])
                    if not fits:
                        break
                    if self._submit_keys(self.keys, urgent=self.keys == self.decoded) is False:
                        break
                    self.keys += 1
                elif self.H > 0 and self.inters < min(target, self.decoded + self.D, self.inter_h):
                    t = time.perf_counter()
                    self.ctx.launch_tokens(1); self.inters += 1
                    self.t_launch += time.perf_counter() - t
                else:
                    break
            self.decode()


T_START = time.time()


def log(msg):
    """Progress on stderr (a run that is cut off by a timeout still says how far it got)."""
    print("[bench %7.1f s] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


def calibrate(env, streams):
    """What ONE lone chain costs, and what frames of this content really store:
    (a) a key frame parsed with nothing else on the GPU: its wall time / its decode steps = the latency of one step of one
        lane, the unit of the entropy decode's own roof (lanes / step latency);
    (b) coefficient blocks (and packed words) per key / inter frame of a few streams -> bytes a frame in flight holds."""
    import alfalfa_amd as aa
    ctx, F, width, height, threads = env["ctx"], env["F"], env["width"], env["height"], env["threads"]
    S = len(streams)
    mbs_per_frame = env["mbs_per_frame"]
    cal = [aa.Decoder(ctx, width, height) for _ in range(min(4, S))]
    ctx.kernel_stats(reset=True)
    t0 = time.perf_counter()
    ctx.submit_frames([(cal[0], streams[0][0])], threads, route="device")
    key_hdr = cal[0].frame_header(0)                     # waits for the parse
    t_lone_key = time.perf_counter() - t0
    lone_stats = ctx.kernel_stats(reset=True)
    lone_steps = lone_stats["token_steps"]
    ctx.submit_frames([(d, st[0]) for d, st in zip(cal[1:], streams[1:])] + [(d, fr) for d, st in zip(cal, streams) for fr in st[1:]], threads, route="device")
    key_blocks = max([key_hdr["num_coeff_blocks"]] + [d.frame_header(0)["num_coeff_blocks"] for d in cal[1:]])
    inter_blocks = max(d.frame_header(f)["num_coeff_blocks"] for d in cal for f in range(1, F)) if F > 1 else key_blocks
    ctx.sync()
    rest = ctx.kernel_stats(reset=True)
    env["step_latency_us"] = t_lone_key / max(1, lone_steps) * 1e6
    env["urgent_keys_on_host"] = False            # (decided below, once the host's real rate is known)
    env["lone_key_s"] = t_lone_key
    packed = bool(ctx.info()["packed_coefficients"])
    key_bpb = inter_bpb = 32.0
    if packed:
        # bytes a stored block takes, key and inter frames apart (a key frame's blocks are nearly full, an inter frame's hold 2-4 values)
        key_bpb = 2.0 * lone_stats["packed_words"] / max(1, lone_stats["packed_blocks"])
        rest_keys = len(cal) - 1
        inter_words = rest["packed_words"] - rest_keys * lone_stats["packed_words"]
        inter_blks = rest["packed_blocks"] - rest_keys * lone_stats["packed_blocks"]
        inter_bpb = 2.0 * inter_words / inter_blks if inter_blks > 0 and inter_words > 0 else key_bpb
    del cal
    # per frame in flight: the coefficient heap (its blocks + the partly filled last chunk) and the pool (macroblock records, flags,
    # chunk list, packed positions, the compressed frame -- device half of the batch arena)
    comp = env["compressed_bytes"] / max(1, S * F)
    rec = mbs_per_frame * (80 + 1 + (4 if packed else 0)) + 8192
    env["key_coeff_bytes"], env["inter_coeff_bytes"] = int(key_blocks * key_bpb + 65536), int(inter_blocks * inter_bpb + 65536)
    env["key_arena_bytes"], env["inter_arena_bytes"] = int(rec + 1.1 * comp), int(rec + 1.1 * comp)
    env["key_dense_bytes"] = key_blocks * 32
    # does a hand-over of S key frames fit the host share?  The library plans with what its host workers REALLY got through in the calls
    # so far (cores a process sees and cores it gets differ under a CPU quota): give it one hand-over of key frames to measure on
    hs = ctx.info()["host_share_ms"]
    env["keys_on_host"] = False
    env["host_rate_kb_per_ms"] = 0
    key_bytes_total = sum(len(st[0]) for st in streams)
    # (only where the library's own assumption -- 24 KB/ms per usable core -- lets the host take at least half of a hand-over: else it
    # takes none, there is nothing to measure, and 2 x S key frames would go through the lanes for nothing)
    host_lanes = max(1, min(int(os.environ.get("ALFALFA_AMD_HOST_LANES") or 0) or aa.capi.lib().aa_host_cpus(), aa.capi.lib().aa_host_cpus()))      # (this rank's share: run())
    if hs > 0 and S > min(threads, 24) and hs * 24.0e3 * host_lanes >= 0.5 * key_bytes_total:
        for _ in range(2):
            probe = [aa.Decoder(ctx, width, height) for _ in range(S)]
            ctx.submit_frames([(d, st[0]) for d, st in zip(probe, streams)], threads)
            for d in probe:
                d.frame_header(0)
            ctx.sync()
            del probe
        env["host_rate_kb_per_ms"] = ctx.info()["host_rate_kb_per_ms"]
        env["keys_on_host"] = bool(0.9 * key_bytes_total <= hs * env["host_rate_kb_per_ms"] * 1e3)
    # The key frames of the group a pipeline STARTS with on the host route (AA_SUBMIT_HOST): worth it when the host gets through them
    # in well under a key-frame chain -- half a second --, because the call blocks the thread that feeds the pipeline.  Measured on a
    # box that grants 16 CPUs (1.3 s for 480 key frames): first step at 3.0 s instead of 3.9, but the hand-overs behind it start 1.3 s
    # late and the run as a whole is no faster (profiles/r04_bench_sessions.md).
    # (no host batch measured: the library's own assumption then, 24 KB/ms per usable core)
    rate_kb_per_ms = env["host_rate_kb_per_ms"] or 24.0 * host_lanes
    est_ms = key_bytes_total / (rate_kb_per_ms * 1e3)
    # Round 5: that route no longer blocks (HOST LANES: the frames take the device route's pre-pass and arena, worker threads of the
    # context finish them with the `done` word a GPU lane writes) -- the first group's key frames go to the host's cores whenever there
    # are more streams than a small call has, while the lanes take that group's inter frames and the later groups' key frames.
    # How many groups: as many as the host lanes finish before the GPU lanes deliver their first key frames anyway (a key frame's
    # chain beside a full GPU: ~2.8 s) -- two on a rank with 16 cores (1.0 s per group of 480), none on a rank that gets 2 of them.
    env["urgent_groups"] = max(0, min(env["args"].urgent_groups, int(2800.0 / max(est_ms, 1.0))))
    env["urgent_keys_on_host"] = bool(S > 24 and not env["args"].no_urgent_host and env["urgent_groups"] > 0)
    env["urgent_host_estimate_ms"] = round(est_ms)
    ctx.kernel_stats(reset=True)
    env["packed_storage"] = {"key_frame_bytes_per_block": round(key_bpb, 2), "inter_frame_bytes_per_block": round(inter_bpb, 2), "dense_bytes_per_block": 32} if packed else None
    env["planned"] = {"key_frame_heap_bytes": env["key_coeff_bytes"], "inter_frame_heap_bytes": env["inter_coeff_bytes"], "frame_pool_bytes": env["key_arena_bytes"],
                      "key_blocks_per_mb": round(key_blocks / mbs_per_frame, 2), "inter_blocks_per_mb": round(inter_blocks / mbs_per_frame, 2)}
    # what the reconstruction calls of the next steps take from the pool while frames wait: ~3 rasters per stream in flight
    # (output + references being replaced) and two calls' transient dense blocks
    env["recon_reserve"] = int(S * (3 * env["raster_bytes"] + 2 * (inter_blocks * 32 if packed else 0)))


def token_profile_delta(before, after, clock_mhz):
    """The worker waves' own time accounting (k_token_workers, ALFALFA_AMD_TOKEN_PROFILE): 100 MHz ticks summed over the waves."""
    d8 = [a - b for a, b in zip(after, before)]
    tot = d8[3] + d8[4] + d8[5]
    if not d8[7] or tot <= 0:
        return None
    us_step = (d8[5] - d8[0]) / max(1, d8[2]) / 100.0
    return {"wave_seconds": round(tot / 1e8, 2), "frac_steps": round((d8[5] - d8[0]) / tot, 3), "frac_boundary_passes": round(d8[0] / tot, 3),
            "frac_top_up": round(d8[4] / tot, 3), "frac_take_and_begin": round(d8[3] / tot, 3),
            "us_per_wave_step": round(us_step, 4), "cycles_per_wave_step_at_%d_mhz" % clock_mhz: round(us_step * clock_mhz),
            "us_per_boundary_pass": round(d8[0] / max(1, d8[1]) / 100.0, 3), "wave_steps": d8[2], "boundary_passes": d8[1],
            "lanes_with_frame_per_period": round(d8[6] / max(1, d8[7]), 2), "steps_per_period": round(d8[2] / max(1, d8[7]), 2),
            "note": "time of the worker waves while they held frames (waves that linger without work are asleep and not counted)"}


def verify_against_reference(env, decoders_by_index, paths, frames_to_check):
    """Rasters of decoders (index in the stream list -> Decoder, every frame still held) against the REFERENCE decoder run here
    (oracle/_ref/ref_decode): SHA-256 of the three padded planes.  -> dict, or None when the reference binary is not there."""
    import workload
    ref_decode = os.path.join(ROOT, "oracle", "_ref", "ref_decode")
    if not os.path.exists(ref_decode):
        return None
    F = env["F"]

    def ref_hashes(i):
        raw = os.path.join(workload.cache_dir(), "bench_verify_%d_%d.raw" % (os.getpid(), i))
        subprocess.run([ref_decode, paths[i], raw], check=True, stdout=subprocess.DEVNULL)
        with open(raw, "rb") as fh:
            ref = fh.read()
        os.unlink(raw)
        fs = len(ref) // F
        return i, [hashlib.sha256(ref[f * fs:(f + 1) * fs]).digest() for f in frames_to_check]
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        refs = list(ex.map(ref_hashes, sorted(decoders_by_index)))
    bad = [(i, f) for i, hs in refs for f, hsh in zip(frames_to_check, hs)
           if hashlib.sha256(decoders_by_index[i].raster_bytes(f)).digest() != hsh]
    if bad:
        raise SystemExit("PARITY FAILURE (%s): HIP output differs from the reference decoder on (stream, frame) %r" % (env["config"], bad[:8]))
    return {"streams_checked": len(refs), "frames_per_stream": len(frames_to_check), "bit_exact": True}


def make_env(args, ctx, config, S, F, rank, world, threads):
    import alfalfa_amd as aa
    import workload
    from alfalfa_amd import sharding
    width, height = workload.CONFIGS[config][:2]
    # stream ids are disjoint across ranks; their CONTENT is drawn from a pool of synthetic videos so that the one-off
    # generation cost (reference encoder, cached on disk) stays bounded on an 8-GPU node
    synth = workload.CONFIGS[config][2] == "synth"       # (the pure-Python stream writer is slow: few distinct streams)
    pool = 8 if synth else (120 if config == args.config else 24)
    seeds = [100 + (g - 100) % pool for g in sharding.stream_ids(rank, world, S)]
    t0 = time.time()
    paths = workload.make_streams(config, F, seeds)
    t_gen = time.time() - t0
    streams = [aa.read_ivf(p)[2] for p in paths]
    distinct = {}
    for i, sd in enumerate(seeds):
        distinct.setdefault(sd, i)
    env = {"args": args, "ctx": ctx, "config": config, "width": width, "height": height, "F": F, "S": S, "threads": threads,
           "seeds": seeds, "paths": paths, "streams": streams, "t_gen": t_gen, "distinct": sorted(distinct.values()),
           "mbs_per_frame": ((width + 15) // 16) * ((height + 15) // 16),
           "compressed_bytes": sum(len(fr) for st in streams for fr in st), "deliver_ring": None}
    env["mbs_per_step"] = S * F * env["mbs_per_frame"]
    probe = aa.Decoder(ctx, width, height)
    env["plane_sizes"] = probe.plane_sizes()
    env["raster_bytes"] = sum(env["plane_sizes"])
    del probe
    return env


def run_secondary(args, ctx, config, rank, world, threads):
    """A BASELINE config other than the headline one, end to end, a few steps from an empty pipeline to an empty pipeline, the last
    step's rasters checked against the reference decoder in the run."""
    F = 6 if config.endswith("_subpel") else args.frames
    S = args.secondary_streams
    env = make_env(args, ctx, config, S, F, rank, world, threads)
    calibrate(env, env["streams"])
    steps = args.secondary_steps
    pipe = Pipeline(env, env["streams"], min(args.key_ahead, steps), min(args.depth, steps))
    pipe.run(1)                              # (pools of this geometry reach their size)
    ctx.sync()
    pipe.keep_group = pipe.decoded + steps - 1
    ctx.kernel_stats(reset=True)
    t0 = time.perf_counter()
    pipe.run(steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    st = ctx.kernel_stats(reset=True)
    kept = pipe.kept.get(pipe.keep_group)
    verified = None
    if rank == 0 and not args.no_verify and kept:
        verified = verify_against_reference(env, {i: kept[i] for i in env["distinct"]}, env["paths"], sorted({0, F // 2, F - 1}))
    import workload
    cfg = workload.CONFIGS[config]
    out = {"value": round(S * F * env["mbs_per_frame"] * steps / dt, 1), "unit": "macroblocks/s", "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
           "workload": "%d streams x %d frames of %dx%d (%s, y_ac_qi %d, loop filter %d), end to end, empty pipeline to empty pipeline"
                       % (S, F, env["width"], env["height"], "all key frames" if config.endswith("_intra") else "1 key + %d inter" % (F - 1), cfg[3], cfg[4]),
           "macroblocks_per_step": S * F * env["mbs_per_frame"], "compressed_bytes_per_mb": round(env["compressed_bytes"] / (S * F * env["mbs_per_frame"]), 2),
           "lone_key_frame_parse_s": round(env["lone_key_s"], 4), "token_steps_per_mb": round(st["token_steps"] / max(1, st["parsed_macroblocks"]), 1),
           "distinct_streams": len(env["distinct"]), "verified_bit_exact_vs_reference": verified, "stream_generation_s": round(env["t_gen"], 1),
           "frames_handed_back_for_lack_of_memory": st["nomem_retries"], "heap_grows": st["heap_grows"], "host_waited_for_parse_ms_per_step": round(st["parse_wait_ms"] / steps, 1)}
    if len(pipe.done_t) > steps and steps > 1:
        d = pipe.done_t[-steps:]
        out["between_fill_and_drain_value"] = round(S * F * env["mbs_per_frame"] * (steps - 1) / (d[-1] - d[0]), 1)
    del pipe, kept
    # the reference CPU decoder on a stream of THIS config (north_star: 720p and 1080p "next to the reference CPU path"): a bounded
    # sample, ~4 s of one core
    ref_time = os.path.join(ROOT, "oracle", "_ref", "ref_time")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and os.path.exists(ref_time):
        try:
            reps = max(1, int(round(4.0 / (F * env["mbs_per_frame"] / 150000.0))))
            r = json.loads(subprocess.run([ref_time, env["paths"][0], str(reps)], check=True, capture_output=True, text=True, timeout=120).stdout)
            out["cpu_baseline"] = {"value": round(r["mb_per_s"], 1), "unit": "macroblocks/s", "cores": 1, "kind": "reference",
                                   "sample": "the first stream of this config (%d frames %dx%d) decoded %d times by oracle/_ref/ref_time; reference built without x86 asm"
                                             % (F, env["width"], env["height"], reps)}
        except Exception as e:
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def lane_per_partition_leg(args, config, rank, world, threads):
    """The multi-partition config once more in a context of its own with ONE LANE PER DCT PARTITION (aa_ctx_set_lane_per_partition: a
    per-context switch, off by default -- the shared above-row flags cost every frame of the context 128 bytes of LDS per lane and the
    kernel 10 registers): the same streams, steps and check as the `secondary` figure beside it."""
    import alfalfa_amd as aa
    ctx2 = aa.Context(int(os.environ.get("LOCAL_RANK", "0")))
    ctx2.set_memory_limit(int(32e9))
    ctx2.set_lane_per_partition(True)
    try:
        r = run_secondary(args, ctx2, config, rank, world, threads)
        r["lane_per_partition"] = bool(ctx2.info()["lane_per_partition"])
        r.pop("cpu_baseline", None)
        return r
    finally:
        ctx2.sync()
        del ctx2


def two_cpu_leg(args):
    """The headline workload in a child process that has TWO CPUs (sched_setaffinity: aa_host_cpus() then says 2, so the context gets 2
    host lanes and the pre-pass 2 threads): the host budget of one rank when eight ranks share the 16 CPUs this pool's boxes grant
    (the N = 1 headline leans on 16 CPUs that eight ranks will not have).  Run BEFORE this process takes its HBM."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.two_cpu_steps), "--warmup", "2", "--threads", "2", "--config", args.config,
           "--streams", str(args.streams), "--frames", str(args.frames), "--key-ahead", str(args.key_ahead), "--depth", str(args.depth), "--hbm-gb", str(args.hbm_gb),
           "--secondary=", "--small-batches=", "--no-cpu-baseline", "--no-verify", "--no-device-half", "--lanes-only-steps", "0", "--deliver-steps", "0", "--two-cpu-steps", "0"]
    env = dict(os.environ)
    env["AA_BENCH_CPUS"] = "2"; env["ALFALFA_AMD_HOST_LANES"] = "2"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "AA_BENCH_FORCE_DIST"):
        env.pop(k, None)
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:                          # (must not take the headline down)
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    t = d.get("timed_region") or {}
    return {"value": d["value"], "unit": "macroblocks/s", "steps": d["steps"], "ms_per_step": d["ms_per_step"], "cpus": 2, "host_lanes": 2, "host_threads": 2,
            "first_step_done_at_ms": (t.get("step_done_at_ms") or [None])[0], "between_fill_and_drain_value": (d.get("steady_state") or {}).get("value"),
            "host_prepass_and_staging_ms_per_step": round((d.get("stages") or {}).get("host_prepass_and_staging_s_per_step", 0) * 1e3, 1),
            "frames_parsed_on_host_cores": t.get("frames_parsed_on_host_cores"), "leg_wall_s": round(time.perf_counter() - t0, 1),
            "note": "a child process of bench.py confined to 2 CPUs (empty pipeline to empty pipeline over fewer steps than `value`: the fill weighs more); "
                    "not verified against the reference in the child (the parent's run is)"}


def main():
    args = parse_args()
    if os.environ.get("AA_BENCH_CPUS"):             # (the 2-CPU leg's child: before anything counts cores or starts threads)
        os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:int(os.environ["AA_BENCH_CPUS"])]))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")); local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    dist = None
    if world > 1 or os.environ.get("AA_BENCH_FORCE_DIST"):     # the env switch lets a 1-GPU box exercise the RCCL path
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dev_index = int(os.environ.get("AA_BENCH_DEVICE", local_rank))     # (a dry run of N ranks on ONE GPU: AA_BENCH_DEVICE=0)
        torch.cuda.set_device(dev_index)
        backend = os.environ.get("AA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    dev_index = int(os.environ.get("AA_BENCH_DEVICE", local_rank))

    import alfalfa_amd as aa
    import workload
    from alfalfa_amd import sharding

    S, F = args.streams, args.frames
    # host workers of the header pre-pass: this rank's share of the cores (measured on the 256-core box: with the 32 workers an
    # 8-GPU node leaves a rank, a step's pre-pass + staging takes 27 ms and the end-to-end rate is within run-to-run noise of
    # the 256-worker rate)
    host_cpus = aa.capi.lib().aa_host_cpus()                # (what the process can really use: the cgroup quota counts -- the round-4 box shows 256, grants 16)
    threads = args.threads or max(1, min(os.cpu_count() or 1, 2 * host_cpus) // max(1, local_world))
    # ... and of the context's HOST LANES (worker threads that finish key frames on the host's cores): a rank gets its share of the
    # granted CPUs, not all of them -- eight ranks with sixteen lanes each on a 16-CPU grant would plan with eight times the cores there
    # are (the library sizes its lanes, and the rate it plans the host share with, from this variable; a value already set stands)
    rank_cpus = max(1, host_cpus // max(1, local_world))
    os.environ.setdefault("ALFALFA_AMD_HOST_LANES", str(rank_cpus))

    ctx = aa.Context(dev_index)
    ctx.set_schedule(args.schedule)
    hbm_budget = min(args.hbm_gb * 1e9, 0.9 * ctx.memory()[0])
    # (the library refuses hand-overs by what the context HOLDS; what it has TAKEN from the device can sit a few hundred MB above that --
    # free pieces of sizes nobody asks for again are not re-split -- so the limit it is given is the budget less 1 GB, and `memory` in the
    # line checks the budget itself)
    ctx.set_memory_limit(int(hbm_budget - 1e9))
    if args.dense:
        ctx.set_packed_coefficients(False)
    if args.host_share_ms is not None:
        ctx.set_host_share_ms(args.host_share_ms)
    env = make_env(args, ctx, args.config, S, F, rank, world, threads)
    width, height, streams, paths, seeds = env["width"], env["height"], env["streams"], env["paths"], env["seeds"]
    mbs_per_frame, mbs_per_step, compressed_bytes = env["mbs_per_frame"], env["mbs_per_step"], env["compressed_bytes"]
    t_gen = env["t_gen"]

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # ---- multi-GPU only: one-shot entry-state hand-off (outside the timed region).  Rank 0 decodes the head (key frame)
    # of a shared GOP; its DecoderState blob and reference raster are broadcast (RCCL over xGMI) and every rank continues
    # the GOP from the imported state; all ranks must end on the same raster as a straight decode. ----
    handoff = None
    if dist is not None:
        import torch
        tdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
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
        if tdev == "cuda":
            dist.broadcast(planes, src=0)
        else:                                                   # (gloo dry run: through host memory)
            hp = planes.cpu(); dist.broadcast(hp, src=0); planes.copy_(hp)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        blob = sharding.broadcast_bytes(dist, blob, 0, device=tdev)
        cont.import_state(blob)
        cont.import_reference_device(base, base + ysz, base + ysz + usz)
        last = None
        for fr in shared[1:]:
            _, last = cont.get_frame_output(fr)
        digest = sharding.sha256(cont.raster_bytes(last))
        agree = sharding.digests_agree(dist, digest, device=tdev)
        if rank == 0:
            straight = aa.Decoder(ctx, width, height)
            for fr in shared:
                _, fi = straight.get_frame_output(fr)
            agree = agree and sharding.sha256(straight.raster_bytes(fi)) == digest
        handoff = {"raster_bytes": ysz + usz + vsz, "state_bytes": len(blob), "broadcast_ms": round(t_bcast * 1e3, 3), "backend": dist.get_backend(),
                   "world_size": world, "continuations_agree": bool(agree)}
        if not agree:
            raise SystemExit("entry-state hand-off mismatch across ranks")

    # ---- the end-to-end pipeline ----
    log("streams ready (%d distinct, generated in %.1f s)" % (len(env["distinct"]), t_gen))
    two_cpus = None
    if rank == 0 and world == 1 and dist is None and args.two_cpu_steps > 0:
        two_cpus = two_cpu_leg(args)                # (now: the streams are in the cache, and this process holds no HBM to speak of yet)
        log("2-CPU leg: %s" % ({k: two_cpus.get(k) for k in ("value", "first_step_done_at_ms", "between_fill_and_drain_value", "leg_wall_s", "error")},))
    calibrate(env, streams)
    log("calibrated: lone key frame %.3f s, step %.4f us, keys on host: %s, planned %s" % (env["lone_key_s"], env["step_latency_us"], env["keys_on_host"], env["planned"]))
    step_latency_us = env["step_latency_us"]
    plane_sizes, raster_bytes = env["plane_sizes"], env["raster_bytes"]
    if args.deliver:
        env["deliver_ring"] = [ctx.pinned_alloc(S * raster_bytes) for _ in range(DELIVER_RING)]
    K, D = max(1, args.key_ahead), max(1, min(args.depth, args.key_ahead))
    # what the look-ahead would hold if memory were free (the run itself is bounded by Pipeline._room)
    planned_need_gb = round(S * ((0.5 * K + 1.5) * (env["key_coeff_bytes"] + env["key_arena_bytes"]) + (0.6 * D + 1.0) * (F - 1) * (env["inter_coeff_bytes"] + env["inter_arena_bytes"])
                                 + 5 * raster_bytes) / 1e9, 1)
    pipe = Pipeline(env, streams, K, D, args.header_ahead)
    log("priming")
    pipe.run(max(2, pipe.K + 1))        # priming (K + 1 steps: the key-frame look-ahead reaches its full depth, as it does in the timed region) (untimed, before the warm-up): the pools and the coefficient heap reach their working
    pipe.run(args.warmup)               # size, so that first-touch allocations (hipMalloc / hipMemMap / hipHostMalloc) are not what the steps measure
    barrier()
    log("warm-up done; timed region starts")
    pipe.host_s = pipe.t_launch = pipe.t_decode = pipe.t_release = 0.0; pipe.done_t = []; pipe.refused = pipe.refused_by_the_library = 0; pipe.urgent_groups = 0
    pipe.step_series = []; pipe.t_room = 0.0; pipe.t_decode_mark = 0.0; pipe.wait_mark = (0, 0)
    pipe.keep_group = pipe.decoded + args.steps - 1          # the last TIMED step keeps the frames of its distinct streams: they are what is verified
    ctx.kernel_stats(reset=True)
    prof0 = ctx.info()["token_profile"]
    if args.profile_timed:
        ctx.profile(True)
    t0 = time.perf_counter()
    delivered0 = pipe.delivered_bytes
    pipe.run(args.steps)
    ctx.sync()
    elapsed = time.perf_counter() - t0
    log("timed region: %.3f s for %d steps, steps done at %s ms, hand-overs put off %d" % (elapsed, args.steps, [round((t - t0) * 1e3) for t in pipe.done_t], pipe.refused))
    delivery = None
    if args.deliver:
        delivery = {"bytes_per_step": (pipe.delivered_bytes - delivered0) // args.steps, "gb_per_s": round((pipe.delivered_bytes - delivered0) / elapsed / 1e9, 2),
                    "copies_per_frame_index": 1,
                    "note": "every reconstructed frame gathered into one staging piece and copied to pinned host memory on the copy stream inside the timed region "
                            "(%.2f MB per frame); `value` of THIS run includes it" % (raster_bytes / 1e6)}
    hbm_free, hbm_total = ctx.memory()
    urgent_groups_timed = pipe.urgent_groups
    tstats = ctx.kernel_stats(reset=True); ctx.profile(False)
    info = ctx.info()
    clock_mhz = info.get("clock_mhz") or 2400
    token_profile = token_profile_delta(prof0, info["token_profile"], clock_mhz)
    series = pipe.step_series or []
    pipe.step_series = None
    timed_region = {"step_done_at_ms": [round((t - t0) * 1e3) for t in pipe.done_t], "host_ms_per_step_in_aa_ctx_get_info": round(pipe.t_room / args.steps * 1e3, 1),
                    "per_step": {"what": "after each step: [ms of the step the host spent inside its aa_decode_batch calls (waits for parses and for binding buffers included), "
                                         "worker workgroups alive, jobs waiting in the queue, coefficient heap in use GB, pool in use GB, ms of it waiting for the device parser, ms of it waiting for the compute stream]", "series": [list(x) for x in series]},
                    "host_ms_per_step": {"submit": round(pipe.host_s / args.steps * 1e3, 1), "launch_tokens": round(pipe.t_launch / args.steps * 1e3, 1),
                                         "decode_batch_calls_incl_wait_for_parse": round(pipe.t_decode / args.steps * 1e3, 1),
                                         "release": round(pipe.t_release / args.steps * 1e3, 1)},
                    "host_waited_for_parse_ms_per_step": round(tstats["parse_wait_ms"] / args.steps, 2),
                    "host_waited_for_compute_stream_ms_per_step": round(tstats["bind_wait_ms"] / args.steps, 2),
                    "host_in_pool_allocator_ms_per_step": round(tstats["alloc_ms"] / args.steps, 2), "slab_mallocs": tstats["slab_mallocs"], "row_handoff_rereads_since_context_creation": tstats["row_handoff_rereads"], "of_which_the_poll_repeated_was_still_stale": tstats["row_handoff_stale_polls"],
                    "heap_grows": tstats["heap_grows"], "frames_handed_back_for_lack_of_memory": tstats["nomem_retries"], "frames_evicted": tstats["frames_evicted"],
                    "worker_grids_launched": tstats["worker_launches"], "worker_workgroups_launched": tstats["worker_wgs"], "worker_grids_retired": tstats["worker_retires"],
                    "pool_waits": tstats["pool_waits"], "pool_wait_ms_per_step": round(tstats["pool_wait_ms"] / args.steps, 2),
                    "hand_overs_put_off_for_lack_of_room": pipe.refused, "of_which_refused_by_the_library_at_its_memory_limit": pipe.refused_by_the_library,
                    "frames_parsed_on_host_cores": tstats["host_routed_frames"]}
    # the entropy decode against ITS roof: a lane decodes one bool per step, a step takes what it takes (measured on a lone chain),
    # the GPU holds `lanes` chains -> lanes / step latency bools per second at best
    lanes_total = info["token_lanes_per_workgroup"] * info["token_workgroups_capacity"]
    bools = tstats["token_steps"]
    lanes_roof = {"lanes": lanes_total, "lanes_per_workgroup": info["token_lanes_per_workgroup"], "workgroups_per_cu": info["token_workgroups_capacity"] // max(1, info["compute_units"]),
                  "live_lanes_of_64": round(info["token_lanes_per_workgroup"] / 64.0, 3),
                  "lane_lds_bytes": info["token_lane_lds_bytes"], "workgroup_lds_bytes": info["token_workgroup_lds_bytes"],
                  "step_latency_us_lone_chain": round(step_latency_us, 4),
                  "roof_bools_per_s": round(lanes_total / (step_latency_us * 1e-6)), "sustained_bools_per_s": round(bools / elapsed),
                  "frac": round(bools / elapsed / (lanes_total / (step_latency_us * 1e-6)), 4), "bools_per_step": round(bools / args.steps),
                  "in_kernel_accounting": token_profile,
                  "note": "decode steps of the frames parsed in the timed region (an upper bound on bools) over the timed region's wall time; step latency = lone key frame submit->parsed / its steps"}
    memory = {"limit_gb": round(info["memory_limit_bytes"] / 1e9, 1), "pool_gb": round(info["pool_bytes"] / 1e9, 2), "coefficient_heap_mapped_gb": round(info["heap_mapped_bytes"] / 1e9, 2),
              "hbm_taken_by_the_context_gb": round((info["pool_bytes"] + (info["heap_mapped_bytes"] if info["heap_is_virtual"] else 0)) / 1e9, 2),
              "inside_the_budget": bool(info["pool_bytes"] + (info["heap_mapped_bytes"] if info["heap_is_virtual"] else 0) <= hbm_budget), "budget_gb": round(hbm_budget / 1e9, 1),
              "pinned_host_gb": round(info["pinned_host_bytes"] / 1e9, 2), "heap_is_virtual": bool(info["heap_is_virtual"]),
              "hbm_in_use_on_device_gb": round((hbm_total - hbm_free) / 1e9, 1),
              "packed_storage": env["packed_storage"], "planned": env["planned"]}
    if args.profile_timed:
        timed_region["kernel_ms_per_step"] = {k: round(v / args.steps, 2) for k, v in tstats.items() if k.endswith("_ms") and "wait" not in k}
    host_submit_s = pipe.host_s / max(1, args.steps)
    # between fill and drain: the MEAN interval between reconstruction hand-overs after the first one (host side, i.e. when the
    # parse a step waited for was done) -- the timed region itself also pays for filling and draining the pipeline
    steady_ms = (pipe.done_t[-1] - pipe.done_t[0]) / (len(pipe.done_t) - 1) * 1e3 if len(pipe.done_t) > 1 else None
    per_rank = None
    if dist is not None:
        import torch
        tdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([elapsed, pipe.host_s / max(1, args.steps), tstats["parse_wait_ms"] / max(1, args.steps) * 1e-3, float(threads),
                             info["pool_bytes"] + (info["heap_mapped_bytes"] if info["heap_is_virtual"] else 0), info["pinned_host_bytes"]],
                            dtype=torch.float64, device=tdev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "elapsed_s": round(float(v[0]), 4), "host_prepass_and_staging_s_per_step": round(float(v[1]), 4),
                     "host_waited_for_parse_s_per_step": round(float(v[2]), 4), "host_threads": int(v[3]),
                     "hbm_gb": round(float(v[4]) / 1e9, 2), "pinned_host_gb": round(float(v[5]) / 1e9, 2)} for r, v in enumerate(every)]
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    ms_per_step = elapsed / args.steps * 1e3
    value = world * mbs_per_step * args.steps / elapsed

    # ---- bit-exactness against the REFERENCE decoder, on the output of a TIMED step: every distinct stream of the last step of the
    # timed region, first / middle / last frame (the last one depends on all the others through the references) ----
    verified = None
    kept = pipe.kept.pop(pipe.keep_group, None)
    if rank == 0 and not args.no_verify and kept:
        verified = verify_against_reference(env, {i: kept[i] for i in env["distinct"]}, paths, sorted({0, F // 2, F - 1}))
        if verified:
            verified["what"] = "rasters written by the last step of the timed region"
    del kept

    log("verified: %s" % (verified,))
    # ---- per-kernel timing of one more (un-pipelined) step: HIP events on the streams the kernels run on ----
    ctx.profile(True); ctx.kernel_stats(reset=True)
    g = pipe.decoded
    t0 = time.perf_counter()
    pipe._submit_keys(g); pipe._submit_inters(g); pipe.keys += 1; pipe.inter_h += 1; pipe.inters += 1
    ctx.sync()
    t_parse_alone = time.perf_counter() - t0
    pipe.decode(release=False)
    ctx.sync()
    kstats = ctx.kernel_stats(reset=True); ctx.profile(False)
    verify_decs, verify_base = pipe.groups[g], 0

    # macroblocks of the profiled step by kind (the records are in HBM: ask them)
    split_mbs = whole_mbs = intra_mbs = 0
    for i in env["distinct"]:                                               # one decoder per distinct stream, scaled up
        mult = seeds.count(seeds[i])
        for f in range(F):
            _, mb, _ = verify_decs[i].read_records(verify_base + f)
            inter = (mb["flags"] & 4) != 0
            sp = int((inter & (mb["y_mode"] == 9)).sum())
            split_mbs += mult * sp; whole_mbs += mult * (int(inter.sum()) - sp); intra_mbs += mult * int((~inter).sum())
    launches_per_step = {"recon_inter": max(1, kstats["recon_inter_launches"]), "recon_split": max(1, kstats["recon_split_launches"]),
                         "recon_intra": max(1, kstats["recon_intra_launches"]),
                         "loopfilter": max(1, kstats["loopfilter_launches"]), "parse_headers": max(1, kstats["parse_launches"])}
    units = {"recon_inter": whole_mbs, "recon_split": split_mbs, "recon_intra": intra_mbs,
             "loopfilter": S * F * mbs_per_frame, "parse_tokens": S * F * mbs_per_frame, "parse_headers": S * F * mbs_per_frame}
    traffic, traffic_source = pmc_traffic(args.config)
    traffic = traffic or {}

    def roof(k):
        ms = kstats[k + "_ms"]
        if ms <= 0 or units[k] == 0:
            return None
        n = launches_per_step[k]
        bytes_per_launch = BYTES_PER_MB[k] * units[k] / n
        achieved = bytes_per_launch / (ms / n * 1e-3) / 1e9
        tr = traffic.get(k)
        return {"bound": "hbm", "kernel": KERNEL_NAMES[k], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None if tr is None else round(tr * units[k] / n),
                "avg_launch_us": round(ms / n * 1e3, 3), "launches_per_step": n, "ms_per_step": round(ms, 3),
                "algorithmic_bytes_per_launch": round(bytes_per_launch), "traffic_source": TRAFFIC_SOURCES.get(k) if tr is not None else None,
                "measured": "one un-pipelined step after the timed region, HIP events on the kernel's stream; the step's frames are parsed before its reconstruction starts, so the worker waves beside it are idle or gone (they leave 100 ms after their last frame): the pipelined figures are in profiles/r05_kernel_trace.md"}
    roofs = {k: roof(k) for k in BYTES_PER_MB if k != "parse_tokens"}
    # k_token_workers is RESIDENT: its workgroups draw frames from a queue for as long as there is work, so there is no launch to
    # put events around.  Its roofline is priced on the time it was resident for the work it did: the whole timed region (the
    # driver's clock), during which it parsed `parsed_macroblocks` macroblocks.  One "launch" = one step's share of that.
    tok_mbs = tstats["parsed_macroblocks"]
    tok_bytes_per_step = BYTES_PER_MB["parse_tokens"] * tok_mbs / args.steps
    tok_achieved = BYTES_PER_MB["parse_tokens"] * tok_mbs / elapsed / 1e9
    tr = traffic.get("parse_tokens")
    roofs["parse_tokens"] = {"bound": "hbm", "kernel": KERNEL_NAMES["parse_tokens"], "achieved": round(tok_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(tok_achieved / HBM_PEAK_GBS, 5), "traffic": None if tr is None else round(tr * tok_mbs / args.steps),
                             "avg_launch_us": round(ms_per_step * 1e3, 3), "launches_per_step": 1, "ms_per_step": round(ms_per_step, 3),
                             "algorithmic_bytes_per_launch": round(tok_bytes_per_step), "macroblocks_parsed_in_the_timed_region": tok_mbs,
                             "duration_source": "resident kernel: the timed region's wall time / steps (its workgroups hold their CUs for the whole region); "
                                                "HBM is the nominal roof the contract prices against -- what binds this kernel is VALU issue of one wave per SIMD "
                                                "at live_lanes_of_64 lane use: entropy_decode_roof (lanes / step latency, in-kernel time accounting)"}
    # the dominant kernel by GPU time in a step: the resident workers hold every CU for the whole step
    roofline = dict(roofs["parse_tokens"])
    roofline["entropy_decode_alone_ms"] = round(t_parse_alone * 1e3, 1)      # (the un-pipelined latency of one step's chains: a latency, not a duration of the pipelined run)
    roofline["traffic_source"] = TRAFFIC_SOURCES.get("parse_tokens") or traffic_source
    roofline["path_frac_of_hbm_peak"] = round(value / world * PATH_BYTES_PER_MB / (HBM_PEAK_GBS * 1e9), 5)
    # what the whole path really moves per macroblock: every kernel's measured traffic (PMC passes: profiles/pmc_traffic.json) weighted by
    # the macroblocks it touches in a step -- the two parse kernels, the expansion pass packed storage adds, reconstruction, loop filter
    if traffic:
        packed = bool(info.get("packed_coefficients"))
        all_mbs = max(1, units["loopfilter"])
        # (round 6: no expansion pass any more -- the reconstruction kernels read the packed words themselves)
        per_kernel = {"parse_tokens": all_mbs, "parse_headers": all_mbs,
                      "recon_inter": units["recon_inter"] + units["recon_split"], "recon_intra": units["recon_intra"], "loopfilter": all_mbs}
        if all(k in traffic for k in per_kernel if per_kernel[k]):
            tb = sum(traffic[k] * n for k, n in per_kernel.items() if n) / all_mbs
            roofline["path_traffic_bytes_per_mb"] = round(tb, 1)
            roofline["path_traffic_over_algorithmic"] = round(tb / PATH_BYTES_PER_MB, 2)

    # ---- bit-exactness of the profile step too (all three formats of evidence agree: timed step, profile step, pytest) ----
    verified_profile_step = None
    if rank == 0 and not args.no_verify:
        verified_profile_step = verify_against_reference(env, {i: verify_decs[i] for i in env["distinct"][:24]}, paths, [F - 1])

    # ---- device half alone (round-1 metric): the step just parsed stays resident, reconstruction replayed ----
    device_half = None
    if not args.no_device_half:
        reps = max(2, min(args.steps, 5))

        def replay():
            for f in range(F):
                ctx.decode_batch(verify_decs, [verify_base + f] * S)
            for d in verify_decs:
                d.rewind_to(verify_base)
        for d in verify_decs:
            d.rewind_to(verify_base)
        replay(); ctx.sync()
        ctx.profile(True); ctx.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            replay()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        hs = ctx.kernel_stats(reset=True); ctx.profile(False)
        alone = {}
        for k in ("recon_inter", "recon_split", "recon_intra", "loopfilter"):
            n = hs[k + "_launches"]
            if n and units[k]:
                ms = hs[k + "_ms"] / n
                alone[k] = {"kernel": KERNEL_NAMES[k], "avg_launch_us": round(ms * 1e3, 1),
                            "frac": round(BYTES_PER_MB[k] * units[k] / launches_per_step[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        device_half = {"value": round(mbs_per_step / dt, 1), "unit": "macroblocks/s", "ms_per_step": round(dt * 1e3, 3), "kernels_alone": alone,
                       "note": "reconstruction + loop filter only, parsed records resident in HBM (round-1 metric); kernels_alone: the same kernels with idle worker grids "
                               "(lingering waves are asleep), frac = algorithmic bytes / launch time / HBM peak"}
    pipe_K, pipe_D = pipe.K, pipe.D
    mbs_whole_run = (pipe.frames_submitted + min(4, S) * F) * mbs_per_frame       # (+ the calibration frames)
    del pipe, verify_decs

    log("profile step and device half done")
    # ---- small batches: the reference's actual callers (one stream, one 8-chunk ExCamera bundle), end to end ----
    small = {}
    if rank == 0 and args.small_batches:
        for n in [int(x) for x in args.small_batches.split(",") if x]:
            if n >= S:
                continue
            # few streams are parsed by host workers (aa_submit_frames routes them): no chains of seconds to hide, so no deep
            # look-ahead either -- one group ahead keeps the GPU's reconstruction and the host's parse overlapped
            host_routed = n <= min(threads, 24)
            # (round 6: such calls are parsed frame-parallel by the context's host lanes and return at once -- a deeper look-ahead
            # keeps all the host's cores busy: AA_BENCH_SMALL_DEPTH groups ahead)
            small_depth = int(os.environ.get("AA_BENCH_SMALL_DEPTH", "3"))
            p = Pipeline(env, streams[:n], small_depth if host_routed else pipe_K, small_depth if host_routed else pipe_D, 0 if host_routed else args.header_ahead)
            p.run(3); ctx.sync()
            reps = max(6, p.K)
            t0 = time.perf_counter()
            p.run(reps); ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            # the same streams through the host parser path (aa_stream_decode: one host core per stream, no batching)
            d1 = aa.Decoder(ctx, width, height)
            t0 = time.perf_counter()
            for fr in streams[0]:
                d1.get_frame_output(fr)
            ctx.sync()
            dt_host = time.perf_counter() - t0
            small[str(n)] = {"mb_per_s": round(n * F * mbs_per_frame / dt, 1), "ms_per_step": round(dt * 1e3, 2),
                             "route": "host lanes, frame-parallel (aa_submit_frames, few streams: header pre-pass in the call, every frame body on a worker thread)" if host_routed else "GPU token lanes"}
            # the figure above is empty pipeline -> empty pipeline over `reps` steps (it pays for the first key-frame chain); between
            # fill and drain: the mean interval of the hand-overs after the first one of the timed run
            done = p.done_t[-reps:]
            if len(done) == reps and reps > 1 and done[-1] > done[0]:
                small[str(n)]["between_fill_and_drain_mb_per_s"] = round(n * F * mbs_per_frame * (reps - 1) / (done[-1] - done[0]), 1)
                small[str(n)]["steps_timed"] = reps
            small.setdefault("1_host_parser", {"mb_per_s": round(F * mbs_per_frame / dt_host, 1), "ms_per_frame": round(dt_host / F * 1e3, 2),
                                               "note": "aa_stream_decode: serial BoolDecoder on one host core, frame by frame (Decoder::get_frame_output)"})
            del p, d1

    log("small batches done: %s" % ({k: v.get("mb_per_s") for k, v in small.items()},))
    # ---- the other BASELINE configs, end to end, parity checked in the run ----
    secondary = {}
    if args.secondary:
        for cfg_name in [c for c in args.secondary.split(",") if c and c != args.config]:
            try:
                secondary[cfg_name] = run_secondary(args, ctx, cfg_name, rank, world, threads)
            except SystemExit:
                raise
            except Exception as e:                      # (a secondary figure that cannot be had must not take the headline down with it)
                secondary[cfg_name] = {"error": "%s: %s" % (type(e).__name__, e)}
            log("secondary %s: %s" % (cfg_name, {k: secondary[cfg_name].get(k) for k in ("value", "ms_per_step", "error")}))
            if cfg_name.endswith("_subpel") and "error" not in secondary[cfg_name]:
                try:
                    secondary[cfg_name]["with_a_lane_per_partition"] = lane_per_partition_leg(args, cfg_name, rank, world, threads)
                except SystemExit:
                    raise
                except Exception as e:
                    secondary[cfg_name]["with_a_lane_per_partition"] = {"error": "%s: %s" % (type(e).__name__, e)}
                log("secondary %s with a lane per partition: %s" % (cfg_name, {k: secondary[cfg_name]["with_a_lane_per_partition"].get(k) for k in ("value", "ms_per_step", "error")}))

    # ---- every frame on the GPU's token lanes (host_share_ms = 0): the same workload, key frames `--key-ahead` steps ahead ----
    lanes_only = None
    if args.lanes_only_steps > 0 and (env.get("keys_on_host") or env.get("urgent_keys_on_host")):
        share = ctx.info()["host_share_ms"]
        ctx.set_host_share_ms(0)
        env2 = dict(env); env2["keys_on_host"] = False; env2["urgent_keys_on_host"] = False; env2["deliver_ring"] = None
        p = Pipeline(env2, streams, K, D, args.header_ahead)
        ctx.sync(); ctx.kernel_stats(reset=True)
        t0 = time.perf_counter()
        p.run(args.lanes_only_steps); ctx.sync()
        dt = time.perf_counter() - t0
        st2 = ctx.kernel_stats(reset=True)
        lanes_only = {"value": round(world * mbs_per_step * args.lanes_only_steps / dt, 1), "unit": "macroblocks/s", "steps": args.lanes_only_steps, "ms_per_step": round(dt / args.lanes_only_steps * 1e3, 2),
                      "first_step_done_at_ms": round((p.done_t[0] - t0) * 1e3), "frames_parsed_on_host_cores": st2["host_routed_frames"],
                      "note": "empty pipeline to empty pipeline like `value` (fewer steps: the fill weighs more); this rank only"}
        if len(p.done_t) > 1:
            lanes_only["between_fill_and_drain_value"] = round(world * mbs_per_step * (len(p.done_t) - 1) / (p.done_t[-1] - p.done_t[0]), 1)
        ctx.set_host_share_ms(share)
        del p

    # ---- every frame DELIVERED (what vp8decode / xc-decode-bundle do with every shown frame): the same workload a few steps more,
    # each reconstructed frame gathered and copied to pinned host memory beside the next frame's reconstruction ----
    if delivery is None and args.deliver_steps > 0 and rank == 0:
        try:
            env3 = dict(env); env3["deliver_ring"] = [ctx.pinned_alloc(S * raster_bytes) for _ in range(DELIVER_RING)]
            p = Pipeline(env3, streams, K, D, args.header_ahead)
            ctx.sync()
            t0 = time.perf_counter()
            p.run(args.deliver_steps); ctx.download_wait(); ctx.sync()
            dt = time.perf_counter() - t0
            delivery = {"leg": "a separate run of %d steps after the timed region (the headline `value` does not deliver); empty pipeline to empty pipeline" % args.deliver_steps,
                        "value": round(mbs_per_step * args.deliver_steps / dt, 1), "unit": "macroblocks/s", "ms_per_step": round(dt / args.deliver_steps * 1e3, 2),
                        "bytes_per_step": p.delivered_bytes // args.deliver_steps, "gb_per_s": round(p.delivered_bytes / dt / 1e9, 2), "copies_per_frame_index": 1,
                        "destination_ring": DELIVER_RING, "first_step_done_at_ms": round((p.done_t[0] - t0) * 1e3),
                        "note": "what the bus gives a lone 1.5-GB device -> pinned-host copy on such a box: 57 GB/s (tools/pcie_probe.py, profiles/r06_pcie_probe.json)"}
            if len(p.done_t) > 1:
                delivery["between_fill_and_drain_gb_per_s"] = round(p.delivered_bytes * (len(p.done_t) - 1) / len(p.done_t) / (p.done_t[-1] - p.done_t[0]) / 1e9, 2)
            del p
            for a_ in env3["deliver_ring"]:
                ctx.pinned_free(a_)
        except Exception as e:                          # (must not take the headline down)
            delivery = {"error": "%s: %s" % (type(e).__name__, e)}
        log("delivery leg: %s" % ({k: delivery.get(k) for k in ("value", "gb_per_s", "between_fill_and_drain_gb_per_s", "error")},))

    # ---- host parser (the product's C++ BoolDecoder path used for single streams): rate per core ----
    pp = aa.Parser(width, height)
    t0 = time.perf_counter()
    for fr in streams[0]:
        pp.parse(fr)
    parser_only = len(streams[0]) * mbs_per_frame / (time.perf_counter() - t0)

    # ---- cpu_baseline leg (rank 0, 1 GPU only): the REFERENCE decoder (oracle/_ref, built without x86 asm), one thread
    # and 8 concurrent processes ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref_time = os.path.join(ROOT, "oracle", "_ref", "ref_time")
        if os.path.exists(ref_time):
            reps = max(1, int(round(12.0 / (F * mbs_per_frame / 90000.0))))
            out = subprocess.run([ref_time, paths[0], str(reps)], check=True, capture_output=True, text=True).stdout
            r = json.loads(out)
            t0 = time.perf_counter()
            procs = [subprocess.Popen([ref_time, paths[i % len(paths)], str(max(1, reps // 2))], stdout=subprocess.PIPE, text=True) for i in range(8)]
            outs = [json.loads(p.communicate()[0]) for p in procs]
            wall8 = time.perf_counter() - t0
            cpu_baseline = {"value": round(r["mb_per_s"], 1), "unit": "macroblocks/s", "cores": 1, "kind": "reference",
                            "sample": "stream seed %d (%d frames %dx%d) decoded %d times by oracle/_ref/ref_time; "
                                      "reference built without x86 asm (no assembler in the image)" % (seeds[0], F, width, height, reps),
                            "phases": {"parse_fraction": round(r["parse_s"] / r["seconds"], 3),
                                       "reconstruct_and_loopfilter_fraction": round(r["decode_s"] / r["seconds"], 3)},
                            "eight_processes": {"value": round(sum(o["macroblocks"] for o in outs) / wall8, 1), "cores": 8,
                                                "note": "8 concurrent ref_time processes on 8 streams, wall clock incl. process start"}}
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

    if rank == 0:
        cfg = workload.CONFIGS[args.config]
        shape = "all key frames" if args.config.endswith("_intra") else "1 key + %d inter" % (F - 1)
        line = {
            "metric": "1080p macroblocks/s decode (bit-exact vs reference)" if height == 1080 else "%dp macroblocks/s decode (bit-exact vs reference)" % height,
            "value": round(value, 1), "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d independent %dx%d streams per GPU x %d frames (%s, y_ac_qi %d, loop filter %d), END TO END: "
                                   "compressed frames in host memory -> entropy decode on the GPU -> reconstruction + loop filter -> rasters in HBM"
                                   % (args.config, S, width, height, F, shape, cfg[3], cfg[4]),
                       "streams_per_gpu": S, "frames_per_stream": F, "macroblocks_per_step_per_gpu": mbs_per_step,
                       "compressed_bytes_per_mb": round(compressed_bytes / mbs_per_step, 2), "sharding": "streams, one shard per GPU, no data-path collective",
                       "schedule": args.schedule, "key_frames_ahead": pipe_K, "inter_frames_ahead": pipe_D, "inter_headers_ahead_of_tokens": args.header_ahead, "host_threads": threads,
                       "hbm_budget_gb": round(hbm_budget / 1e9, 1), "look_ahead_if_memory_were_free_gb": planned_need_gb, "hbm_taken_by_the_context_gb": memory["hbm_taken_by_the_context_gb"],
                       "coefficient_storage": "packed" if info["packed_coefficients"] else "dense"},
            "memory": memory, "entropy_decode_roof": lanes_roof, "delivery": delivery, "macroblocks_parsed_whole_run": mbs_whole_run,
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            "kernels": roofs, "units_per_step": units, "launches_per_step": launches_per_step, "device_half": device_half,
            "steady_state": None if steady_ms is None else {"ms_per_step": round(steady_ms, 3), "value": round(world * mbs_per_step / (steady_ms * 1e-3), 1),
                                                            "note": "mean interval between step hand-overs after the first one (the series is timed_region.step_done_at_ms); `value` itself also pays for filling and draining the pipeline"},
            "stages": {"host_prepass_and_staging_s_per_step": round(host_submit_s, 4),
                       "entropy_decode_alone_s_per_step": round(t_parse_alone, 4),
                       "note": "entropy_decode_alone = one step's submit -> parse finished with nothing else on the GPU (a latency: the longest chain, a key frame)"},
            "timed_region": timed_region, "per_rank": per_rank, "small_batches": small, "secondary": secondary, "all_frames_on_gpu_lanes": lanes_only, "per_rank_on_2_cpus": two_cpus,
            "host_share": {"host_cpus_usable": host_cpus, "host_cpus_visible": os.cpu_count(), "host_lanes_of_this_rank": int(os.environ.get("ALFALFA_AMD_HOST_LANES") or 0), "urgent_groups_planned": env.get("urgent_groups"), "urgent_key_frames_on_host": bool(env.get("urgent_keys_on_host")), "a_group_of_key_frames_on_the_host_route_would_take_ms": env.get("urgent_host_estimate_ms"),
                           "groups_whose_key_frames_took_the_host_route_in_the_timed_region": urgent_groups_timed,
                           "host_share_ms": info["host_share_ms"], "key_frames_parsed_by_host_workers": bool(env.get("keys_on_host")), "host_threads": threads,
                           "host_rate_kb_per_ms_measured": info["host_rate_kb_per_ms"], "equivalent_cores_at_24_kb_per_ms": round(info["host_rate_kb_per_ms"] / 24.0, 1),
                           "host_batch_ms_per_step": round(tstats["host_batch_ms"] / args.steps, 1), "host_batch_parse_cpu_ms_per_step": round(tstats["host_batch_parse_cpu_ms"] / args.steps, 1),
                           "host_batch_arena_ms_per_step": round(tstats["host_batch_arena_ms"] / args.steps, 1), "pinned_allocations_in_timed_region": tstats["pinned_allocs"],
                           "note": "aa_submit_frames parses a hand-over's KEY frames on host cores when that fits host_share_ms on the rank's threads (one per stream and group "
                                   "of pictures: the long chains); inter frames -- 11 of 12 frames, ~80 % of the bools -- are always decoded by GPU lanes. "
                                   "all_frames_on_gpu_lanes is the same workload with host_share_ms = 0"},
            "host": {"parser_mb_per_s_per_core": round(parser_only, 1), "stream_generation_s": round(t_gen, 1)},
            "kernel_stats": kstats, "verified_bit_exact_vs_reference": verified, "verified_profile_step": verified_profile_step, "entry_state_handoff": handoff,
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
