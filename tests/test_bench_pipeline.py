"""bench.py's Pipeline -- the caller-side scheduler of the end-to-end run (what xc-decode-bundle is to the reference: decode-bundle.cc:56-99)
-- against a fake context, on the CPU: the order of hand-overs (the group about to be reconstructed first, key frames leading the inter
frames by K - D groups), admission by what pool and heap can still get, empty-to-empty runs, the kept group of the verified step, and
the urgent host route of an empty pipeline."""
import importlib.util
import os
import types

import pytest

from conftest import ROOT


@pytest.fixture()
def bench(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import alfalfa_amd as aa

    class FakeDecoder:
        count = 0

        def __init__(self, ctx, w, h):
            FakeDecoder.count += 1
            self.h = types.SimpleNamespace(value=FakeDecoder.count)
            self.released = 0

        def release_before(self, n):
            self.released = n
    monkeypatch.setattr(aa, "Decoder", FakeDecoder)
    return mod


class FakeCtx:
    """Books like aa_ctx_get_info's: every frame handed over takes `arena` bytes of pool and `coeff` bytes of heap until its group is decoded."""

    def __init__(self, limit, arena=10, coeff=10, heap_limit=None):
        self.limit, self.arena, self.coeff, self.heap_limit = limit, arena, coeff, heap_limit or limit
        self.log, self.in_flight = [], 0

    def submit_prepared(self, prepared, threads, defer, route="auto"):
        arr = prepared[0]
        self.log.append(("submit", len(arr), route))
        self.in_flight += len(arr)

    def decode_batch(self, decoders, idx):
        self.log.append(("decode", idx[0]))

    def info(self):
        return {"memory_limit_bytes": self.limit, "pool_bytes": self.in_flight * self.arena, "pool_free_bytes": 0, "pool_pending_bytes": 0,
                "heap_mapped_bytes": self.in_flight * self.coeff, "heap_limit_bytes": self.heap_limit, "heap_used_bytes": self.in_flight * self.coeff,
                "heap_free_chunks": 0, "lanes_starved": 0, "token_workgroups_alive": 0, "jobs_waiting": 0}

    def launch_tokens(self, n):
        return 0


def make(bench, ctx, n_streams=4, F=3, K=4, D=2, urgent=False):
    args = types.SimpleNamespace(trace_memory=False, overcommit=1.0, urgent_groups=2)
    env = {"ctx": ctx, "F": F, "args": args, "width": 64, "height": 64, "threads": 1, "distinct": [0, 2], "recon_reserve": 0,
           "key_coeff_bytes": ctx.coeff, "inter_coeff_bytes": ctx.coeff, "key_arena_bytes": ctx.arena, "inter_arena_bytes": ctx.arena, "key_dense_bytes": 0,
           "urgent_keys_on_host": urgent, "deliver_ring": None}
    streams = [[b"k%d" % i] + [b"i%d_%d" % (i, f) for f in range(1, F)] for i in range(n_streams)]
    p = bench.Pipeline(env, streams, K, D)
    real_decode = p.decode

    def decode_and_free(release=True):          # a decoded group's frames leave the fake context's books
        real_decode(release)
        ctx.in_flight -= n_streams * F
    p.decode = decode_and_free
    return p


def test_hand_overs_lead_reconstruction_and_runs_are_empty_to_empty(bench):
    ctx = FakeCtx(limit=10 ** 9)
    p = make(bench, ctx, K=4, D=2)
    p.run(6)
    assert p.decoded == 6 and p.keys == 6 and p.inter_h == 6 and not p.groups and ctx.in_flight == 0
    # the first hand-overs: keys of group 0, its inter frames, then key frames K - D = 2 groups ahead of the inter frames
    kinds = [(e[0], e[1]) for e in ctx.log[:8]]
    assert kinds[0] == ("submit", 4) and kinds[1] == ("submit", 8)           # keys g0 (4 streams), inter g0 (4 x 2 frames)
    # every group: its 3 frame indices decoded in order, after its frames were handed over
    decodes = [e[1] for e in ctx.log if e[0] == "decode"]
    assert decodes == [0, 1, 2] * 6
    first_decode = next(i for i, e in enumerate(ctx.log) if e[0] == "decode")
    handed = sum(e[1] for e in ctx.log[:first_decode] if e[0] == "submit")
    assert handed >= 4 * 3 and handed <= 4 * 3 * 4                          # at least group 0, at most K groups of keys + D of inter frames
    # a second run starts from empty again and ends empty
    p.run(3)
    assert p.decoded == 9 and ctx.in_flight == 0 and not p.groups


def test_admission_keeps_the_books_inside_the_limit(bench):
    F, n = 3, 4
    per_group = n * F * 20                                              # pool + heap bytes of one group on the fake books
    ctx = FakeCtx(limit=int(2.5 * per_group))
    p = make(bench, ctx, K=6, D=4)
    peak = 0
    real = ctx.submit_prepared

    def watch(*a, **k):
        nonlocal peak
        real(*a, **k)
        peak = max(peak, ctx.in_flight * 20)
    ctx.submit_prepared = watch
    p.run(8)
    assert p.decoded == 8 and p.refused > 0                              # hand-overs were put off ...
    assert peak <= ctx.limit + per_group                                 # ... and what was in flight stayed inside the limit (+ the group that must go)
    assert ctx.in_flight == 0


def test_the_last_timed_group_keeps_the_frames_of_its_distinct_streams(bench):
    ctx = FakeCtx(limit=10 ** 9)
    p = make(bench, ctx, K=2, D=2)
    p.keep_group = 2
    p.run(3)
    kept = p.kept[2]
    assert [d.released for d in kept] == [0, 3, 0, 3]                    # streams 0 and 2 (env["distinct"]) released nothing
    assert 0 not in p.kept and 1 not in p.kept


def test_an_empty_pipeline_takes_the_host_route_for_the_group_it_starts_with(bench):
    ctx = FakeCtx(limit=10 ** 9)
    p = make(bench, ctx, K=6, D=2, urgent=True)
    p.run(5)
    submits = [e for e in ctx.log if e[0] == "submit"]
    # the key frames of the first two groups to the host lanes (the call does not wait for them), then the first group's inter frames
    assert [s[2] for s in submits[:3]] == ["host", "host", "auto"] and submits[2][1] == 8
    assert p.urgent_groups == 2 and p.decoded == 5 and ctx.in_flight == 0
    assert sum(1 for s in submits if s[2] == "host") == 2


def test_calibration_runs_against_a_fake_context(bench, monkeypatch):
    """calibrate(): the lone key frame, the frames' sizes, the host-share probe (skipped where the host cannot take half of a hand-over) and the
    urgent-route decision -- every statement executed once on the CPU, so that a typo cannot wait for the GPU box to be found."""
    import alfalfa_amd as aa

    class Ctx(FakeCtx):
        def __init__(self):
            super().__init__(limit=10 ** 12)
            self.calls = 0

        def submit_frames(self, pairs, threads=0, defer_tokens=False, route="auto"):
            self.calls += 1
            return list(range(len(pairs)))

        def kernel_stats(self, reset=False):
            return {"token_steps": 1000, "packed_words": 4000, "packed_blocks": 400, "host_batch_ms": 0.0}

        def sync(self):
            pass

        def info(self):
            d = super().info()
            d.update(host_share_ms=80, packed_coefficients=1, host_rate_kb_per_ms=600)
            return d

    def frame_header(self, i):
        return {"num_coeff_blocks": 500}
    monkeypatch.setattr(aa.Decoder, "frame_header", frame_header, raising=False)
    args = types.SimpleNamespace(no_urgent_host=False, urgent_host=False, urgent_groups=2)
    for S, key_size in ((40, 2000), (40, 2_000_000), (8, 2000)):
        ctx = Ctx()
        streams = [[b"k" * key_size] + [b"i" * 100] * 2 for _ in range(S)]
        env = {"ctx": ctx, "F": 3, "width": 64, "height": 64, "threads": 4, "mbs_per_frame": 16, "compressed_bytes": sum(len(f) for st in streams for f in st),
               "raster_bytes": 64 * 64 * 3 // 2, "args": args}
        bench.calibrate(env, streams)
        assert env["key_coeff_bytes"] > 0 and env["inter_coeff_bytes"] > 0 and env["recon_reserve"] > 0 and env["step_latency_us"] > 0
        assert env["packed_storage"]["dense_bytes_per_block"] == 32
        cpus = aa.capi.lib().aa_host_cpus()
        probed = S > min(env["threads"], 24) and 80 * 24.0e3 * cpus >= 0.5 * S * key_size      # (a call with few streams takes the per-stream host route anyway)
        assert ctx.calls == 2 + (2 if probed else 0)
        assert env["keys_on_host"] == (probed and 0.9 * S * key_size <= 80 * 600e3)
        assert isinstance(env["urgent_keys_on_host"], bool) and (S > 24 or not env["urgent_keys_on_host"])
        assert 0 <= env["urgent_groups"] <= 2 and (env["urgent_groups"] > 0 or not env["urgent_keys_on_host"])
    # a rank that gets ONE of the host's cores (eight ranks on a small CPU grant): 40 key frames of 2 MB would take its lane 3.3 s --
    # longer than the GPU lanes take for theirs -- so no group's key frames are planned for the host route
    monkeypatch.setenv("ALFALFA_AMD_HOST_LANES", "1")
    ctx = Ctx()
    streams = [[b"k" * 2_000_000] + [b"i" * 100] * 2 for _ in range(40)]
    env = {"ctx": ctx, "F": 3, "width": 64, "height": 64, "threads": 4, "mbs_per_frame": 16, "compressed_bytes": sum(len(f) for st in streams for f in st),
           "raster_bytes": 64 * 64 * 3 // 2, "args": args}
    bench.calibrate(env, streams)
    assert ctx.calls == 2 and env["urgent_groups"] == 0 and env["urgent_keys_on_host"] is False and env["urgent_host_estimate_ms"] > 2800
