"""Device-side entropy decode (aa_submit_frames: macroblock headers and tokens parsed by GPU lanes) on a real MI355X:
  * every macroblock record and every coefficient block the GPU parser leaves in HBM equals, byte for byte, what the host
    parser (aa_parser_parse) produces -- goldens, 24 synthetic feature seeds, truncated frames, extreme geometries;
  * frames decoded through that path equal the oracle / the committed reference hashes;
  * many streams x many frames in one call, interleaved with host-parsed frames, with frames released on the way."""
import numpy as np
import pytest

import alfalfa_amd as aa
import vp8_oracle as vo
from conftest import GOLDEN, golden_frames, sha256

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def device_route(monkeypatch):
    """These tests are about the GPU parser: small calls would otherwise be routed to host workers (aa_submit_frames, "route")."""
    monkeypatch.setenv("ALFALFA_AMD_ROUTE", "device")


def assert_records_equal(got, want, what):
    gh, gmb, gcf = got
    wh, wmb, wcf = want
    assert gh == wh, (what, {k: (gh[k], wh[k]) for k in wh if gh[k] != wh[k]})
    a, b = gmb.reshape(-1).view(np.uint8).reshape(-1, 80), wmb.reshape(-1).view(np.uint8).reshape(-1, 80)
    if not (a == b).all():
        bad = np.nonzero((a != b).any(axis=1))[0]
        m = int(bad[0])
        raise AssertionError("%s: %d macroblock records differ, first mb %d: gpu %r host %r" % (what, len(bad), m, gmb.reshape(-1)[m], wmb.reshape(-1)[m]))
    assert gcf.shape == wcf.shape and (gcf == wcf).all(), "%s: coefficient blocks differ" % what


def check_stream(ctx, w, h, frames, want_hashes=None, per_call=None):
    """Submit the stream to the GPU parser (all frames in one call, or per_call at a time), compare records with the host
    parser's and rasters with the oracle's."""
    dec, host, ora = aa.Decoder(ctx, w, h), aa.Parser(w, h), vo.OracleDecoder(w, h)
    per_call = per_call or len(frames)
    for base in range(0, len(frames), per_call):
        part = frames[base:base + per_call]
        idx = ctx.submit_frames([(dec, fr) for fr in part])
        assert idx == list(range(base, base + len(part)))
        for k, fr in enumerate(part):
            fi = base + k
            assert_records_equal(dec.read_records(fi), host.parse(fr), "frame %d" % fi)
            ctx.decode_batch([dec], [fi])
            ora.decode(fr)
            got = dec.raster_bytes(fi)
            assert got == ora.raster_bytes(), "frame %d raster" % fi
            if want_hashes:
                assert sha256(got) == want_hashes[fi]


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_gpu_parser_matches_host_parser_and_reference(gpu_ctx, name):
    w, h, frames = golden_frames(name)
    check_stream(gpu_ctx, w, h, frames, GOLDEN[name]["raster_sha256"])


@pytest.mark.parametrize("seed", list(range(200, 224)))
def test_gpu_parser_on_synthetic_feature_streams(gpu_ctx, seed):
    import vp8_synth
    sizes = [(96, 80), (33, 17), (64, 64), (175, 143), (16, 16), (200, 48), (320, 176), (48, 256)]
    w, h = sizes[seed % len(sizes)]
    check_stream(gpu_ctx, w, h, vp8_synth.feature_stream(w, h, seed, 8).frames, per_call=3 if seed % 2 else None)


@pytest.mark.parametrize("name", ["qcif_q30_lf24", "w200_q40_lf63s7", "qcif_allkey_q20"])
def test_gpu_parser_on_truncated_frames(gpu_ctx, name):
    from test_parser_vs_oracle import truncated
    w, h, frames = golden_frames(name)
    check_stream(gpu_ctx, w, h, truncated(frames))


def test_gpu_parser_on_extreme_geometries(gpu_ctx):
    import vp8_synth
    for w, h, seed in ((16, 4096, 901), (4096, 16, 902), (24, 1000, 903), (2000, 32, 904)):
        check_stream(gpu_ctx, w, h, vp8_synth.feature_stream(w, h, seed, 3).frames)


def test_many_streams_many_frames_in_one_call(gpu_ctx):
    """The bench's shape in small: every frame of every stream handed over in ONE call (frames of a stream in order, streams
    interleaved), then decoded in lock step; old frames released on the way."""
    names = ["qcif_q30_lf24", "qcif_q30", "qcif_allkey_q20", "cif_q60_lf40s5", "w200_q40_lf63s7", "synth_96x80_s1", "synth_175x143_s3"] * 3
    streams = [golden_frames(n) for n in names]
    decs = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    nf = min(len(f) for _, _, f in streams)
    pairs = [(decs[i], streams[i][2][f]) for f in range(nf) for i in range(len(decs))]
    idx = gpu_ctx.submit_frames(pairs, threads=4)
    assert idx == [f for f in range(nf) for _ in decs]
    for f in range(nf):
        gpu_ctx.decode_batch(decs, [f] * len(decs))
        if f >= 2:
            for d in decs:
                d.release_before(f - 1)
    for d, n in zip(decs, names):
        for f in (nf - 2, nf - 1):
            assert sha256(d.raster_bytes(f)) == GOLDEN[n]["raster_sha256"][f], (n, f)


def test_host_and_device_parsed_frames_interleave(gpu_ctx):
    """A stream may switch between aa_stream_parse (host) and aa_submit_frames (device) at any frame, segmentation map and
    all: the persistent state travels with it."""
    import vp8_synth
    for seed, (w, h) in ((203, (175, 143)), (211, (175, 143)), (208, (96, 80)), (216, (96, 80))):
        frames = vp8_synth.feature_stream(w, h, seed, 10).frames
        dec, ora = aa.Decoder(gpu_ctx, w, h), vo.OracleDecoder(w, h)
        for i, fr in enumerate(frames):
            if (i // 2) % 2 == 0:
                fi = gpu_ctx.submit_frames([(dec, fr)])[0]
                gpu_ctx.decode_batch([dec], [fi])
            else:
                _, fi = dec.get_frame_output(fr)
            ora.decode(fr)
            assert dec.raster_bytes(fi) == ora.raster_bytes(), (seed, i)
        # and the DecoderState that comes out is the one a pure host parse reaches
        p = aa.Parser(w, h)
        for fr in frames:
            p.parse(fr)
        assert dec.export_state() == p.export_state()


def test_bad_frame_in_a_batch(gpu_ctx):
    """A bitstream error stops ITS stream at that frame (same error class as the host parser) and nothing else."""
    w, h, frames = golden_frames("qcif_q30")
    a, b = aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h)
    bad = bytearray(frames[1]); bad[0] |= (2 << 1)          # VP8 version 2: Unsupported
    with pytest.raises(aa.AlfalfaError) as e:
        gpu_ctx.submit_frames([(a, frames[0]), (b, frames[0]), (a, bytes(bad)), (b, frames[1]), (a, frames[2]), (b, frames[2])])
    assert e.value.kind == "Unsupported"
    assert a.frame_count() == 1 and b.frame_count() == 3
    for f in range(3):
        gpu_ctx.decode_batch([b], [f])
        assert sha256(b.raster_bytes(f)) == GOLDEN["qcif_q30"]["raster_sha256"][f]
    # stream a goes on from where it stopped
    idx = gpu_ctx.submit_frames([(a, frames[1]), (a, frames[2])])
    assert idx == [1, 2]
    for f in range(3):
        gpu_ctx.decode_batch([a], [f])
    assert sha256(a.raster_bytes(2)) == GOLDEN["qcif_q30"]["raster_sha256"][2]


def test_long_stream_holds_memory_flat(gpu_ctx):
    """2000 frames through one decoder with aa_stream_release_before trailing by a few frames: HBM use stays flat
    (records and rasters are recycled; RasterHandle semantics of raster_handle.cc:113-122)."""
    w, h, frames = golden_frames("cif_q60_lf40s5")
    dec = aa.Decoder(gpu_ctx, w, h)
    want = GOLDEN["cif_q60_lf40s5"]["raster_sha256"]
    used = []
    n = 0
    for rep in range(2000 // len(frames)):
        use_gpu_parser = rep % 2 == 0
        if use_gpu_parser:
            idx = gpu_ctx.submit_frames([(dec, fr) for fr in frames])
            for fi in idx:
                gpu_ctx.decode_batch([dec], [fi])
        else:
            for fr in frames:
                _, fi = dec.get_frame_output(fr)
        n += len(frames)
        assert sha256(dec.raster_bytes(n - 1)) == want[len(frames) - 1]
        dec.release_before(n - 1)
        if rep % 25 == 24:
            used.append(gpu_ctx.memory()[0])
    assert max(used[1:]) - min(used[1:]) <= 64 << 20, used


def test_async_download_into_pinned_memory(gpu_ctx):
    """aa_stream_download_async: shown frames leave on the copy stream while the next frames are being decoded."""
    import ctypes as C
    from alfalfa_amd import capi
    L = capi.lib()
    name = "cif_q60_lf40s5"
    w, h, frames = golden_frames(name)
    dec = aa.Decoder(gpu_ctx, w, h)
    ysz, usz, vsz = dec.plane_sizes()
    bufs = []
    for fr in frames:
        p = C.c_void_p()
        capi.check(L.aa_pinned_alloc(gpu_ctx.h, ysz + usz + vsz, C.byref(p)))
        bufs.append(p)
        _, fi = dec.get_frame_output(fr)
        capi.check(L.aa_stream_download_async(dec.h, fi, C.c_void_p(p.value), C.c_void_p(p.value + ysz), C.c_void_p(p.value + ysz + usz)))
    capi.check(L.aa_stream_download_wait(dec.h))
    for i, p in enumerate(bufs):
        assert sha256(C.string_at(p.value, ysz + usz + vsz)) == GOLDEN[name]["raster_sha256"][i], i
        L.aa_pinned_free(p)


def test_two_phase_submit(gpu_ctx):
    """AA_SUBMIT_DEFER_TOKENS: macroblock headers at submit, tokens (and the coefficient blocks) at aa_launch_tokens -- or at
    the first call that needs the records.  Deferred and one-shot batches interleave; frames may be released in between."""
    import vp8_synth
    w, h, frames = golden_frames("cif_q60_lf40s5")
    want = GOLDEN["cif_q60_lf40s5"]["raster_sha256"]
    syn = vp8_synth.feature_stream(175, 143, 219, 6).frames
    a, b, c = aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, 175, 143)
    host, hsyn = aa.Parser(w, h), aa.Parser(175, 143)
    free0 = gpu_ctx.memory()[0]
    ia = gpu_ctx.submit_frames([(a, fr) for fr in frames[:4]], defer_tokens=True)          # batch 1 (deferred)
    ic = gpu_ctx.submit_frames([(c, fr) for fr in syn], defer_tokens=True)                 # batch 2 (deferred)
    ib = gpu_ctx.submit_frames([(b, fr) for fr in frames[:4]])                             # batch 3 (one shot)
    assert ia == [0, 1, 2, 3] and ib == [0, 1, 2, 3]
    assert gpu_ctx.launch_tokens(1) == 1                                                   # batch 1
    for f in range(4):
        want_rec = host.parse(frames[f])
        assert_records_equal(a.read_records(f), want_rec, "deferred frame %d" % f)
        assert_records_equal(b.read_records(f), want_rec, "one-shot frame %d" % f)
        gpu_ctx.decode_batch([a, b], [f, f])
        assert sha256(a.raster_bytes(f)) == want[f] and sha256(b.raster_bytes(f)) == want[f]
    # batch 2 was never launched explicitly: the first use of its records does it
    ora = vo.OracleDecoder(175, 143)
    for f, fr in enumerate(syn):
        if f == 0:
            assert_records_equal(c.read_records(0), hsyn.parse(fr), "implicit launch")
        else:
            hsyn.parse(fr)
        gpu_ctx.decode_batch([c], [f])
        ora.decode(fr)
        assert c.raster_bytes(f) == ora.raster_bytes(), f
    assert gpu_ctx.launch_tokens(0) == 0
    # a deferred batch whose frames go away before the second phase
    d = aa.Decoder(gpu_ctx, w, h)
    gpu_ctx.submit_frames([(d, fr) for fr in frames[:3]], defer_tokens=True)
    del d
    assert gpu_ctx.launch_tokens(0) == 0
    # ... and the stream carries on after a deferred batch like after any other
    assert len(frames) > 4
    ia2 = gpu_ctx.submit_frames([(a, fr) for fr in frames[4:8]], defer_tokens=True)
    assert ia2 == list(range(4, min(8, len(frames))))
    for f in ia2:
        gpu_ctx.decode_batch([a], [f])
        assert sha256(a.raster_bytes(f)) == want[f]
    assert free0 > 0


def test_gpu_parser_on_long_runs_of_skipped_macroblocks(gpu_ctx):
    """Skipped macroblocks take no decode steps, so long runs of them outrun the token lane's 64-entry flag ring and the lane
    waits for the next top-up (tok::macroblock_boundary) -- frames that are (almost) entirely skipped, single and multi-partition."""
    import vp8_synth
    for w, h, seed, density in ((1920, 48, 31, 0.0), (1920, 48, 32, 0.004), (640, 360, 33, 0.002), (4096, 16, 34, 0.0)):
        s = vp8_synth.SynthStream(w, h, seed)
        s.frame(key=True, q_index=30, skip_prob=3, density=density, skip_rate=1.0, intra_bpred=0.2)
        for k in range(3):
            s.frame(key=False, q_index=30, skip_prob=2 + k, density=density, skip_rate=1.0, log2_parts=k % 3, lf_level=8)
        check_stream(gpu_ctx, w, h, s.frames)


def test_context_info_and_demand_sized_coefficient_storage(gpu_ctx):
    """aa_ctx_get_info: the token workers' shape and what the context holds.  A frame's coefficients take what its non-zero
    blocks need (64-KB chunks of one heap), not 25 blocks per macroblock."""
    w, h, frames = golden_frames("cif_q60_lf40s5")
    dec = aa.Decoder(gpu_ctx, w, h)
    gpu_ctx.submit_frames([(dec, fr) for fr in frames[:3]])
    hdrs = [dec.frame_header(i) for i in range(3)]
    info = gpu_ctx.info()
    assert info["token_lanes_per_workgroup"] >= 1 and info["token_workgroups_capacity"] >= info["compute_units"] >= 1
    assert 952 <= info["token_lane_lds_bytes"] <= 4096 and info["heap_mapped_bytes"] > 0 and info["heap_limit_bytes"] >= info["heap_mapped_bytes"]
    nmb = hdrs[0]["num_macroblocks"]
    used_chunks = info["heap_used_bytes"] // 65536
    assert used_chunks >= 3                                                     # (other tests' frames may be alive too)
    if not info["packed_coefficients"]:
        assert sum(-(-hd["num_coeff_blocks"] // 2048) for hd in hdrs) <= used_chunks
    else:                                                                        # a mask word + at least one value per stored block
        assert 2 * sum(hd["num_coeff_blocks"] for hd in hdrs) <= used_chunks * 32768
    assert max(hd["num_coeff_blocks"] for hd in hdrs) < 25 * nmb
    for i in range(3):
        gpu_ctx.decode_batch([dec], [i])
    assert sha256(dec.raster_bytes(2)) == GOLDEN["cif_q60_lf40s5"]["raster_sha256"][2]


@pytest.mark.parametrize("packed", [True, False], ids=["packed", "dense"])
@pytest.mark.parametrize("vmm", [True, False])
def test_coefficient_heap_runs_out_and_frames_are_run_again(vmm, packed, monkeypatch):
    """A heap far too small for what is submitted (8 MB = 128 chunks, 60 CIF key frames want more): lanes wait, hand their frames
    back (TOK_NO_MEMORY), the runtime runs them again as memory comes back -- and says AA_ERR_NO_MEMORY (repeatable) when the
    caller has to release frames first.  Every raster still equals the reference's.  Both heap kinds: mapped on demand
    (hipMemMap) and one fixed allocation."""
    limit_mb = 4 if packed else 8                       # (packed frames are a third the size: a smaller heap runs dry the same way)
    monkeypatch.setenv("ALFALFA_AMD_HEAP_GROW_MB", "2")
    monkeypatch.setenv("ALFALFA_AMD_HEAP_LIMIT_MB", str(limit_mb))
    if not vmm:
        monkeypatch.setenv("ALFALFA_AMD_NO_VMM", "1")
    ctx = aa.Context(0)
    ctx.set_packed_coefficients(packed)
    name = "cif_q60_lf40s5"
    w, h, frames = golden_frames(name)
    decs = [aa.Decoder(ctx, w, h) for _ in range(60)]
    ctx.submit_frames([(d, frames[0]) for d in decs] + [(d, frames[1]) for d in decs])
    assert bool(ctx.info()["heap_is_virtual"]) == vmm
    import time
    time.sleep(2.5)         # (lanes that found the pool empty give up after 2 s: let them, whatever the order in which the frames are asked for below)
    # decode whatever can be decoded, release it at once (its chunks go back), come back to the frames that were refused
    nxt, refused = [0] * len(decs), 0
    for rnd in range(200):
        todo = [k for k in range(len(decs)) if nxt[k] < 2]
        if not todo:
            break
        progress = 0
        for k in todo:
            d, f = decs[k], nxt[k]
            try:
                ctx.decode_batch([d], [f])
            except aa.AlfalfaError as e:
                assert e.kind == "NoMemory", e
                refused += 1
                continue
            assert sha256(d.raster_bytes(f)) == GOLDEN[name]["raster_sha256"][f], (f, k)
            d.release_before(f + 1)
            nxt[k] += 1; progress += 1
        ctx.sync()
        assert progress, "a whole round of decode calls was refused although decoded frames had been released"
    assert all(n == 2 for n in nxt)
    st = ctx.kernel_stats()
    assert st["nomem_retries"] > 0 or refused > 0, st
    assert ctx.info()["heap_mapped_bytes"] <= limit_mb << 20


def test_small_calls_are_routed_to_host_workers_and_give_the_same_records(gpu_ctx, monkeypatch):
    """aa_submit_frames with few streams: the frames are parsed by the host's cores instead of a GPU lane each -- same records,
    same rasters, counted in host_routed_frames; AA_SUBMIT_DEVICE forces the GPU parser.  Two host routes (round 6): a call that
    brings several frames per stream is parsed FRAME-PARALLEL by the context's host lanes (header pre-pass in the call, every
    frame body an independent chain on a worker thread, the call does not wait); ALFALFA_AMD_FEW_ROUTE=streams keeps round 3's
    route (one worker per stream, Parser::parse, inside the call), which is also what one-frame-per-stream calls and streams
    that use segmentation take."""
    monkeypatch.delenv("ALFALFA_AMD_ROUTE", raising=False)
    monkeypatch.delenv("ALFALFA_AMD_FEW_ROUTE", raising=False)
    names = ["qcif_q30_lf24", "cif_q60_lf40s5", "synth_175x143_s3", "w200_q40_lf63s7", "qcif_allkey_q20"]
    streams = [golden_frames(n) for n in names]
    nf = min(len(f) for _, _, f in streams)
    before = gpu_ctx.kernel_stats()["host_routed_frames"]
    auto = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    per_stream = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    one_by_one = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    dev = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    gpu_ctx.submit_frames([(auto[i], streams[i][2][f]) for i in range(len(names)) for f in range(nf)], threads=8)
    routed = gpu_ctx.kernel_stats()["host_routed_frames"] - before
    # (a frame that switches segmentation on leaves the host lanes for a GPU lane, and so do the frames behind it: only the
    # synthetic stream can)
    assert (len(names) - 1) * nf <= routed <= len(names) * nf, routed
    before = gpu_ctx.kernel_stats()["host_routed_frames"]
    monkeypatch.setenv("ALFALFA_AMD_FEW_ROUTE", "streams")
    gpu_ctx.submit_frames([(per_stream[i], streams[i][2][f]) for i in range(len(names)) for f in range(nf)], threads=8)
    monkeypatch.delenv("ALFALFA_AMD_FEW_ROUTE")
    assert gpu_ctx.kernel_stats()["host_routed_frames"] == before + len(names) * nf
    for f in range(nf):                                   # one frame per stream and call: the per-stream route again
        gpu_ctx.submit_frames([(one_by_one[i], streams[i][2][f]) for i in range(len(names))], threads=8)
    assert gpu_ctx.kernel_stats()["host_routed_frames"] == before + 2 * len(names) * nf
    before = gpu_ctx.kernel_stats()["host_routed_frames"]
    gpu_ctx.submit_frames([(dev[i], streams[i][2][f]) for i in range(len(names)) for f in range(nf)], threads=8, route="device")
    assert gpu_ctx.kernel_stats()["host_routed_frames"] == before
    for i, n in enumerate(names):
        for f in range(nf):
            want = dev[i].read_records(f)
            assert_records_equal(auto[i].read_records(f), want, "%s frame %d (host lanes)" % (n, f))
            assert_records_equal(per_stream[i].read_records(f), want, "%s frame %d (a worker per stream)" % (n, f))
            assert_records_equal(one_by_one[i].read_records(f), want, "%s frame %d (frame by frame)" % (n, f))
    for f in range(nf):
        for ds in (auto, per_stream, one_by_one, dev):
            gpu_ctx.decode_batch(ds, [f] * len(names))
    for i, n in enumerate(names):
        assert (sha256(auto[i].raster_bytes(nf - 1)) == sha256(per_stream[i].raster_bytes(nf - 1)) == sha256(one_by_one[i].raster_bytes(nf - 1))
                == sha256(dev[i].raster_bytes(nf - 1)) == GOLDEN[n]["raster_sha256"][nf - 1]), n
    # the frame-parallel route with a look-ahead that is released early: frames still on a worker are waited for, not freed under it
    w, h, frames = golden_frames("cif_q60_lf40s5")
    d = aa.Decoder(gpu_ctx, w, h)
    gpu_ctx.submit_frames([(d, fr) for fr in frames], threads=8)
    gpu_ctx.decode_batch([d], [0])
    d.release_before(len(frames))
    del d
    # a bitstream error stops ITS stream only, on the host route as on the device route
    w, h, frames = golden_frames("qcif_q30")
    a, b = aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h)
    with pytest.raises(aa.AlfalfaError):
        gpu_ctx.submit_frames([(a, frames[0]), (a, b"\x00\x00"), (a, frames[1]), (b, frames[0]), (b, frames[1])])
    assert a.frame_count() == 1 and b.frame_count() == 2


@pytest.mark.parametrize("name", ["qcif_q30_lf24", "cif_q60_lf40s5"])
def test_error_concealment_on_the_gpu_parser_and_the_host_parser(gpu_ctx, name):
    """Decoder::set_error_concealment (decoder.hh:298): frames cut inside the first partition, at its end, or before their tag
    is complete are accepted -- by the GPU parser (the host pre-pass reads the tag, the lanes read zeros past the end) and by the
    host parser -- and decode to what the oracle (pinned to the reference built with the flag on, test_parser_vs_oracle.py)
    produces; without the flag they are refused as before."""
    from test_parser_vs_oracle import cut_for_concealment
    w, h, frames = golden_frames(name)
    cut = cut_for_concealment(frames)
    plain = aa.Decoder(gpu_ctx, w, h)
    with pytest.raises(aa.AlfalfaError) as e:
        gpu_ctx.submit_frames([(plain, cut[0]), (plain, cut[1])])
    assert e.value.kind == "Invalid" and plain.frame_count() == 1
    dev, host, ora = aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h), vo.OracleDecoder(w, h)
    for d in (dev, host, ora):
        d.set_error_concealment(True)
    assert dev.error_concealment()
    idx = gpu_ctx.submit_frames([(dev, fr) for fr in cut])
    assert idx == list(range(len(cut)))
    for i, fr in enumerate(cut):
        gpu_ctx.decode_batch([dev], [i])
        shown, fi = host.get_frame_output(fr)
        assert ora.decode(fr) == shown
        want = ora.raster_bytes()
        assert dev.raster_bytes(i) == want, "GPU parser, frame %d" % i
        assert host.raster_bytes(fi) == want, "host parser, frame %d" % i


def test_frames_given_as_records_decode_like_their_bitstream(gpu_ctx):
    """aa_stream_append_records: the records one decoder parsed -- on the GPU or on the host -- handed to ANOTHER decoder as
    records (no bitstream) reconstruct to the same rasters: Encoder::write_frame's reference update without the serialise ->
    parse round trip (encoder.cc:146-160)."""
    for name, route in (("cif_q60_lf40s5", "device"), ("synth_175x143_s3", "host"), ("qcif_q30_lf24", "device")):
        w, h, frames = golden_frames(name)
        src, dst = aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h)
        gpu_ctx.submit_frames([(src, fr) for fr in frames], route=route)
        for i in range(len(frames)):
            hdr, mb, cf = src.read_records(i)
            fi = dst.append_records(hdr, mb, cf)
            assert fi == i
            gpu_ctx.decode_batch([src], [i]); gpu_ctx.decode_batch([dst], [fi])
            assert sha256(dst.raster_bytes(fi)) == sha256(src.raster_bytes(i)) == GOLDEN[name]["raster_sha256"][i], (name, i)
    w, h, frames = golden_frames("qcif_q30")
    d = aa.Decoder(gpu_ctx, w, h)
    hdr, mb, cf = aa.Parser(w, h).parse(frames[0])
    bad = mb.copy(); bad.reshape(-1)[3]["coeff_index"] = 10 ** 6
    with pytest.raises(aa.AlfalfaError) as e:
        d.append_records(hdr, bad, cf)
    assert e.value.kind == "BadArgument"


def test_key_frames_of_big_calls_are_parsed_by_host_workers(gpu_ctx, monkeypatch):
    """aa_submit_frames with many streams: the call's KEY frames (the long chains) go to the context's host lanes -- host cores in the
    role of token lanes, the call does not wait for them --, the inter frames to the GPU's lanes (aa_ctx_set_host_share_ms); records
    and rasters are those of the all-device route; a stream whose frames in the call are not all key frames stays on the GPU's lanes,
    and so does one that uses segmentation (its persistent map lives on the device); 0 switches the host share off."""
    monkeypatch.delenv("ALFALFA_AMD_ROUTE", raising=False)
    names = ["qcif_q30_lf24", "cif_q60_lf40s5", "synth_175x143_s3", "w200_q40_lf63s7", "qcif_q30", "qvga_q100"]
    streams = [golden_frames(names[i % len(names)]) for i in range(30)]
    nf = 3
    assert gpu_ctx.info()["host_share_ms"] > 0
    hyb = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    dev = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    before = gpu_ctx.kernel_stats()["host_routed_frames"]
    # the last stream hands its key frame over together with an inter frame: not a key-frame-only stream -> lanes
    idx = gpu_ctx.submit_frames([(d, st[2][0]) for d, st in zip(hyb, streams)] + [(hyb[-1], streams[-1][2][1])], threads=8)
    assert idx == [0] * 30 + [1]
    on_host = sum(1 for i in range(29) if names[i % len(names)] != "synth_175x143_s3")      # (that golden's key frame switches segmentation on)
    assert gpu_ctx.kernel_stats()["host_routed_frames"] == before + on_host == before + 24
    gpu_ctx.submit_frames([(d, st[2][f]) for d, st in zip(hyb[:-1], streams) for f in (1, 2)] + [(hyb[-1], streams[-1][2][2])], threads=8)
    assert gpu_ctx.kernel_stats()["host_routed_frames"] == before + on_host     # (inter frames: lanes)
    gpu_ctx.submit_frames([(d, st[2][f]) for d, st in zip(dev, streams) for f in range(nf)], threads=8, route="device")
    assert gpu_ctx.kernel_stats()["host_routed_frames"] == before + on_host
    for i in (0, 1, 2, 3, 4, 5, 29):
        for f in range(nf):
            assert_records_equal(hyb[i].read_records(f), dev[i].read_records(f), "stream %d frame %d" % (i, f))
    for f in range(nf):
        gpu_ctx.decode_batch(hyb, [f] * len(hyb)); gpu_ctx.decode_batch(dev, [f] * len(dev))
    for i in range(30):
        want = GOLDEN[names[i % len(names)]]["raster_sha256"][nf - 1]
        assert sha256(hyb[i].raster_bytes(nf - 1)) == sha256(dev[i].raster_bytes(nf - 1)) == want, i
    # a bitstream error in a host-parsed key frame stops ITS stream only
    w, h, frames = golden_frames("qcif_q30")
    ds = [aa.Decoder(gpu_ctx, w, h) for _ in range(26)]
    bad = bytes(frames[0][:3]) + b"\x00\x00\x00" + bytes(frames[0][6:])             # broken start code
    with pytest.raises(aa.AlfalfaError):
        gpu_ctx.submit_frames([(d, bad if k == 3 else frames[0]) for k, d in enumerate(ds)], threads=4)
    assert [d.frame_count() for d in ds] == [0 if k == 3 else 1 for k in range(26)]
    gpu_ctx.decode_batch([d for k, d in enumerate(ds) if k != 3], [0] * 25)
    assert sha256(ds[0].raster_bytes(0)) == GOLDEN["qcif_q30"]["raster_sha256"][0]
    # switched off: every frame of a big call goes to the lanes
    gpu_ctx.set_host_share_ms(0)
    try:
        off = [aa.Decoder(gpu_ctx, w, h) for _ in range(26)]
        n0 = gpu_ctx.kernel_stats()["host_routed_frames"]
        gpu_ctx.submit_frames([(d, frames[0]) for d in off], threads=4)
        assert gpu_ctx.kernel_stats()["host_routed_frames"] == n0
        gpu_ctx.decode_batch(off, [0] * 26)
        assert sha256(off[25].raster_bytes(0)) == GOLDEN["qcif_q30"]["raster_sha256"][0]
    finally:
        gpu_ctx.set_host_share_ms(80)


def test_host_lanes_take_a_big_calls_frames_without_blocking_it(gpu_ctx, monkeypatch):
    """AA_SUBMIT_HOST on a call with many streams: HOST LANES.  The frames take the device route's pre-pass (frame indices at
    once, inter frames of the same streams can follow immediately), worker threads of the context parse them the way a GPU lane
    does and finish them with the same `done` word; records and rasters are those of the all-device route -- key frames and
    inter frames, multi-partition frames, truncated frames; frames of a stream that uses segmentation stay on the GPU's lanes
    (the persistent map lives there); a frame released before its worker got to it is waited for, not lost."""
    import time
    import vp8_synth
    monkeypatch.delenv("ALFALFA_AMD_ROUTE", raising=False)
    names = ["qcif_q30_lf24", "cif_q60_lf40s5", "synth_175x143_s3", "w200_q40_lf63s7", "qcif_q30", "qvga_q100"]
    streams = [golden_frames(names[i % len(names)]) for i in range(36)]
    nf = 3
    hl = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    dev = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    before = gpu_ctx.kernel_stats()["host_routed_frames"]
    t0 = time.perf_counter()
    idx = gpu_ctx.submit_frames([(d, st[2][0]) for d, st in zip(hl, streams)], threads=8, route="host")
    assert idx == [0] * 36
    # the inter frames of the same streams at once, on the default route (the lanes): nothing waits for the key frames' parse
    gpu_ctx.submit_frames([(d, st[2][f]) for d, st in zip(hl, streams) for f in (1, 2)], threads=8)
    routed = gpu_ctx.kernel_stats()["host_routed_frames"] - before
    assert routed == 30, routed        # (synth_175x143_s3's key frame switches segmentation on: those 6 streams stay on the GPU's lanes)
    gpu_ctx.submit_frames([(d, st[2][f]) for d, st in zip(dev, streams) for f in range(nf)], threads=8, route="device")
    for i in range(36):
        for f in range(nf):
            assert_records_equal(hl[i].read_records(f), dev[i].read_records(f), "stream %d frame %d" % (i, f))
    for f in range(nf):
        gpu_ctx.decode_batch(hl, [f] * len(hl)); gpu_ctx.decode_batch(dev, [f] * len(dev))
    for i in range(36):
        want = GOLDEN[names[i % len(names)]]["raster_sha256"][nf - 1]
        assert sha256(hl[i].raster_bytes(nf - 1)) == sha256(dev[i].raster_bytes(nf - 1)) == want, i
    # whole feature streams (SPLITMV, golden / altref, 1-8 partitions; the segmentation seeds fall back to the lanes) through host lanes
    sizes = [(96, 80), (33, 17), (64, 64), (175, 143), (16, 16), (200, 48), (320, 176), (48, 256)]
    for seed in range(200, 212):
        w, h = sizes[seed % len(sizes)]
        frames = vp8_synth.feature_stream(w, h, seed, 6).frames
        ds = [aa.Decoder(gpu_ctx, w, h) for _ in range(26)]
        gpu_ctx.submit_frames([(d, fr) for d in ds for fr in frames], threads=8, route="host")
        host = aa.Parser(w, h)
        for f, fr in enumerate(frames):
            want = host.parse(fr)
            for d in (ds[0], ds[25]):
                assert_records_equal(d.read_records(f), want, "seed %d frame %d" % (seed, f))
    # released before the worker got to it (a decoder dropped right after the hand-over): the release waits for the `done` word
    w, h, frames = golden_frames("qcif_q30")
    ds = [aa.Decoder(gpu_ctx, w, h) for _ in range(40)]
    gpu_ctx.submit_frames([(d, frames[0]) for d in ds], threads=4, route="host")
    del ds
    ds = [aa.Decoder(gpu_ctx, w, h) for _ in range(40)]
    gpu_ctx.submit_frames([(d, frames[0]) for d in ds], threads=4, route="host")
    gpu_ctx.decode_batch(ds, [0] * 40)
    assert sha256(ds[39].raster_bytes(0)) == GOLDEN["qcif_q30"]["raster_sha256"][0]
