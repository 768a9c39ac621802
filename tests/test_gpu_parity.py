"""Parity tests proper (need a real MI355X): the HIP path, called through the C ABI, against the oracle on the
same inputs and against the committed reference hashes.  Bit-exact: every byte of all three padded planes of
every decoded frame (hidden frames included)."""
import numpy as np
import pytest

import alfalfa_amd as aa
import vp8_oracle as vo
from conftest import GOLDEN, golden_frames, sha256

pytestmark = pytest.mark.gpu


def first_diff(a, b, pw, ph):
    a = np.frombuffer(a, np.uint8); b = np.frombuffer(b, np.uint8)
    bad = np.nonzero(a != b)[0]
    if not len(bad):
        return "equal"
    o = int(bad[0])
    if o < pw * ph:
        return "%d bytes differ; first: Y x=%d y=%d (mb %d,%d) got %d want %d" % (len(bad), o % pw, o // pw, (o % pw) // 16, (o // pw) // 16, a[o], b[o])
    o2 = (o - pw * ph) % (pw * ph // 4); cw = pw // 2
    return "%d bytes differ; first: chroma plane %d x=%d y=%d got %d want %d" % (len(bad), (o - pw * ph) // (pw * ph // 4), o2 % cw, o2 // cw, a[o], b[o])


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_hip_matches_oracle_and_reference(gpu_ctx, name):
    g = GOLDEN[name]
    w, h, frames = golden_frames(name)
    dec = aa.Decoder(gpu_ctx, w, h)
    ora = vo.OracleDecoder(w, h)
    for i, fr in enumerate(frames):
        shown, fi = dec.get_frame_output(fr)
        assert fi == i and shown == bool(g["shown"][i])
        ora.decode(fr)
        got, want = dec.raster_bytes(fi), ora.raster_bytes()
        assert got == want, "frame %d: %s" % (i, first_diff(got, want, dec.padded_width, dec.padded_height))
        assert sha256(got) == g["raster_sha256"][i]


def test_batched_lockstep_equals_single_stream(gpu_ctx):
    """aa_decode_batch over N streams == N independent decoders."""
    names = ["qcif_q30_lf24", "qcif_q30", "qcif_allkey_q20"]
    decs, streams = [], []
    for n in names:
        w, h, frames = golden_frames(n)
        d = aa.Decoder(gpu_ctx, w, h)
        for fr in frames[:4]:
            d.parse_frame(fr)
        d.upload()
        decs.append(d); streams.append(n)
    for f in range(4):
        gpu_ctx.decode_batch(decs, [f] * len(decs))
    for d, n in zip(decs, streams):
        for f in range(4):
            assert sha256(d.raster_bytes(f)) == GOLDEN[n]["raster_sha256"][f], (n, f)
    # replay of resident frames is idempotent (bench.py relies on it)
    for d in decs:
        d.rewind()
    for f in range(4):
        gpu_ctx.decode_batch(decs, [f] * len(decs))
    for d, n in zip(decs, streams):
        assert sha256(d.raster_bytes(3)) == GOLDEN[n]["raster_sha256"][3]


@pytest.mark.parametrize("seed", list(range(200, 224)))
def test_hip_matches_oracle_on_synthetic_feature_streams(gpu_ctx, seed, tmp_path):
    """SPLITMV / golden+altref / segmentation / multi-partition / LF deltas / hidden frames / far MVs / odd sizes."""
    import vp8_synth
    sizes = [(96, 80), (33, 17), (64, 64), (175, 143), (16, 16), (200, 48), (320, 176), (48, 256)]
    w, h = sizes[seed % len(sizes)]
    st = vp8_synth.feature_stream(w, h, seed, 8)
    dec, ora = aa.Decoder(gpu_ctx, w, h), vo.OracleDecoder(w, h)
    # ... and against the REFERENCE decoder itself where it travelled with the snapshot (oracle/_ref/ref_decode, every frame's padded
    # planes): the same seeds are oracle == reference on the CPU side too (tests/test_synth_streams.py GPU_SEEDS)
    ref = None
    if vo.ref_available():
        ivf, raw = str(tmp_path / "s.ivf"), str(tmp_path / "s.raw")
        vo.write_ivf(ivf, w, h, st.frames)
        vo.ref_decode(ivf, raw)
        ref = open(raw, "rb").read()
    fs = dec.padded_width * dec.padded_height * 3 // 2
    for i, fr in enumerate(st.frames):
        shown, fi = dec.get_frame_output(fr)
        assert ora.decode(fr) == shown
        got, want = dec.raster_bytes(fi), ora.raster_bytes()
        assert got == want, "seed %d frame %d (%s): %s" % (seed, i, ora.frame_info(), first_diff(got, want, dec.padded_width, dec.padded_height))
        if ref is not None:
            assert got == ref[i * fs:(i + 1) * fs], "seed %d frame %d: HIP differs from the reference decoder" % (seed, i)


def test_row_pipelined_schedule_under_load(gpu_ctx):
    """The in-launch ordering of the row-pipelined kernels (per-XCD tickets, progress words, hand-off through the XCD's L2)
    must give the same bytes as the launch-per-diagonal schedule when the chip is full: 16 concurrent 720p
    streams (inter frames with loop filter + an all-intra stream), every frame of every stream, repeated."""
    import hashlib
    import workload
    paths = workload.make_streams("720p_inter", 5, list(range(300, 314))) + workload.make_streams("720p_intra", 5, [400, 401])
    streams = [aa.read_ivf(p) for p in paths]
    decs = []
    for w, h, frames in streams:
        d = aa.Decoder(gpu_ctx, w, h)
        for fr in frames:
            d.parse_frame(fr)
        d.upload(); decs.append(d)

    def run(schedule):
        gpu_ctx.set_schedule(schedule)
        for d in decs:
            d.rewind()
        for f in range(5):
            gpu_ctx.decode_batch(decs, [f] * len(decs))
        gpu_ctx.sync()
        return [[hashlib.sha256(d.raster_bytes(f)).hexdigest() for f in range(5)] for d in decs]

    try:
        want = run("diagonal")
        # one stream is also pinned to the oracle so that "both schedules wrong in the same way" cannot pass
        ora = vo.OracleDecoder(streams[0][0], streams[0][1])
        for f, fr in enumerate(streams[0][2]):
            ora.decode(fr)
            assert hashlib.sha256(ora.raster_bytes()).hexdigest() == want[0][f]
        for rep in range(4):
            got = run("rows")
            assert got == want, "repeat %d: row-pipelined output differs" % rep
    finally:
        gpu_ctx.set_schedule("rows")



def test_release_staging_then_parse_more(gpu_ctx):
    """aa_stream_release_staging gives the pinned staging back (and to the context's pool); frames parsed afterwards are
    staged in fresh (pooled) pinned memory and decode exactly as before."""
    name = "w200_q40_lf63s7"
    w, h, frames = golden_frames(name)
    a, b = aa.Decoder(gpu_ctx, w, h), aa.Decoder(gpu_ctx, w, h)
    for fr in frames[:3]:
        a.parse_frame(fr)
    a.release_staging()                      # uploads, waits, frees the host side
    for fr in frames[:3]:
        b.parse_frame(fr)                    # reuses the pooled pinned chunk
    for f in range(3):
        gpu_ctx.decode_batch([a, b], [f, f])
    b.release_staging()
    for fr in frames[3:]:
        a.parse_frame(fr); b.parse_frame(fr)
    for f in range(3, len(frames)):
        gpu_ctx.decode_batch([a, b], [f, f])
    for f in range(len(frames)):
        assert sha256(a.raster_bytes(f)) == GOLDEN[name]["raster_sha256"][f], f
        assert sha256(b.raster_bytes(f)) == GOLDEN[name]["raster_sha256"][f], f


def test_row_kernels_do_not_depend_on_residency():
    """Deadlock freedom and ordering of the ticketed row kernels must not rely on every workgroup being resident:
    ALFALFA_AMD_TEST_LDS_PAD adds 100 KB of dynamic LDS to their launches (one workgroup per CU = 256 resident waves for
    540 macroblock rows per launch); same bytes as the unconstrained run.  The hook is read once per process -> subprocesses."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = (
        "import sys, hashlib, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import alfalfa_amd as aa, workload\n"
        "paths = workload.make_streams('720p_inter', 3, list(range(300, 348)))\n"
        "ctx = aa.Context(0); decs = []\n"
        "for p in paths:\n"
        "    w, h, frames = aa.read_ivf(p); d = aa.Decoder(ctx, w, h)\n"
        "    for fr in frames: d.parse_frame(fr)\n"
        "    decs.append(d)\n"
        "for f in range(3): ctx.decode_batch(decs, [f] * len(decs))\n"
        "ctx.sync()\n"
        "print(json.dumps([hashlib.sha256(d.raster_bytes(2)).hexdigest() for d in decs]))\n" % (ROOT, os.path.join(ROOT, "tools")))
    outs = []
    for pad in (None, "100000"):
        env = dict(os.environ)
        env.pop("ALFALFA_AMD_TEST_LDS_PAD", None)
        if pad:
            env["ALFALFA_AMD_TEST_LDS_PAD"] = pad
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] and len(set(outs[0])) > 1


@pytest.mark.parametrize("name", ["qcif_q30_lf24", "w200_q40_lf63s7"])
def test_hip_matches_oracle_on_truncated_frames(gpu_ctx, name):
    """Frames whose partitions end early (zeros are read past the end, bool_decoder.hh:56-65): garbage in, the SAME garbage out."""
    from test_parser_vs_oracle import truncated
    w, h, frames = golden_frames(name)
    dec, ora = aa.Decoder(gpu_ctx, w, h), vo.OracleDecoder(w, h)
    for i, fr in enumerate(truncated(frames)):
        _, fi = dec.get_frame_output(fr)
        ora.decode(fr)
        assert dec.raster_bytes(fi) == ora.raster_bytes(), (name, i)


def test_hip_matches_oracle_on_large_and_extreme_geometries(gpu_ctx):
    """2560x1440 (160x90 macroblocks, more than two 8-macroblock strips per row and rows beyond one residency round for a
    single group), a one-macroblock-wide 16x4096 column and a one-row 4096x16 strip (the degenerate wavefronts)."""
    import vp8_synth
    import workload
    cases = [aa.read_ivf(workload.make_stream("1440p_inter_lf", 3, 500))]
    for w, h, seed in ((16, 4096, 901), (4096, 16, 902), (24, 1000, 903)):
        cases.append((w, h, vp8_synth.feature_stream(w, h, seed, 4).frames))
    for w, h, frames in cases:
        dec, ora = aa.Decoder(gpu_ctx, w, h), vo.OracleDecoder(w, h)
        for i, fr in enumerate(frames):
            shown, fi = dec.get_frame_output(fr)
            assert ora.decode(fr) == shown
            got, want = dec.raster_bytes(fi), ora.raster_bytes()
            assert got == want, "%dx%d frame %d: %s" % (w, h, i, first_diff(got, want, dec.padded_width, dec.padded_height))


def test_first_allocation_of_a_context_is_a_big_piece():
    """Device pool: a raster above 128 MiB gets an allocation of its own.  When that is the FIRST allocation of a context, the
    next ordinary request must still open a slab (round-1 bug: it was carved out of a slab that did not exist)."""
    ctx = aa.Context(0)
    big = aa.Decoder(ctx, 12000, 8000)            # 96 M luma pixels: one raster = 144 MB
    w, h, frames = golden_frames("qcif_q30_lf24")
    small = aa.Decoder(ctx, w, h)
    for i, fr in enumerate(frames[:4]):
        _, fi = small.get_frame_output(fr)
        assert sha256(small.raster_bytes(fi)) == GOLDEN["qcif_q30_lf24"]["raster_sha256"][i]
    del big, small


def test_rasters_released_while_binding_are_not_recycled_inside_the_same_launch(gpu_ctx):
    """Round-2 review: with an idle compute stream and every handle released after every step, a raster that bind_frame gives
    back (an old reference) could come out of the pool again as ANOTHER stream's output raster of the same aa_decode_batch --
    one frame then read a plane another was writing.  Many streams, key / inter / inter ..., release + sync after every step."""
    name = "qcif_q30_lf24"
    w, h, frames = golden_frames(name)
    decs = [aa.Decoder(gpu_ctx, w, h) for _ in range(12)]
    for f, fr in enumerate(frames[:6]):
        idx = [d.parse_frame(fr)[0] for d in decs]
        gpu_ctx.decode_batch(decs, idx)
        got = [sha256(d.raster_bytes(f)) for d in decs]
        assert all(g == GOLDEN[name]["raster_sha256"][f] for g in got), (f, [g == GOLDEN[name]["raster_sha256"][f] for g in got])
        for d in decs:
            d.release_before(f + 1)
        gpu_ctx.sync()


def test_a_whole_frame_index_is_delivered_by_one_gather_and_one_copy(gpu_ctx):
    """aa_download_batch_async: the rasters of n decoders in lock step -- of different sizes -- gathered on the device and copied to
    pinned host memory in one piece; every raster byte for byte what aa_stream_download gives, and the reference's hash."""
    import ctypes as C
    names = ["qcif_q30_lf24", "cif_q60_lf40s5", "synth_175x143_s3", "w200_q40_lf63s7", "qcif_q30"]
    streams = [golden_frames(n) for n in names]
    nf = min(len(f) for _, _, f in streams)
    decs = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    sizes = [sum(d.plane_sizes()) for d in decs]
    stride = (max(sizes) + 255) & ~255
    ring = [gpu_ctx.pinned_alloc(stride * len(decs)) for _ in range(2)]
    try:
        for f in range(nf):
            for d, (_, _, frames) in zip(decs, streams):
                d.parse_frame(frames[f])
            gpu_ctx.decode_batch(decs, [f] * len(decs))
            gpu_ctx.download_batch_async(decs, [f] * len(decs), ring[f & 1], stride)      # (arrives while the next frame index is decoded)
        gpu_ctx.download_wait()
        for f in (nf - 2, nf - 1):
            for i, (d, n) in enumerate(zip(decs, names)):
                got = C.string_at(ring[f & 1] + i * stride, sizes[i])
                assert got == d.raster_bytes(f), (n, f)
                assert sha256(got) == GOLDEN[n]["raster_sha256"][f], (n, f)
    finally:
        gpu_ctx.sync()
        for p in ring:
            gpu_ctx.pinned_free(p)
    with pytest.raises(aa.AlfalfaError):                    # a stride smaller than a raster is refused, nothing is written
        gpu_ctx.download_batch_async(decs, [0] * len(decs), 0x1000, 16)


def test_a_ring_of_destination_buffers_waits_for_the_copy_that_used_the_buffer(gpu_ctx):
    """aa_ctx_download_wait_until(r - 1) in front of the reuse of one of r destination buffers: every frame index, checked as soon as
    the call says its copy has arrived (i.e. r - 1 frame indices later), is the raster aa_stream_download gives."""
    import ctypes as C
    names = ["qcif_q30_lf24", "cif_q60_lf40s5", "qcif_q30"]
    streams = [golden_frames(n) for n in names]
    nf = min(len(f) for _, _, f in streams)
    decs = [aa.Decoder(gpu_ctx, w, h) for w, h, _ in streams]
    sizes = [sum(d.plane_sizes()) for d in decs]
    stride = (max(sizes) + 255) & ~255
    R = 3
    ring = [gpu_ctx.pinned_alloc(stride * len(decs)) for _ in range(R)]
    checked = 0

    def check(f):
        for i, (d, n) in enumerate(zip(decs, names)):
            assert C.string_at(ring[f % R] + i * stride, sizes[i]) == d.raster_bytes(f), (n, f)
    try:
        for f in range(nf):
            for d, (_, _, frames) in zip(decs, streams):
                d.parse_frame(frames[f])
            gpu_ctx.decode_batch(decs, [f] * len(decs))
            gpu_ctx.download_wait(R - 1)            # the copy of frame index f - R (this slab's last user) is through
            if f >= R:
                check(f - R); checked += 1
            gpu_ctx.download_batch_async(decs, [f] * len(decs), ring[f % R], stride)
        gpu_ctx.download_wait(0)
        for f in range(max(0, nf - R), nf):
            check(f); checked += 1
        assert checked == nf
        with pytest.raises(aa.AlfalfaError):
            gpu_ctx.download_wait(-1)
    finally:
        gpu_ctx.sync()
        for p in ring:
            gpu_ctx.pinned_free(p)


def test_an_expired_row_kernel_wait_is_reported_with_what_the_waiting_wave_saw():
    """Fault injection into the row kernels' in-launch hand-off (ALFALFA_AMD_LF_DEBUG bit 5: row 1 of every unit never publishes its
    progress; bits 8-20: the wait's bounds brought down to a second and 2^16 polls): row 2 must give up -- not hang --, the call that
    synchronises must fail with AA_ERR_HIP, and the message must carry what the waiting wave saw of its unit's rows (row 0 complete,
    row 1 silent).  A process of its own: the hook is read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import alfalfa_amd as aa
from conftest import golden_frames
w, h, frames = golden_frames("qcif_q30_lf24")
ctx = aa.Context(0)
decs = [aa.Decoder(ctx, w, h) for _ in range(4)]
try:
    for f in range(2):
        for d in decs:
            d.parse_frame(frames[f])
        ctx.decode_batch(decs, [f] * len(decs))
    ctx.sync()
    print("NO ERROR")
except aa.AlfalfaError as e:
    print("ERROR %s: %s" % (e.kind, e))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALFALFA_AMD_LF_DEBUG=str(32 | (4 << 8) | (16 << 16)))
    r = subprocess.run([sys.executable, "-c", code, root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:]
    assert out.startswith("ERROR HipError"), out
    assert "k_loopfilter_rows4: a bounded wait for the macroblock row above expired" in out, out
    import re
    assert re.search(r"first at unit 0 row [2-8], needed 1 saw 0", out), out        # (rows 2 .. 8 all wait; whichever gives up first says so)
    assert "columns done per row of that unit when it gave up: 11,0,0,0,0,0,0,0,0" in out, out
