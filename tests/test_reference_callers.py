"""The drop-in boundary, tested with the reference's OWN callers: src/tests/decode-to-stdout.cc (its golden-test driver),
frontend/vp8decode.cc and frontend/decode-bundle.cc are compiled UNMODIFIED -- read from /root/reference at build time,
never copied -- against include/alfalfa_amd/compat/ (headers with the reference's names), -Wall -Wextra -Werror, and
linked to the MI355X library.  Builds happen where the reference is (this container; __graft_entry__.build() too); the
binaries travel to the GPU box under tests/cpp/_build/ like the other built artefacts."""
import hashlib
import os
import subprocess

import pytest

from conftest import GOLDEN, GOLDEN_DIR, ROOT
from test_cpp_mirror import _write_ivf, y4m_payload

REF = "/root/reference/src"
BUILD = os.path.join(ROOT, "tests", "cpp", "_build", "ref_callers")
CALLERS = {"decode-to-stdout": "tests/decode-to-stdout.cc", "vp8decode": "frontend/vp8decode.cc", "xc-decode-bundle": "frontend/decode-bundle.cc",
           "xc-dump": "frontend/xc-dump.cc"}


def build_callers():
    """-> {name: path}.  Rebuilds where the reference sources are; otherwise uses what was built before."""
    from alfalfa_amd import build as b
    b.build()
    os.makedirs(BUILD, exist_ok=True)
    libdir = os.path.dirname(b.LIB)
    out = {}
    for name, rel in CALLERS.items():
        exe, src = os.path.join(BUILD, name), os.path.join(REF, rel)
        if os.path.exists(src):
            hdr = os.path.join(ROOT, "include", "alfalfa_amd", "alfalfa.hh")
            if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(hdr), os.path.getmtime(b.LIB)):
                subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include", "alfalfa_amd", "compat"),
                                src, "-o", exe, "-L" + libdir, "-lalfalfa_amd", "-Wl,-rpath," + libdir], check=True)
        if os.path.exists(exe):
            out[name] = exe
    return out


def build_state_check():
    from test_cpp_mirror import build_exe
    return build_exe(os.path.join(ROOT, "tests", "cpp", "decoder_state_check.cc"), "decoder_state_check")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are not on this box")
def test_reference_callers_compile_unmodified_and_fail_loudly_without_gpu():
    from alfalfa_amd import capi
    exes = build_callers()
    assert sorted(exes) == sorted(CALLERS)
    build_state_check()
    if capi.device_count() == 0:
        r = subprocess.run([exes["decode-to-stdout"], os.path.join(GOLDEN_DIR, "qcif_q30.ivf")], capture_output=True)
        assert r.returncode != 0 and b"no HIP device" in r.stderr and r.stdout == b""


def _need(exes, name):
    if name not in exes:
        pytest.skip("%s was not built (build where /root/reference exists)" % name)
    return exes[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_reference_decode_to_stdout_on_our_library(name):
    """The reference's own golden test (src/tests/decoding.test: sha1sum of `decode-to-stdout FILE`), its own driver."""
    exe = _need(build_callers(), "decode-to-stdout")
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, name + ".ivf")], capture_output=True, check=True)
    assert hashlib.sha1(r.stdout).hexdigest() == GOLDEN[name]["display_sha1"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["qcif_q30_lf24", "w200_q40_lf63s7", "synth_175x143_s3"])
def test_reference_vp8decode_on_our_library(tmp_path, name):
    exe = _need(build_callers(), "vp8decode")
    out = tmp_path / "o.y4m"
    subprocess.run([exe, "-o", str(out), os.path.join(GOLDEN_DIR, name + ".ivf")], check=True)
    assert hashlib.sha1(y4m_payload(out.read_bytes(), name)).hexdigest() == GOLDEN[name]["display_sha1"]


@pytest.mark.gpu
def test_reference_vp8decode_resumes_from_a_state_file_written_by_the_reference(tmp_path):
    from conftest import golden_frames
    exe = _need(build_callers(), "vp8decode")
    name, n = "qcif_q30_lf24", 3
    _, _, frames = golden_frames(name)
    cont, rest, whole = str(tmp_path / "cont.ivf"), str(tmp_path / "rest.y4m"), str(tmp_path / "whole.y4m")
    _write_ivf(cont, name, frames[n:])
    subprocess.run([exe, "-o", whole, os.path.join(GOLDEN_DIR, name + ".ivf")], check=True)
    subprocess.run([exe, "-s", os.path.join(GOLDEN_DIR, "%s_f%d.state" % (name, n)), "-o", rest, cont], check=True)
    full, tail = y4m_payload(open(whole, "rb").read(), name), y4m_payload(open(rest, "rb").read(), name)
    frame = len(full) // sum(GOLDEN[name]["shown"])
    assert tail == full[sum(GOLDEN[name]["shown"][:n]) * frame:]


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("qcif_q30_lf24", 3), ("synth_175x143_s3", 2), ("synth_96x80_s1", 4), ("w200_q40_lf63s7", 5)])
def test_reference_xc_dump_writes_the_state_file_the_reference_writes(tmp_path, name, n):
    """frontend/xc-dump.cc (SURVEY 8b caller list), unmodified on the shim: two-step decode of the first n frames
    (UncompressedChunk, parse_frame<KeyFrame|InterFrame>, decode_frame), then Decoder::serialize -- byte for byte the .state file
    the reference decoder wrote after the same frames (tests/golden/*.state, made by oracle/_ref/ref_state)."""
    from conftest import golden_frames
    exe = _need(build_callers(), "xc-dump")
    _, _, frames = golden_frames(name)
    whole, out = str(tmp_path / "whole.ivf"), str(tmp_path / "dump.state")
    _write_ivf(whole, name, frames)             # (an IVF header that names the minihash of a fresh decoder, as xc-dump checks: xc-dump.cc:110-112)
    subprocess.run([exe, "-f", str(n - 1), whole, out], check=True)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN_DIR, "%s_f%d.state" % (name, n)), "rb").read()
    # ... and from that state on (xc-dump -S): the state after the rest of the stream equals a straight run's
    cont, a, b = str(tmp_path / "cont.ivf"), str(tmp_path / "a.state"), str(tmp_path / "b.state")
    _write_ivf(cont, name, frames[n:])
    if not name.startswith("synth"):             # (a state file carries the LAST reference only: streams of the reference encoder predict from nothing else)
        subprocess.run([exe, "-S", out, cont, a], check=True)
        subprocess.run([exe, whole, b], check=True)
        assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.gpu
def test_reference_decode_bundle_on_our_library_checks_minihash(tmp_path):
    """xc-decode-bundle: pieces named on stdin, one video on stdout; every piece's IVF header carries the minihash the decoder
    must be in when it starts (decode-bundle.cc:82-87) -- the reference's values, from tests/golden/hash_golden.json."""
    import json
    import struct
    from conftest import golden_frames
    exe = _need(build_callers(), "xc-decode-bundle")
    hashes = json.load(open(os.path.join(GOLDEN_DIR, "hash_golden.json")))
    name = "w200_q40_lf63s7"
    _, _, frames = golden_frames(name)
    cuts = [(0, 3), (3, 4), (4, len(frames))]
    paths = []
    for a, b in cuts:
        p = str(tmp_path / ("piece_%d.ivf" % a))
        _write_ivf(p, name, frames[a:b])
        if a:                                    # expected entry state = the reference's minihash after frame a-1
            data = bytearray(open(p, "rb").read())
            struct.pack_into("<I", data, 28, hashes[name]["minihash"][a - 1])
            open(p, "wb").write(data)
        paths.append(p)
    r = subprocess.run([exe], input=("\n".join(paths) + "\n").encode(), capture_output=True)
    assert r.returncode == 0, r.stderr[-500:]
    assert hashlib.sha1(y4m_payload(r.stdout, name)).hexdigest() == GOLDEN[name]["display_sha1"]
    # a wrong entry hash is refused
    data = bytearray(open(paths[1], "rb").read())
    struct.pack_into("<I", data, 28, hashes[name]["minihash"][cuts[1][0] - 1] ^ 0x10)
    open(paths[1], "wb").write(data)
    r = subprocess.run([exe], input=("\n".join(paths) + "\n").encode(), capture_output=True)
    assert r.returncode != 0 and b"Hash mismatch" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["synth_96x80_s1", "synth_175x143_s3", "cif_q60_lf40s5"])
def test_decoder_state_members_and_flat_memory(name):
    """Decoder( DecoderState, References ), get_state(), ==, get_hash(), minihash(), wire-format round trips; then ~2000
    frames through FilePlayer with every RasterHandle dying right away: HBM stays flat."""
    import json
    exe = build_state_check()
    frames = GOLDEN[name]["frames"]
    loops = 2000 // frames if name == "cif_q60_lf40s5" else 0
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, name + ".ivf"), str(loops)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    hashes = json.load(open(os.path.join(GOLDEN_DIR, "hash_golden.json")))[name]
    first = r.stdout.splitlines()[0].split()
    assert int(first[1], 16) == hashes["minihash"][-1] and int(first[3], 16) == hashes["state"][-1]


def build_two_step():
    """tests/cpp/two_step_replay.cc against the compat headers (the reference's header names), like the reference's callers."""
    from alfalfa_amd import build as b
    b.build()
    os.makedirs(BUILD, exist_ok=True)
    libdir = os.path.dirname(b.LIB)
    exe, src = os.path.join(BUILD, "two_step_replay"), os.path.join(ROOT, "tests", "cpp", "two_step_replay.cc")
    hdr = os.path.join(ROOT, "include", "alfalfa_amd", "alfalfa.hh")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(hdr), os.path.getmtime(src), os.path.getmtime(b.LIB)):
        subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include", "alfalfa_amd", "compat"),
                        src, "-o", exe, "-L" + libdir, "-lalfalfa_amd", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_two_step_replay_compiles_against_the_compat_headers():
    build_two_step()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["qcif_q30_lf24", "synth_175x143_s3", "cif_q60_lf40s5"])
def test_two_step_decode_as_xc_enc_replays_it(name):
    """UncompressedChunk + Decoder::parse_frame<F> + Decoder::decode_frame (decoder.hh:262-270), the loop of frontend/xc-enc.cc:286-300,
    and References( MutableRasterHandle && ): same rasters, same decoder hash as the one-step decode."""
    exe = build_two_step()
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, name + ".ivf")], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "MISMATCH" not in r.stdout, (r.stdout[-1500:], r.stderr[-500:])
