"""Synthetic bench/test workloads: deterministic synthetic video -> reference encoder
(oracle/_ref/xc-enc, built in place from the reference) -> header rewrite forcing the loop filter on
(oracle/_ref/ref_rewrite).  Stream generation is tooling around the hot path, not part of it; results are cached
on disk so that repeated bench runs on one box do not re-encode.  Lives in tools/ (not in the product package):
the product never generates or encodes streams.
"""
import hashlib
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ivf_io import read_ivf, write_ivf  # noqa: E402

CONFIGS = {
    # name: (width, height, entropy, y_ac_qi, loop_filter_level, sharpness)   -- BASELINE.json configs[1..3]
    "720p_intra": (1280, 720, "low", 30, 24, 0),
    "720p_inter": (1280, 720, "low", 30, 24, 0),
    "1080p_inter_lf": (1920, 1080, "high", 20, 24, 0),
    "1080p_inter_lf_lowentropy": (1920, 1080, "low", 40, 24, 0),
    # not from the reference encoder (which only emits full-pel LAST-reference vectors): tools/vp8_synth.perf_stream --
    # quarter-pel vectors, ~15 % SPLITMV, golden / altref, four partitions
    "1080p_inter_lf_subpel": (1920, 1080, "synth", 20, 24, 0),
    "cif_inter_lf_subpel": (352, 288, "synth", 20, 24, 0),
    "cif_inter_lf": (352, 288, "high", 20, 24, 0),          # small stand-in for quick checks
    "1440p_inter_lf": (2560, 1440, "high", 20, 24, 0),      # bigger than any bench config: parity test only
    # geometry probes for tools/row_kernel_probe.py: one MB row (no cross-row waits) / one MB column (pure hand-off chain)
    "probe_1row": (1920, 16, "high", 20, 24, 0),
    "probe_1col": (16, 1088, "high", 20, 24, 0),
    "probe_4rows": (1920, 64, "high", 20, 24, 0),
}


def cache_dir():
    d = os.environ.get("ALFALFA_AMD_CACHE", os.path.join(tempfile.gettempdir(), "alfalfa_amd_streams"))
    os.makedirs(d, exist_ok=True)
    return d


def have_reference_tools():
    return all(os.path.exists(os.path.join(REF, t)) for t in ("xc-enc", "ref_rewrite"))


def make_stream(config, frames, seed):
    """-> path of an IVF (1 key frame + frames-1 inter frames; all key frames for *_intra)."""
    key = "%s_f%d_s%d" % (config, frames, seed)
    path = os.path.join(cache_dir(), key + ".ivf")
    if os.path.exists(path):
        return path
    shipped = os.path.join(ROOT, "gpurun_in", "streams", key + ".ivf")     # (pre-generated here, travels with the snapshot: the pure-Python writer is slow)
    if os.path.exists(shipped):
        return shipped
    # several ranks of one node may ask for the same stream at the same time: the first takes a lock file, the others wait
    lock = path + ".lock"
    try:
        os.close(os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY))
    except FileExistsError:
        import time
        deadline = time.time() + 600
        while time.time() < deadline:
            if os.path.exists(path):
                return path
            if not os.path.exists(lock):
                break
            time.sleep(0.2)
        return path if os.path.exists(path) else _generate(config, frames, seed, path, None)     # stale lock: generate anyway
    try:
        return _generate(config, frames, seed, path, lock)
    finally:
        if os.path.exists(lock):
            os.unlink(lock)


def _generate(config, frames, seed, path, lock):
    import make_y4m
    w, h, ent, qi, lf, sharp = CONFIGS[config]
    if os.path.exists(path):
        return path
    if ent == "synth":                                       # our own bitstream writer, no encoder involved
        import vp8_synth
        st = vp8_synth.perf_stream(w, h, seed, frames)
        write_ivf(path + ".tmp", w, h, st.frames)
        os.replace(path + ".tmp", path)
        return path
    if not have_reference_tools():
        raise RuntimeError("oracle/_ref/xc-enc is missing: build it with `make -C oracle ref` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        y4m = os.path.join(td, "in.y4m"); raw = os.path.join(td, "raw.ivf"); out = os.path.join(td, "out.ivf")
        if config.endswith("_intra"):
            chunks = []
            for planes in make_y4m.synth_frames(w, h, frames, seed, ent):
                with open(y4m, "wb") as f:
                    f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\nFRAME\n" % (w, h))
                    for p in planes:
                        f.write(p.tobytes())
                subprocess.run([os.path.join(REF, "xc-enc"), "-i", "y4m", "-y", str(qi), "-q", "rt", "-o", raw, y4m],
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                chunks.append(read_ivf(raw)[2][0])
            write_ivf(raw, w, h, chunks)
        else:
            make_y4m.write_y4m(y4m, w, h, frames, seed, ent)
            subprocess.run([os.path.join(REF, "xc-enc"), "-i", "y4m", "-y", str(qi), "-q", "rt", "-o", raw, y4m],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run([os.path.join(REF, "ref_rewrite"), raw, out, str(lf), str(sharp)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.replace(out, path + ".tmp"); os.replace(path + ".tmp", path)
    return path


def make_streams(config, frames, seeds, workers=None):
    workers = workers or min(len(seeds), os.cpu_count() or 1, 64)
    if CONFIGS[config][2] == "synth":       # pure-Python writer: processes, not threads; distinct seeds once each
        todo = sorted({s for s in seeds if not os.path.exists(os.path.join(cache_dir(), "%s_f%d_s%d.ivf" % (config, frames, s)))
                       and not os.path.exists(os.path.join(ROOT, "gpurun_in", "streams", "%s_f%d_s%d.ivf" % (config, frames, s)))})
        if todo:
            procs = []
            for s in todo:
                procs.append(subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import workload; workload.make_stream(%r, %d, %d)"
                                               % (os.path.join(ROOT, "tools"), config, frames, s)]))
                while sum(p.poll() is None for p in procs) >= workers:
                    import time
                    time.sleep(0.05)
            for p in procs:
                if p.wait():
                    raise RuntimeError("stream generation failed")
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(lambda s: make_stream(config, frames, s), seeds))


def sha256_file(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()
