"""Race hunt for the in-launch hand-off of the row-pipelined kernels at full batch size: decode S x F 1080p frames R
times, hash the LAST frame of every stream (it depends on every earlier frame through the references) and demand
  * identical hashes in every repetition, and
  * for the first K streams, equality with the reference decoder (oracle/_ref/ref_decode) on the same file.
python tools/stress_check.py [--streams 240] [--frames 12] [--reps 3] [--check 6]"""
import argparse
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import alfalfa_amd as aa  # noqa: E402
import workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=240); ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--reps", type=int, default=3); ap.add_argument("--check", type=int, default=6)
    ap.add_argument("--config", default="1080p_inter_lf")
    a = ap.parse_args()
    w, h = workload.CONFIGS[a.config][:2]
    paths = workload.make_streams(a.config, a.frames, [100 + i % 120 for i in range(a.streams)])
    ctx = aa.Context(0)
    decs = []
    for p in paths:
        d = aa.Decoder(ctx, w, h)
        for fr in aa.read_ivf(p)[2]:
            d.parse_frame(fr)
        d.upload(); decs.append(d)
    ctx.sync()
    for d in decs:
        d.release_staging()
    runs = []
    for rep in range(a.reps):
        for f in range(a.frames):
            ctx.decode_batch(decs, [f] * len(decs))
        ctx.sync()
        runs.append([hashlib.sha256(d.raster_bytes(a.frames - 1)).hexdigest() for d in decs])
        for d in decs:
            d.rewind()
        print("rep %d: %d distinct final rasters" % (rep, len(set(runs[-1]))), flush=True)
    assert all(r == runs[0] for r in runs), "repetitions differ: a race"
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_decode")
    pw, ph = (w + 15) // 16 * 16, (h + 15) // 16 * 16
    fs = pw * ph * 3 // 2
    with tempfile.TemporaryDirectory() as td:
        for i in range(min(a.check, a.streams)):
            raw = os.path.join(td, "r.raw")
            subprocess.run([tool, paths[i], raw], check=True, stdout=subprocess.DEVNULL)
            with open(raw, "rb") as f:
                f.seek(fs * (a.frames - 1)); want = hashlib.sha256(f.read(fs)).hexdigest()
            assert want == runs[0][i], "stream %d differs from the reference" % i
    # streams with the same content must agree with each other too
    by_path = {}
    for p, hsh in zip(paths, runs[0]):
        assert by_path.setdefault(p, hsh) == hsh, "two decoders of one file disagree"
    print("OK: %d streams x %d frames x %d repetitions identical; %d streams equal to the reference" % (a.streams, a.frames, a.reps, min(a.check, a.streams)))


if __name__ == "__main__":
    main()
