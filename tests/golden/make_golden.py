#!/usr/bin/env python3
"""Regenerates tests/golden/*.ivf and golden.json with the REFERENCE (oracle/_ref, compiled in place from
/root/reference/src).  Streams: synthetic video (tools/make_y4m.py) -> reference encoder (_ref/xc-enc) ->
optional header rewrite through the reference parser/serialiser (_ref/ref_rewrite).  Expected outputs: SHA-256 of
the three padded planes of EVERY decoded frame as produced by _ref/ref_decode, plus the SHA-1 of the
decode-to-stdout dump (the form src/tests/decoding.test pins).  Run in the build container only."""
import hashlib, json, os, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_y4m, vp8_oracle as vo

REF = os.path.join(ROOT, "oracle", "_ref")
#        name                 w    h   n  seed entropy qi   quality  lf  sharp allkey
CASES = [("qcif_q30",        176, 144,  6,  7, "low",   30, "best",  -1, -1, False),
         ("qcif_q30_lf24",   176, 144,  6,  7, "low",   30, "best",  24,  0, False),
         ("qvga_q100",       320, 240,  6, 10, "low",  100, "best",  -1, -1, False),
         ("s64_q5_rt",        64,  64, 12,  3, "high",   5, "rt",    -1, -1, False),
         ("cif_q60_lf40s5",  352, 288,  5,  9, "high",  60, "rt",    40,  5, False),
         ("qcif_allkey_q20", 176, 144,  4, 21, "high",  20, "best",  30,  2, True),
         ("w200_q40_lf63s7", 208, 112,  8, 33, "high",  40, "rt",    63,  7, False)]


# synthetic feature streams (tools/vp8_synth.py): SPLITMV, golden/altref + sign bias, segmentation, 1..8 partitions,
# loop-filter deltas, hidden frames, probability updates, far-out MVs, odd sizes -- things the reference encoder never emits
#             name              w    h   seed frames
SYNTH_CASES = [("synth_96x80_s1",   96,  80,  1, 8), ("synth_175x143_s3", 175, 143, 3, 6), ("synth_33x17_s7", 33, 17, 7, 10),
               ("synth_200x48_s11", 200, 48, 11, 8), ("synth_64x64_s20", 64, 64, 20, 8)]


def run(*cmd):
    subprocess.run(list(cmd), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    import vp8_synth
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for case in CASES + SYNTH_CASES:
            if len(case) == 5:
                name, w, h, seed, n = case
                lf, allkey = -1, None
            else:
                name, w, h, n, seed, ent, qi, quality, lf, sharp, allkey = case
            ivf = os.path.join(td, name + ".ivf")
            if allkey is None:
                vo.write_ivf(ivf, w, h, vp8_synth.feature_stream(w, h, seed, n).frames)
            elif allkey:
                frames = []
                for k, planes in enumerate(make_y4m.synth_frames(w, h, n, seed, ent)):
                    y4m = os.path.join(td, "one.y4m")
                    with open(y4m, "wb") as f:
                        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\nFRAME\n" % (w, h))
                        for p in planes: f.write(p.tobytes())
                    run(os.path.join(REF, "xc-enc"), "-i", "y4m", "-y", str(qi), "-q", quality, "-o", os.path.join(td, "one.ivf"), y4m)
                    frames.append(vo.read_ivf(os.path.join(td, "one.ivf"))[2][0])
                vo.write_ivf(ivf, w, h, frames)
            else:
                y4m = os.path.join(td, name + ".y4m")
                make_y4m.write_y4m(y4m, w, h, n, seed, ent)
                run(os.path.join(REF, "xc-enc"), "-i", "y4m", "-y", str(qi), "-q", quality, "-o", ivf, y4m)
            if lf >= 0:
                run(os.path.join(REF, "ref_rewrite"), ivf, ivf + ".rw", str(lf), str(sharp)); os.replace(ivf + ".rw", ivf)
            raw = os.path.join(td, name + ".raw")
            info = vo.ref_decode(ivf, raw)
            data = open(raw, "rb").read()
            pw, ph = (w + 15) // 16 * 16, (h + 15) // 16 * 16
            fs = pw * ph * 3 // 2
            assert len(data) == fs * len(info)
            disp = subprocess.run([os.path.join(REF, "decode-to-stdout"), ivf], check=True, capture_output=True).stdout
            dst = os.path.join(HERE, name + ".ivf")
            with open(ivf, "rb") as a, open(dst, "wb") as b: b.write(a.read())
            out[name] = {"width": w, "height": h, "frames": len(info), "bytes": os.path.getsize(dst),
                         "key": [k for k, _ in info], "shown": [s for _, s in info],
                         "raster_sha256": [hashlib.sha256(data[i * fs:(i + 1) * fs]).hexdigest() for i in range(len(info))],
                         "display_sha1": hashlib.sha1(disp).hexdigest()}
            print(name, os.path.getsize(dst), "bytes", len(info), "frames")
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
