#!/usr/bin/env python3
"""Decoder-state fixtures written and consumed by the REFERENCE (oracle/_ref/ref_state = Decoder::serialize and
EncoderStateDeserializer::build<Decoder> of /root/reference/src, compiled in place): for (stream, N)
    <stream>_f<N>.state      the reference decoder serialised after N frames
    state_golden.json        SHA-256 of the file and of the padded planes of every frame the REFERENCE decodes when it
                             resumes from that file (golden and alternative alias LAST after loading, decoder.cc:171-175,
                             so this differs from a straight decode when the stream uses them)
Run in the build container only (needs oracle/_ref)."""
import hashlib, json, os, subprocess, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_state")
CASES = [("qcif_q30_lf24", 3), ("synth_175x143_s3", 2), ("synth_96x80_s1", 4), ("w200_q40_lf63s7", 5)]


def main():
    golden = json.load(open(os.path.join(HERE, "golden.json")))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, n in CASES:
            ivf = os.path.join(HERE, name + ".ivf")
            state = os.path.join(HERE, "%s_f%d.state" % (name, n))
            subprocess.run([TOOL, "save", ivf, str(n), state], check=True)
            raw = os.path.join(td, "r.raw")
            subprocess.run([TOOL, "resume", ivf, str(n), state, raw], check=True)
            g = golden[name]
            pw, ph = (g["width"] + 15) // 16 * 16, (g["height"] + 15) // 16 * 16
            fs = pw * ph * 3 // 2
            data = open(raw, "rb").read()
            assert len(data) == fs * (g["frames"] - n)
            out["%s_f%d" % (name, n)] = {"stream": name, "frames_before": n, "state_sha256": hashlib.sha256(open(state, "rb").read()).hexdigest(),
                                         "resumed_raster_sha256": [hashlib.sha256(data[i * fs:(i + 1) * fs]).hexdigest() for i in range(g["frames"] - n)]}
            print(name, n, os.path.getsize(state), "bytes")
    json.dump(out, open(os.path.join(HERE, "state_golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
