"""Timing of aa_stream_lf_search on a 1080p stream of the benchmark workload: python tools/lf_search_probe.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import alfalfa_amd as aa  # noqa: E402
import workload  # noqa: E402


def main():
    w, h, frames = aa.read_ivf(workload.make_stream("1080p_inter_lf", 4, 100))
    ctx = aa.Context(0)
    dec = aa.Decoder(ctx, w, h)
    for fr in frames[:2]:
        dec.get_frame_output(fr)
    pw, ph = dec.padded_width, dec.padded_height
    orig = np.frombuffer(dec.raster_bytes(1), np.uint8)[:pw * ph].reshape(ph, pw).copy()
    out = {}
    for lo, hi in ((23, 25), (0, 63), (23, 25), (0, 63)):        # (first use pays for the scratch decoders' first-touch allocations)
        t0 = time.perf_counter()
        best, q, qs, _ = dec.lf_search(frames[2], orig, lo, hi)
        out["%d..%d" % (lo, hi)] = {"seconds": round(time.perf_counter() - t0, 4), "best_level": best, "ssim": round(q, 6)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
