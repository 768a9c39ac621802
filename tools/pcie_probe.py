#!/usr/bin/env python3
"""What the bus gives a device -> pinned-host copy on this box (the `delivery` leg of bench.py moves one frame index of every stream,
~1.5 GB, per copy): one copy on one stream, the same bytes split over 2 / 4 / 8 streams (copy engines side by side), the other
direction, and both directions at once.  Prints one JSON line.

    python tools/pcie_probe.py [--mb 1500] [--reps 5]"""
import argparse
import json
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=1500)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    n = a.mb * 1000 * 1000
    dev = torch.empty(n, dtype=torch.uint8, device="cuda:0").random_(0, 255)
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    out = {"bytes": n}
    for ways in (1, 2, 4, 8):
        streams = [torch.cuda.Stream() for _ in range(ways)]
        cut = [n * k // ways for k in range(ways + 1)]
        best = 1e9
        for _ in range(a.reps + 1):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    host[cut[k]:cut[k + 1]].copy_(dev[cut[k]:cut[k + 1]], non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        out["d2h_%d_streams_gb_per_s" % ways] = round(n / best / 1e9, 2)
    # host -> device for comparison
    best = 1e9
    for _ in range(a.reps + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        dev.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    out["h2d_1_stream_gb_per_s"] = round(n / best / 1e9, 2)
    # both directions at once
    dev2 = torch.empty_like(dev); host2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    best = 1e9
    for _ in range(a.reps + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        with torch.cuda.stream(s1):
            host.copy_(dev, non_blocking=True)
        with torch.cuda.stream(s2):
            dev2.copy_(host2, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    out["d2h_beside_h2d_gb_per_s_each"] = round(n / best / 1e9, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
