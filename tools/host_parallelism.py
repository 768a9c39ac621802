#!/usr/bin/env python3
"""How many cores does this process really get?  (Round 4: the GPU box shows 256 hardware threads; 64 host workers parsed 15x what
one does, and 480 key frames on 256 workers took 1.3 s instead of 0.07.)  Prints what the OS says -- visible CPUs, affinity, cgroup
CPU quota -- and what T threads of the product's host parser (aa_parser_parse on a 1080p key frame, GIL released) get through.
    python tools/host_parallelism.py [ivf]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
from alfalfa_amd import capi  # noqa: E402
from ivf_io import read_ivf  # noqa: E402


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def main():
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective",
              "/sys/fs/cgroup/cpuset/cpuset.cpus"):
        print(p, "=", read(p))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_in", "1080p_inter_lf_f3_s105.ivf")
    w, h, frames = read_ivf(path)
    key = frames[0]
    L = capi.lib()
    nmb = ((w + 15) // 16) * ((h + 15) // 16)

    def run(n, out):
        mb = np.zeros(nmb * 80, np.uint8); cf = np.zeros(nmb * 25 * 16 + 16, np.int16); hdr = capi.FrameHeader()
        for _ in range(n):
            p = C.c_void_p(); L.aa_parser_create(w, h, C.byref(p))
            assert L.aa_parser_parse(p, key, len(key), C.byref(hdr), mb.ctypes.data_as(C.c_void_p), cf.ctypes.data_as(C.c_void_p)) == 0
            L.aa_parser_destroy(p)
        out.append(1)
    base = None
    for nt in (1, 4, 16, 32, 64, 128, 256):
        if nt > 2 * (os.cpu_count() or 1):
            break
        reps = 4
        o = []
        ths = [threading.Thread(target=run, args=(reps, o)) for _ in range(nt)]
        t0 = time.perf_counter()
        [t.start() for t in ths]; [t.join() for t in ths]
        dt = time.perf_counter() - t0
        rate = nt * reps / dt
        base = base or rate
        print("%3d threads: %.1f key frames/s (%.1f ms per frame per thread), %.1fx one thread" % (nt, rate, dt / reps * 1e3, rate / base), flush=True)


if __name__ == "__main__":
    main()
