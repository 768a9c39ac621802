#!/usr/bin/env python3
"""N fresh processes, each a short pipelined bench.py run that starts on a new context (first-time growth of the pool and the
coefficient heap under the first reconstruction kernels: the "priming pass" in which every expired row-kernel wait of rounds 5 and 6
happened).  Prints one line per run -- value, whether it ended, row_handoff_rereads (waits that only the slow path's second look
ended, kernels.hip) -- and a summary.

    python tools/priming_stress.py [--runs 20] [--steps 2] [--warmup 2]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", str(a.warmup), "--secondary=", "--small-batches=", "--no-cpu-baseline",
           "--lanes-only-steps", "0", "--deliver-steps", "0", "--two-cpu-steps", "0", "--no-device-half"]
    failed = rescued_runs = rescues = 0
    for i in range(a.runs):
        t = time.time()
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, ALFALFA_AMD_HANDOFF_REPORT="1"))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {}
        n = (d.get("timed_region") or {}).get("row_handoff_rereads_since_context_creation")
        stale = (d.get("timed_region") or {}).get("of_which_the_poll_repeated_was_still_stale")
        ok = r.returncode == 0 and bool(d)
        failed += not ok
        if n:
            rescued_runs += 1; rescues += n
        err = "" if ok else (r.stderr.strip().splitlines() or ["?"])[-1][:600]
        rep = [l for l in r.stderr.splitlines() if "row hand-off waits" in l]
        if rep:
            err = rep[-1][12:] + " " + err
        print("run %2d  rc %d  %5.1f s  value %6.1f M  bit-exact %s  rereads %s (poll still stale: %s)  %s" % (i, r.returncode, time.time() - t, d.get("value", 0) / 1e6,
              (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact"), n, stale, err), flush=True)
    print(json.dumps({"runs": a.runs, "failed": failed, "runs_with_rereads": rescued_runs, "rereads": rescues}))


if __name__ == "__main__":
    main()
