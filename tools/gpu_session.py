#!/usr/bin/env python3
"""One GPU session = one `gpurun` call = a PLAN of legs run one after the other on the box (replaces the per-session shell scripts
of rounds 4 and 5).

    gpurun --timeout 1500 -- 'python tools/gpu_session.py tools/plans/<plan>.txt'

A plan is a text file, one leg per line (# comments, blank lines ignored):

    name | seconds | ENV=value ENV2=value ... | command ...

The leg's stdout / stderr go to gpurun_out/<plan>/<name>.log / .err (merged back into the repo's gpurun_out/ by gpurun); the
session prints one summary line per leg: exit code, seconds, and -- when the leg printed a bench.py JSON line -- its headline
figures.  A leg that runs out of its seconds is killed (its own process group only) and the session goes on: a hung kernel costs a
leg, not the box.
"""
import json
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_summary(path):
    try:
        line = [l for l in open(path) if l.startswith("{")][-1]
        d = json.loads(line)
    except (OSError, IndexError, ValueError):
        return None
    out = {"value_M": round(d.get("value", 0) / 1e6, 2), "ms_per_step": d.get("ms_per_step")}
    t = d.get("timed_region") or {}
    if t:
        out["first_steps_ms"] = (t.get("step_done_at_ms") or [])[:4]
        out["last_ms"] = (t.get("step_done_at_ms") or [None])[-1]
        out["host"] = t.get("host_ms_per_step")
        out["wait_parse"] = t.get("host_waited_for_parse_ms_per_step")
        out["wait_compute"] = t.get("host_waited_for_compute_stream_ms_per_step")
        out["pool_wait"] = t.get("pool_wait_ms_per_step")
    ss = d.get("steady_state") or {}
    if ss:
        out["steady_M"] = round(ss.get("value", 0) / 1e6, 1)
    r = d.get("entropy_decode_roof") or {}
    if r:
        out["bools_G"] = round(r.get("sustained_bools_per_s", 0) / 1e9, 1)
        out["lanes"] = "%sx%s" % (r.get("workgroups_per_cu"), r.get("lanes_per_workgroup"))
        a = r.get("in_kernel_accounting") or {}
        out["us_step"] = a.get("us_per_wave_step"); out["lanes_busy"] = a.get("lanes_with_frame_per_period")
    v = d.get("verified_bit_exact_vs_reference") or {}
    out["bit_exact"] = v.get("bit_exact")
    m = d.get("memory") or {}
    out["hbm_gb"] = m.get("hbm_taken_by_the_context_gb")
    for k in ("small_batches", "delivery"):
        if d.get(k):
            out[k] = {a: (b.get("mb_per_s") or b.get("gb_per_s") if isinstance(b, dict) else b) for a, b in d[k].items()} if k == "small_batches" else \
                {"value_M": round(d[k].get("value", 0) / 1e6, 1), "gb_per_s": d[k].get("gb_per_s")}
    sec = d.get("secondary") or {}
    if sec:
        out["secondary_M"] = {k: round(v.get("value", 0) / 1e6, 1) for k, v in sec.items() if isinstance(v, dict)}
    return out


def main():
    plan_path = sys.argv[1]
    plan = os.path.splitext(os.path.basename(plan_path))[0]
    out_dir = os.path.join(ROOT, "gpurun_out", plan)
    os.makedirs(out_dir, exist_ok=True)
    os.chdir(ROOT)
    t_session = time.time()
    for raw in open(plan_path):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        name, seconds, env_s, cmd = [x.strip() for x in line.split("|", 3)]
        env = dict(os.environ)
        env.setdefault("ALFALFA_AMD_PARSE_TIMEOUT_S", "60")
        for kv in env_s.split():
            k, v = kv.split("=", 1)
            env[k] = v
        log, err = os.path.join(out_dir, name + ".log"), os.path.join(out_dir, name + ".err")
        t0 = time.time()
        with open(log, "w") as fo, open(err, "w") as fe:
            p = subprocess.Popen(cmd, shell=True, stdout=fo, stderr=fe, env=env, preexec_fn=os.setsid)
            try:
                rc = p.wait(timeout=float(seconds))
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)      # the leg's own process group, nothing else
                p.wait()
                rc = "timeout"
        dt = time.time() - t0
        summary = bench_summary(log)
        tail = ""
        if summary is None:
            try:
                tail = " | " + [l.rstrip() for l in open(log) if l.strip()][-1][:300]
            except (OSError, IndexError):
                pass
        print("[%6.0f s] %-28s rc=%s %5.0f s %s%s" % (time.time() - t_session, name, rc, dt, json.dumps(summary) if summary else "", tail), flush=True)


if __name__ == "__main__":
    main()
