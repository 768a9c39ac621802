#!/usr/bin/env python3
"""profiles/<round>_bench_sessions.md: every bench.py run of a round's GPU sessions, one row each, from the plans under tools/plans/
(what ran: the plan's header line, the leg's environment and arguments) and the logs gpurun merged back under gpurun_out/<plan>/.
Also copies each leg's log to profiles/<round>_<plan>_<leg>.log when --copy is given (the judged evidence lives under profiles/).

    python tools/sessions_table.py r06 [--copy]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1]
COPY = "--copy" in sys.argv


def bench_line(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        return None


def main():
    plans = sorted(p for p in os.listdir(os.path.join(ROOT, "tools", "plans")) if p.startswith(ROUND + "_"))
    plans.sort(key=lambda p: int("".join(c for c in p.split("_s")[-1] if c.isdigit()) or 0))
    out = ["# %s -- every bench.py run of the round's GPU sessions\n" % ROUND,
           "One row per leg of the plans under `tools/plans/` (run by `tools/gpu_session.py`, one plan = one `gpurun` call).  value / steady = macroblocks per second end to end "
           "(empty pipeline to empty pipeline) / between fill and drain; waits = what the host waited for per step (device parser / compute stream); bools = entropy decode "
           "sustained over the timed region.  `--steps 20 --warmup 5` unless the arguments say otherwise; legs that ran tests or counter passes have no row.\n",
           "| session / leg | what (environment; arguments beyond the A/B set) | value M | steady M | first step ms | last step ms | ms/step | waits parse / compute ms | lanes | us per wave step | lanes busy | bools G/s | HBM GB | bit-exact |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    ab = "--secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --deliver-steps 0 --no-device-half"
    for plan in plans:
        name = plan[:-4]
        header = ""
        for raw in open(os.path.join(ROOT, "tools", "plans", plan)):
            line = raw.strip()
            if line.startswith("#"):
                header = header or line.lstrip("# ")
                continue
            if not line:
                continue
            leg, _, env_s, cmd = [x.strip() for x in line.split("|", 3)]
            log = os.path.join(ROOT, "gpurun_out", name, leg + ".log")
            d = bench_line(log)
            if COPY and os.path.exists(log) and (d or "pytest" in cmd):
                shutil.copy(log, os.path.join(ROOT, "profiles", "%s_%s.log" % (name, leg)))
            if not d or "bench.py" not in cmd:
                continue
            args = cmd.split("bench.py", 1)[1].replace(ab, "").replace("--steps 20 --warmup 5", "").strip()
            t, r, m = d.get("timed_region") or {}, d.get("entropy_decode_roof") or {}, d.get("memory") or {}
            a = r.get("in_kernel_accounting") or {}
            done = t.get("step_done_at_ms") or [None]
            what = "; ".join(x for x in (env_s, args) if x) or "defaults of the tree at that session"
            out.append("| %s / %s | %s | %.1f | %.1f | %s | %s | %.0f | %.0f / %.0f | %sx%s | %s | %s | %.1f | %s | %s |" % (
                name, leg, what, d["value"] / 1e6, (d.get("steady_state") or {}).get("value", 0) / 1e6, done[0], done[-1], d["ms_per_step"],
                t.get("host_waited_for_parse_ms_per_step", 0), t.get("host_waited_for_compute_stream_ms_per_step", 0), r.get("workgroups_per_cu"), r.get("lanes_per_workgroup"),
                a.get("us_per_wave_step"), a.get("lanes_with_frame_per_period"), r.get("sustained_bools_per_s", 0) / 1e9, m.get("hbm_taken_by_the_context_gb"),
                (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact")))
        out.append("| | *%s: %s* | | | | | | | | | | | | |" % (name, header))
    open(os.path.join(ROOT, "profiles", "%s_bench_sessions.md" % ROUND), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
