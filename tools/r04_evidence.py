#!/usr/bin/env python3
"""gpurun_out/r04*.log (the round's GPU sessions) -> profiles/: the bench lines and test logs that DESIGN.md cites, and
profiles/r04_bench_sessions.md, a table of every bench run of the round (what changed, what it measured).  python tools/r04_evidence.py"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")

RUNS = [
    ("r04a_bench", "session 1: packed default, pool+heap admission counting pending pieces as live; 600 s timeout, no line", None),
    ("r04b_bench", "session 2: + hybrid (381 of 480 key frames on 'host workers' per call), exact-size host arenas; admission starved by 50 GB of pending pieces, submit 1.1 s per step", "r04_bench_session2_hybrid_pending.log"),
    ("r04b_bench_lanes200", "session 2: host share 0, 200 GB (round 3's packed configuration with this round's runtime)", "r04_bench_lanes_only_200gb.log"),
    ("r04c_bench_deliver", "session 3: --deliver with every key frame on host workers: submit 1.45 s per step -- the box grants 16 CPUs, not 256", "r04_bench_session3_deliver_hostkeys.log"),
    ("r04d_bench", "session 4: compute-ordered recycling of rasters / dense transients, host share planned with the measured rate: pool 84 GB + heap 68 GB at a 150-GB limit", "r04_bench_session4.log"),
    ("r04d_bench_deliver", "session 4: --deliver (one gather + one copy per frame index)", "r04_bench_deliver.log"),
    ("r04e_bench", "session 5: the first group's key frames on the host route (first step at 2.7 s, second at 5.2 s: the later groups' key frames started behind the host's second)", "r04_bench_session5.log"),
    ("r04f_bench", "session 6: later groups' key frames to the lanes first, then the first group's on the host route", "r04_bench_session6.log"),
    ("r04f_ab_default", "session 6 A/B (12 steps): default", "r04_ab_default.log"),
    ("r04f_ab_strip4", "session 6 A/B: loop-filter strips of 4 macroblocks (11 KB of LDS per wave)", "r04_ab_lf_strip4.log"),
    ("r04f_ab_wgs3", "session 6 A/B: three worker workgroups per CU (27 lanes each)", "r04_ab_workers3.log"),
    ("r04f_ab_wgs3_strip4", "session 6 A/B: both", "r04_ab_workers3_lf_strip4.log"),
    ("r04g_bench", "session 7: default with the urgent host route (three later key groups to the lanes first)", "r04_bench_session7_urgent.log"),
    ("r04g_bench_again", "session 7: the same again, no secondary figures (run-to-run spread)", None),
    ("r04h_nourgent1", "session 8 (20 steps, no extras): no urgent host route, host share as measured (~43 key frames per call)", None),
    ("r04h_urgent1", "session 8: urgent host route", None),
    ("r04h_nourgent2", "session 8: no urgent host route, again", None),
    ("r04h_urgent2", "session 8: urgent host route, again", None),
    ("r04h_lanesonly", "session 8: no urgent host route, host share 0 (every frame on the lanes)", "r04_bench_lanes_only_150gb.log"),
    ("r04j_bench", "session 10: the same command once more on the final tree (run-to-run spread of the headline: 81-87 M)", "r04_bench_default_again.log"),
    ("r04k_bench", "session 11 (no extras): final tree after the last pool change (parked pieces drained at the limit)", "r04_bench_final_tree_check.log"),
    ("r04l_bench", "session 12 (no extras): priming K + 1 steps instead of D + 1 (heap_grows in the timed region 8 -> 3; the 1-s gap before the fourth step stays)", "r04_bench_final_tree_check2.log"),
    ("r04i_bench", "session 9: **the round's final tree, the driver's command** (host share: half of a call's key frames or none -> none on this box; urgent route off: a group would take 0.7 s)", "r04_bench_default.log"),
]


def line_of(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        return None


def main():
    rows = ["# r04 -- every bench.py run of the round's GPU sessions\n",
            "`python bench.py --steps 20 --warmup 5` unless the row says otherwise (A/B rows: `--steps 12 --warmup 3`, no secondary figures).  "
            "value / steady = macroblocks per second end to end (empty pipeline to empty pipeline) / between fill and drain.\n",
            "| run | what | value M | steady M | first step ms | ms/step | pool GB | heap GB | lanes busy | submit ms/step | key frames on host | bit-exact |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, what, keep in RUNS:
        d = line_of(os.path.join(G, name + ".log"))
        if d is None:
            if os.path.exists(os.path.join(G, name + ".err")) or name == "r04a_bench":
                rows.append("| %s | %s | -- | -- | -- | -- | -- | -- | -- | -- | -- | -- |" % (name, what))
            continue
        tr, mem = d["timed_region"], d["memory"]
        acc = (d["entropy_decode_roof"].get("in_kernel_accounting") or {})
        rows.append("| %s | %s | %.1f | %.1f | %d | %.0f | %.1f | %.1f | %s of %d | %.0f | %d | %s |" % (
            name, what, d["value"] / 1e6, d["steady_state"]["value"] / 1e6, tr["step_done_at_ms"][0], d["ms_per_step"], mem["pool_gb"], mem["coefficient_heap_mapped_gb"],
            acc.get("lanes_with_frame_per_period", "?"), d["entropy_decode_roof"]["lanes_per_workgroup"], tr["host_ms_per_step"]["submit"], tr.get("frames_parsed_on_host_cores", 0),
            (d.get("verified_bit_exact_vs_reference") or {}).get("bit_exact")))
        if keep:
            shutil.copy(os.path.join(G, name + ".log"), os.path.join(P, keep))
            err = os.path.join(G, name + ".err")
            if os.path.exists(err):
                shutil.copy(err, os.path.join(P, keep.replace(".log", ".stderr.log")))
    open(os.path.join(P, "r04_bench_sessions.md"), "w").write("\n".join(rows) + "\n")
    for src, dst in (("r04d_host_parallelism.log", "r04_host_parallelism.log"), ("r04g_gpu_tests.log", "r04_gpu_tests.log"), ("r04c_gpu_tests.log", "r04_gpu_tests_session3.log"),
                     ("r04b_gpu_tests.log", "r04_gpu_tests_session2_lane_per_partition_latency.log"), ("r04a_gpu_tests.log", "r04_gpu_tests_session1_xcd_probe_race.log")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    print(open(os.path.join(P, "r04_bench_sessions.md")).read())


if __name__ == "__main__":
    main()
