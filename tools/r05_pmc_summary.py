#!/usr/bin/env python3
"""gpurun_out/r05b/rec_{FETCH_SIZE,WRITE_SIZE,SQ_WAVES}_results.db (tools/r05_session2.sh: rocprofv3 --pmc passes of
tools/recon_replay.py -- records resident, no worker grid, every dispatch ends) -> profiles/r05_recon_counters.md and the
reconstruction / expansion entries of profiles/pmc_traffic.json.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on
gfx950 FETCH_SIZE tallies 64 bytes per 128-byte request of a wide streaming read (MI355X_MICROARCH.md, HBM); raw and corrected
figures are both printed.      python tools/r05_pmc_summary.py [dir]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_pmc_summary  # noqa: E402

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05b")
KERNELS = [("k_recon_inter4", "recon_inter", 1648), ("k_recon_intra4", "recon_intra", 1264), ("k_loopfilter_rows4", "loopfilter", 768),
           ("k_expand_coeffs", "expand", None), ("k_dense_index", "dense_index", None)]


def replay_line(name):
    for l in open(os.path.join(D, name)):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit("no replay line in " + name)


def main():
    rl = replay_line("rec_FETCH_SIZE.log")
    replays = rl["replays_incl_warmup"]
    kinds = rl["macroblocks_by_kind_per_replay"]
    units = {"recon_inter": kinds["inter_whole"] * replays, "recon_intra": kinds["intra"] * replays, "loopfilter": rl["macroblocks_per_replay"] * replays,
             "expand": rl["macroblocks_per_replay"] * replays, "dense_index": rl["macroblocks_per_replay"] * replays}
    names = [k for k, _, _ in KERNELS]
    fetch = rocpd_pmc_summary.per_kernel(os.path.join(D, "rec_FETCH_SIZE_results.db"), names)
    write = rocpd_pmc_summary.per_kernel(os.path.join(D, "rec_WRITE_SIZE_results.db"), names)
    sq = rocpd_pmc_summary.per_kernel(os.path.join(D, "rec_SQ_WAVES_results.db"), names)
    out = ["# r05 -- counters of the reconstruction and expansion kernels (rocprofv3 --pmc, one counter set per pass)\n",
           "    ALFALFA_AMD_WORKER_LINGER_MS=0 rocprofv3 --kernel-trace --pmc <set> -- python tools/recon_replay.py --streams 120 --frames 4 --reps 2\n",
           "The replay parses %d streams x %d frames of `1080p_inter_lf` on the device (worker waves leave when the queue is empty), waits for every frame, then "
           "reconstructs all of them %d times (1 warm-up + %d): k_dense_index / k_expand_coeffs on the utility stream, k_recon_inter4, k_recon_intra4, "
           "k_loopfilter_rows4 -- **no worker grid resident, every dispatch ends**, which is what a counter pass (it serialises dispatches) needs; round 4's "
           "passes profiled the whole bench pipeline and ran into their timeouts.  Macroblocks per replay: %d (%s).  replay line of the FETCH pass: `%s`\n"
           % (rl["streams"], rl["frames"], replays, rl["reps"], rl["macroblocks_per_replay"], ", ".join("%s %d" % kv for kv in kinds.items()), json.dumps(rl)),
           "## HBM traffic per macroblock\n",
           "| kernel | dispatches | avg duration us | FETCH_SIZE raw B/MB | WRITE_SIZE B/MB | corrected 2 x FETCH + WRITE | algorithmic (SURVEY 8d) | corrected / algorithmic |",
           "|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for kn, key, alg in KERNELS:
        f, w = fetch.get(kn, {}).get("FETCH_SIZE", []), write.get(kn, {}).get("WRITE_SIZE", [])
        if not f or not w or not units[key]:
            continue
        fb = sum(v for v, _ in f) * 1024.0 / units[key] * (replays * 1.0 / replays)
        wb = sum(v for v, _ in w) * 1024.0 / units[key]
        corr = 2 * fb + wb
        traffic[key] = round(corr, 1)
        dur = sum(d for _, d in f) / len(f) / 1e3
        out.append("| %s | %d | %.1f | %.1f | %.1f | **%.1f** | %s | %s |" % (kn, len(f), dur, fb, wb, corr, alg if alg else "-- (the pass packed storage adds)", "%.2f" % (corr / alg) if alg else "--"))
    path_alg = 2416
    mb_all = rl["macroblocks_per_replay"] * replays
    whole = sum(traffic[k] * units[k] for k in traffic) / mb_all
    out += ["", "Whole reconstruction half per macroblock of the replay (every kernel above weighted by the macroblocks it touched): **%.0f B**; with the token lanes' "
            "230 B and the header kernel's 166 B of round 4's passes (profiles/r04_token_workers_counters.md: those kernels' loads and stores are what they were) "
            "the path moves about **%.0f B per macroblock against 2 416 algorithmic = %.2f x**.\n" % (whole, whole + 230 + 166, (whole + 396) / path_alg),
            "## Issue counters (the same replay, SQ pass)\n",
            "| kernel | SQ_WAVES per dispatch | VALU / MB | SALU / MB | LDS instr / MB | SQ_WAVE_CYCLES per wave (x 4 = cycles) | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES |",
            "|---|---|---|---|---|---|---|"]
    for kn, key, _ in KERNELS:
        c = sq.get(kn, {})
        if not c or not units[key]:
            continue
        s = lambda n: sum(v for v, _ in c.get(n, []))
        nd = len(c.get("SQ_WAVES", [])) or 1
        out.append("| %s | %.0f | %.1f | %.1f | %.1f | %.0f | %.2f |" % (kn, s("SQ_WAVES") / nd, s("SQ_INSTS_VALU") / units[key], s("SQ_INSTS_SALU") / units[key], s("SQ_INSTS_LDS") / units[key],
                                                                       s("SQ_WAVE_CYCLES") / max(1, s("SQ_WAVES")), s("SQ_WAIT_INST_ANY") / max(1, s("SQ_WAVE_CYCLES"))))
    out += ["", "## What the SQ counters count (calibrated on these kernels: known grids, known durations)\n",
            "* **SQ_WAVES, SQ_INSTS_* are chip-wide sums, not one XCD's**: k_expand_coeffs is launched with 130 560 x 120 threads = 244 800 waves and SQ_WAVES reads 244 800; "
            "k_dense_index 30 720 threads = 480 waves, reads 480; k_loopfilter_rows4 2 176 one-wave workgroups, reads 2 176.  k_loopfilter_rows4's SQ_INSTS_VALU per "
            "macroblock is the 188 the ISA listing gives (DESIGN 4.2).  Round 4's token-worker sheet multiplied its counters by 8 (\"one of the 8 XCDs\"): "
            "that factor was wrong, and with it the \"134 VALU instructions per wave step\" and the \"74.5 % of what one wave can issue\" of DESIGN section 5.",
            "* **SQ_WAVE_CYCLES counts quad-cycles per wave** (MI355X_MICROARCH.md says so; here: k_dense_index, 480 waves that live for most of a 55-us kernel = 132 k cycles "
            "at 2.4 GHz, reads 27.9 k per wave = 112 k cycles).  **SQ_BUSY_CYCLES is summed over 32 shader engines** (k_loopfilter_rows4: 978 us = 2.35 M cycles, reads 70.8 M).",
            ""]
    open(os.path.join(ROOT, "profiles", "r05_recon_counters.md"), "w").write("\n".join(out))
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(tj))
    cfg = d.setdefault("1080p_inter_lf", {})
    src = d.setdefault("per_kernel_source", {})
    for k in ("recon_inter", "recon_intra", "loopfilter", "expand", "dense_index"):
        if k in traffic:
            cfg[k] = traffic[k]
            src[k] = "r05 PMC passes of tools/recon_replay.py (profiles/r05_recon_counters.md)"
    d["round"] = "r05"
    json.dump(d, open(tj, "w"), indent=1, sort_keys=True)
    print("\n".join(out))


if __name__ == "__main__":
    main()
