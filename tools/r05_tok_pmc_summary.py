#!/usr/bin/env python3
"""gpurun_out/r05d/tok_{FETCH_SIZE,WRITE_SIZE,SQ_WAVES}_results.db (tools/r05_session4.sh: rocprofv3 --pmc passes of tools/parse_probe.py,
the entropy decode alone with ALFALFA_AMD_WORKER_LINGER_MS=0) -> profiles/r05_token_workers_counters.md + the parse entries of
profiles/pmc_traffic.json.      python tools/r05_tok_pmc_summary.py [dir]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_pmc_summary  # noqa: E402

D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05d")
K = ["k_token_workers", "k_parse_mb_headers"]


def probe(name):
    for l in open(os.path.join(D, name)):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit("no probe line in " + name)


def main():
    pf, pw, ps = probe("tok_FETCH_SIZE.log"), probe("tok_WRITE_SIZE.log"), probe("tok_SQ_WAVES.log")
    mbs = pf["chains"] * 8160
    f = rocpd_pmc_summary.per_kernel(os.path.join(D, "tok_FETCH_SIZE_results.db"), K)
    w = rocpd_pmc_summary.per_kernel(os.path.join(D, "tok_WRITE_SIZE_results.db"), K)
    q = rocpd_pmc_summary.per_kernel(os.path.join(D, "tok_SQ_WAVES_results.db"), K)
    s = lambda d, k, n: sum(v for v, _ in d.get(k, {}).get(n, []))
    out = ["# r05 -- counters of the RESIDENT token workers (and the header kernel), entropy decode alone, this round's kernel\n",
           "    ALFALFA_AMD_WORKER_LINGER_MS=0 ALFALFA_AMD_ROUTE=device rocprofv3 --kernel-trace --pmc <set> -- python tools/parse_probe.py --streams 96 --frames 12 --reps 1\n",
           "Counter-friendly mode: with a linger of 0 a worker wave leaves when it has no frame and the queue is empty, so the dispatch ends.  %d chains = %d macroblocks, every frame on "
           "the GPU's lanes (key frames too), 3 workgroups of 37 lanes per CU.  probe line of the SQ pass: `%s`\n" % (pf["chains"], mbs, json.dumps(ps)),
           "## HBM traffic per macroblock\n",
           "| kernel | FETCH_SIZE raw B/MB | WRITE_SIZE B/MB | corrected 2 x FETCH + WRITE | algorithmic (SURVEY 8d) | corrected / algorithmic |", "|---|---|---|---|---|---|"]
    traffic = {}
    for kn, key, alg in (("k_token_workers", "parse_tokens", 880), ("k_parse_mb_headers", "parse_headers", 80)):
        fb, wb = s(f, kn, "FETCH_SIZE") * 1024.0 / mbs, s(w, kn, "WRITE_SIZE") * 1024.0 / mbs
        traffic[key] = round(2 * fb + wb, 1)
        out.append("| %s | %.1f | %.1f | **%.1f** | %d | %.2f |" % (kn, fb, wb, 2 * fb + wb, alg, (2 * fb + wb) / alg))
    out += ["", "(k_token_workers moves less than the survey's 880 B/MB: it stores PACKED coefficients -- a mask word + the non-zero values of a block -- where the survey's model has 800 B "
            "of dense blocks; the dense blocks are written by k_expand_coeffs, whose traffic is in profiles/r05_recon_counters.md.)\n",
            "## Issue accounting of k_token_workers\n"]
    wave_steps = ps["profile"]["wave_steps"]
    valu, salu, lds = s(q, "k_token_workers", "SQ_INSTS_VALU"), s(q, "k_token_workers", "SQ_INSTS_SALU"), s(q, "k_token_workers", "SQ_INSTS_LDS")
    wc, wi, av = s(q, "k_token_workers", "SQ_WAVE_CYCLES"), s(q, "k_token_workers", "SQ_WAIT_INST_ANY"), s(q, "k_token_workers", "SQ_ACTIVE_INST_VALU")
    us = ps["profile"]["us_per_wave_step"]
    out += ["```", "SQ_WAVES %d   SQ_WAVE_CYCLES %.4g (quad-cycles, chip-wide)   SQ_INSTS_VALU %.4g   SQ_INSTS_SALU %.4g   SQ_INSTS_LDS %.4g   SQ_ACTIVE_INST_VALU %.4g   SQ_WAIT_INST_ANY %.4g"
            % (s(q, "k_token_workers", "SQ_WAVES"), wc, valu, salu, lds, av, wi),
            "wave steps of the run (in-kernel accounting, the same pass): %d at %.4f us = %.0f cycles at 2.4 GHz, %.1f lanes of a wave holding a frame" % (wave_steps, us, us * 2400, ps["profile"]["lanes_with_frame_per_period"]),
            "per wave step (counters are CHIP-WIDE sums -- calibrated in profiles/r05_recon_counters.md -- so no per-XCD factor): %.1f VALU + %.1f SALU + %.1f LDS instructions" % (valu / wave_steps, salu / wave_steps, lds / wave_steps),
            "VALU active %.0f %% of the waves' time (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, both quad-cycles: a wave64 VALU instruction of a lone wave holds its issue slot for one quad-cycle), "
            "SQ_WAIT_INST_ANY %.1f %%; the rest -- %.0f %% -- is s_waitcnt on LDS reads, scalar work and taken branches" % (100.0 * av / wc, 100.0 * wi / wc, 100.0 * (1 - av / wc - wi / wc - salu / wc)),
            "```", "",
            "What this replaces: round 4's sheet multiplied the same counters by 8 (one XCD's share, it assumed) and reported 134 VALU instructions per wave step and 74.5 percent of what "
            "one wave can issue.  The counters are chip-wide (SQ_WAVES of kernels with known grids reads the whole grid, profiles/r05_recon_counters.md), so those two figures were wrong; "
            "this sheet divides by the wave steps the kernel itself counted in the same pass.",
            ""]
    open(os.path.join(ROOT, "profiles", "r05_token_workers_counters.md"), "w").write("\n".join(out))
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(tj))
    for k, v in traffic.items():
        d["1080p_inter_lf"][k] = v
        d["per_kernel_source"][k] = "r05 PMC passes of tools/parse_probe.py (profiles/r05_token_workers_counters.md)"
    d["round"] = "r05"
    json.dump(d, open(tj, "w"), indent=1, sort_keys=True)
    print("\n".join(out))


if __name__ == "__main__":
    main()
