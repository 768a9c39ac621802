#!/usr/bin/env python3
"""rocprofv3 --pmc passes of a round -> profiles/<round>_recon_counters.md, profiles/<round>_token_workers_counters.md and
profiles/pmc_traffic.json (what bench.py's `roofline.traffic` / `path_traffic_bytes_per_mb` quote).

    python tools/pmc_summary.py r06 gpurun_out/<plan>        # the directory a `pmc` plan of tools/gpu_session.py wrote into

Expects, in that directory, one pass per counter set (gpurun refuses --pmc beside the trace domains that crash nodes; the kernel
trace alone is fine):
    rec_{FETCH_SIZE,WRITE_SIZE,SQ_WAVES}_results.db + rec_*.log     tools/recon_replay.py  (reconstruction kernels, no worker grid resident)
    tok_{FETCH_SIZE,WRITE_SIZE,SQ_WAVES}_results.db + tok_*.log     tools/parse_probe.py   (entropy decode alone, ALFALFA_AMD_WORKER_LINGER_MS=0)
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies 64 bytes per 128-byte request of a wide streaming
read (MI355X_MICROARCH.md, HBM section); raw and corrected figures are both printed.  SQ counters are chip-wide sums, SQ_WAVE_CYCLES
quad-cycles (calibrated in round 5 on kernels with known grids: profiles/r05_recon_counters.md)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_pmc_summary  # noqa: E402

ROUND, D = sys.argv[1], sys.argv[2]
RECON = [("k_recon_inter4", "recon_inter", 1648), ("k_recon_intra4", "recon_intra", 1264), ("k_loopfilter_rows4", "loopfilter", 768)]
TOK = [("k_token_workers", "parse_tokens", 880), ("k_parse_mb_headers", "parse_headers", 80)]


def line_of(name):
    for l in open(os.path.join(D, name)):
        if l.startswith("{"):
            return json.loads(l)
    raise SystemExit("no JSON line in " + name)


def db(prefix, counter):
    return os.path.join(D, "%s_%s_results.db" % (prefix, counter))


def recon(traffic):
    rl = line_of("rec_FETCH_SIZE.log")
    replays, kinds = rl["replays_incl_warmup"], rl["macroblocks_by_kind_per_replay"]
    units = {"recon_inter": kinds["inter_whole"] * replays, "recon_intra": kinds["intra"] * replays, "loopfilter": rl["macroblocks_per_replay"] * replays}
    names = [k for k, _, _ in RECON]
    fetch, write, sq = (rocpd_pmc_summary.per_kernel(db("rec", c), names) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES"))
    out = ["# %s -- counters of the reconstruction kernels (rocprofv3 --pmc, one counter set per pass)\n" % ROUND,
           "    ALFALFA_AMD_WORKER_LINGER_MS=0 rocprofv3 --kernel-trace --pmc <set> -- python tools/recon_replay.py --streams %d --frames %d --reps %d\n" % (rl["streams"], rl["frames"], rl["reps"]),
           "The replay parses the frames on the device (worker waves leave when the queue is empty), waits for every frame, then reconstructs all of them %d times "
           "(1 warm-up + %d): k_recon_inter4, k_recon_intra4, k_loopfilter_rows4 -- no worker grid resident, every dispatch ends.  **Since round 6 there is no "
           "k_dense_index / k_expand_coeffs**: the reconstruction kernels read the packed coefficient words themselves (coeff_pack.hh), so what those two kernels moved "
           "(972 B per macroblock in round 5) is gone and k_recon_*'s own fetches are the packed words instead of dense blocks.  Macroblocks per replay: %d (%s).  "
           "replay line of the FETCH pass: `%s`\n" % (replays, rl["reps"], rl["macroblocks_per_replay"], ", ".join("%s %d" % kv for kv in kinds.items()), json.dumps(rl)),
           "## HBM traffic per macroblock\n",
           "| kernel | dispatches | avg duration us | FETCH_SIZE raw B/MB | WRITE_SIZE B/MB | corrected 2 x FETCH + WRITE | algorithmic (SURVEY 8d) | corrected / algorithmic |",
           "|---|---|---|---|---|---|---|---|"]
    for kn, key, alg in RECON:
        f, w = fetch.get(kn, {}).get("FETCH_SIZE", []), write.get(kn, {}).get("WRITE_SIZE", [])
        if not f or not w or not units[key]:
            continue
        fb, wb = sum(v for v, _ in f) * 1024.0 / units[key], sum(v for v, _ in w) * 1024.0 / units[key]
        traffic[key] = round(2 * fb + wb, 1)
        out.append("| %s | %d | %.1f | %.1f | %.1f | **%.1f** | %d | %.2f |" % (kn, len(f), sum(d for _, d in f) / len(f) / 1e3, fb, wb, 2 * fb + wb, alg, (2 * fb + wb) / alg))
    mb_all = rl["macroblocks_per_replay"] * replays
    whole = sum(traffic[k] * units[k] for _, k, _ in RECON if k in traffic) / mb_all
    out += ["", "Whole reconstruction half per macroblock of the replay (every kernel weighted by the macroblocks it touched): **%.0f B** (round 5, with the expansion "
            "pass: 3 658 B).\n" % whole,
            "## Issue counters (the same replay, SQ pass)\n",
            "| kernel | SQ_WAVES per dispatch | VALU / MB | SALU / MB | LDS instr / MB | SQ_WAVE_CYCLES per wave (x 4 = cycles) | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES |", "|---|---|---|---|---|---|---|"]
    for kn, key, _ in RECON:
        c = sq.get(kn, {})
        if not c or not units[key]:
            continue
        s = lambda n: sum(v for v, _ in c.get(n, []))
        nd = len(c.get("SQ_WAVES", [])) or 1
        out.append("| %s | %.0f | %.1f | %.1f | %.1f | %.0f | %.2f |" % (kn, s("SQ_WAVES") / nd, s("SQ_INSTS_VALU") / units[key], s("SQ_INSTS_SALU") / units[key], s("SQ_INSTS_LDS") / units[key],
                                                                       s("SQ_WAVE_CYCLES") / max(1, s("SQ_WAVES")), s("SQ_WAIT_INST_ANY") / max(1, s("SQ_WAVE_CYCLES"))))
    open(os.path.join(ROOT, "profiles", "%s_recon_counters.md" % ROUND), "w").write("\n".join(out) + "\n")
    return "\n".join(out)


def tokens(traffic):
    pf, ps = line_of("tok_FETCH_SIZE.log"), line_of("tok_SQ_WAVES.log")
    mbs = pf["chains"] * 8160
    K = [k for k, _, _ in TOK]
    f, w, q = (rocpd_pmc_summary.per_kernel(db("tok", c), K) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES"))
    s = lambda d, k, n: sum(v for v, _ in d.get(k, {}).get(n, []))
    out = ["# %s -- counters of the RESIDENT token workers (and the header kernel), entropy decode alone, this round's kernel\n" % ROUND,
           "    ALFALFA_AMD_WORKER_LINGER_MS=0 ALFALFA_AMD_ROUTE=device ALFALFA_AMD_TOKEN_PROFILE=1 rocprofv3 --kernel-trace --pmc <set> -- python tools/parse_probe.py --streams 96 --frames 12 --reps 1\n",
           "Counter-friendly mode: with a linger of 0 a worker wave leaves when it has no frame and the queue is empty, so the dispatch ends.  %d chains = %d macroblocks, every frame on "
           "the GPU's lanes (key frames too), %d lanes per workgroup.  probe line of the SQ pass: `%s`\n" % (pf["chains"], mbs, ps["lanes_per_wg"], json.dumps(ps)),
           "## HBM traffic per macroblock\n",
           "| kernel | FETCH_SIZE raw B/MB | WRITE_SIZE B/MB | corrected 2 x FETCH + WRITE | algorithmic (SURVEY 8d) | corrected / algorithmic |", "|---|---|---|---|---|---|"]
    for kn, key, alg in TOK:
        fb, wb = s(f, kn, "FETCH_SIZE") * 1024.0 / mbs, s(w, kn, "WRITE_SIZE") * 1024.0 / mbs
        traffic[key] = round(2 * fb + wb, 1)
        out.append("| %s | %.1f | %.1f | **%.1f** | %d | %.2f |" % (kn, fb, wb, 2 * fb + wb, alg, (2 * fb + wb) / alg))
    out += ["", "(k_token_workers moves less than the survey's 880 B/MB: it stores PACKED coefficients -- 25 mask slots + the non-zero values of a macroblock -- where the survey's "
            "model has 800 B of dense blocks, and since round 6 nobody writes dense blocks at all: the reconstruction kernels read the packed words.)\n",
            "## Issue accounting of k_token_workers\n"]
    wave_steps, us = ps["profile"]["wave_steps"], ps["profile"]["us_per_wave_step"]
    valu, salu, lds = (s(q, "k_token_workers", n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
    wc, wi, av = (s(q, "k_token_workers", n) for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"))
    out += ["```", "SQ_WAVES %d   SQ_WAVE_CYCLES %.4g (quad-cycles, chip-wide)   SQ_INSTS_VALU %.4g   SQ_INSTS_SALU %.4g   SQ_INSTS_LDS %.4g   SQ_ACTIVE_INST_VALU %.4g   SQ_WAIT_INST_ANY %.4g"
            % (s(q, "k_token_workers", "SQ_WAVES"), wc, valu, salu, lds, av, wi),
            "wave steps of the run (in-kernel accounting, the same pass): %d at %.4f us = %.0f cycles at 2.4 GHz, %.1f lanes of a wave holding a frame" % (wave_steps, us, us * 2400, ps["profile"]["lanes_with_frame_per_period"]),
            "per wave step (block-end and boundary passes, top-ups included): %.1f VALU + %.1f SALU + %.1f LDS instructions   (round 5: 66.8 + 9.9 + 3.2)" % (valu / wave_steps, salu / wave_steps, lds / wave_steps),
            "VALU active %.0f %% of the waves' time (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, both quad-cycles), SQ_WAIT_INST_ANY %.1f %%" % (100.0 * av / max(1, wc), 100.0 * wi / max(1, wc)),
            "```", ""]
    open(os.path.join(ROOT, "profiles", "%s_token_workers_counters.md" % ROUND), "w").write("\n".join(out) + "\n")
    return "\n".join(out)


def main():
    traffic = {}
    text = []
    if os.path.exists(db("rec", "FETCH_SIZE")):
        text.append(recon(traffic))
    if os.path.exists(db("tok", "FETCH_SIZE")):
        text.append(tokens(traffic))
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = {"1080p_inter_lf": {}, "per_kernel_source": {}}
    if os.path.exists(tj):
        old = json.load(open(tj))
        if old.get("round") == ROUND:
            d = old
    for k, v in traffic.items():
        d["1080p_inter_lf"][k] = v
        d["per_kernel_source"][k] = "%s PMC passes of %s (profiles/%s_%s_counters.md)" % (ROUND, "tools/parse_probe.py" if k.startswith("parse") else "tools/recon_replay.py", ROUND,
                                                                                        "token_workers" if k.startswith("parse") else "recon")
    d["round"] = ROUND
    d["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (bytes = 2 x FETCH_SIZE + WRITE_SIZE per macroblock); which passes: per_kernel_source"
    json.dump(d, open(tj, "w"), indent=1, sort_keys=True)
    print("\n\n".join(text))


if __name__ == "__main__":
    main()
