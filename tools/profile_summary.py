#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_<config>/ (tools/profile_round.sh) -> profiles/<tag>_<config>.md (kernel trace table, counters per
launch, HBM traffic per macroblock) and profiles/pmc_traffic.json (what bench.py reports as roofline.traffic for that config).
HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes: on gfx950 FETCH_SIZE counts 64 bytes per 128-byte request of a wide
streaming read (MI355X_MICROARCH.md, HBM); both the raw and the corrected figure are printed.
python tools/profile_summary.py r02 1080p_inter_lf [...]"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary  # noqa: E402

KINDS = collections.OrderedDict([("k_recon_inter4", "recon_inter"), ("k_recon_inter(", "recon_split"), ("k_recon_intra4", "recon_intra"),
                                 ("k_loopfilter_rows4", "loopfilter"), ("k_token_workers", "parse_tokens"), ("k_parse_mb_headers", "parse_headers")])


def kind_of(name):
    for k, v in KINDS.items():
        if k in name:
            return v
    return None


def per_kernel(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_pmc_event")][0][len("rocpd_pmc_event"):]
    q = f"""select d.id, ks.kernel_name, p.name, sum(e.value), d.end-d.start from rocpd_pmc_event{suf} e
      join rocpd_info_pmc{suf} p on e.pmc_id=p.id join rocpd_kernel_dispatch{suf} d on e.event_id=d.event_id
      join rocpd_info_kernel_symbol{suf} ks on d.kernel_id=ks.id group by 1,3 order by 1"""
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for _did, kn, pn, v, dur in c.execute(q):
        k = kind_of(kn)
        if k:
            by[k][pn].append((v, dur))
    return by


def main():
    tag, configs = sys.argv[1], sys.argv[2:]
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for cfg in configs:
        d = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, cfg))
        line = [l for l in open(os.path.join(d, "kt.log")) if l.startswith("{")][-1]
        bench = json.loads(line)
        units, lps = bench["units_per_step"], bench["launches_per_step"]
        out = ["# %s -- %s: rocprofv3 kernel trace and counters of bench.py\n" % (tag, cfg),
               "Command (tools/profile_round.sh; every counter set in its own pass):",
               "    python bench.py --config %s --steps 2 --warmup 0 --small-batches= --no-cpu-baseline --no-verify\n" % cfg,
               "bench line of the kernel-trace pass: value %.1f %s, ms_per_step %.1f, steady state %s\n" % (bench["value"], bench["unit"], bench["ms_per_step"], bench.get("steady_state")),
               "## Kernel trace (tools/rocpd_summary.py)\n", rocpd_summary.summarise(os.path.join(d, "kt_results.db")), ""]
        out += ["## HIP-event timing inside bench.py of the same run (must agree with the trace)\n", "```",
                json.dumps({k: (v and {"avg_launch_us": v["avg_launch_us"], "launches_per_step": v["launches_per_step"], "frac_of_hbm_peak": v["frac"]}) for k, v in bench["kernels"].items()}, indent=1), "```", ""]
        tot = collections.defaultdict(dict)

        def bench_of(name):
            try:
                return json.loads([l for l in open(os.path.join(d, name + ".log")) if l.startswith("{")][-1])
            except (OSError, IndexError, ValueError):
                return {}
        fetch_bench, write_bench = bench_of("fetch"), bench_of("write")
        if fetch_bench:
            units, lps = fetch_bench["units_per_step"], fetch_bench["launches_per_step"]     # (the counter passes run their own, smaller workload)
        out += ["## Counters per launch, summed over the chip\n", "```"]
        for name in ("fetch", "write", "sq"):
            db = os.path.join(d, name + "_results.db")
            if not os.path.exists(db):
                continue
            for k, v in sorted(per_kernel(db).items()):  # noqa
                for pn, vals in sorted(v.items()):
                    vs = [x[0] for x in vals]
                    out.append("%s %s %s launches %d avg %.4g max %.4g avg_dur_us %.1f" % (name, k, pn, len(vs), sum(vs) / len(vs), max(vs), sum(x[1] for x in vals) / len(vals) / 1e3))
                    tot[k][pn] = (sum(vs), len(vs))
        out += ["```", "", "## HBM traffic per macroblock (bytes; units = macroblocks the kernel processes per step, from the bench line)\n",
                "| kernel | macroblocks/step | FETCH_SIZE raw | WRITE_SIZE | raw total | corrected (2 x FETCH + WRITE) | algorithmic (SURVEY 8d) | corrected / algorithmic |", "|---|---|---|---|---|---|---|---|"]
        alg = {"recon_inter": 1648, "recon_split": 1648, "recon_intra": 1264, "loopfilter": 768, "parse_tokens": 880, "parse_headers": 80}
        t = {}
        for k in alg:
            if k not in tot or "FETCH_SIZE" not in tot[k] or "WRITE_SIZE" not in tot[k] or not units.get(k):
                continue
            if k in ("parse_tokens", "parse_headers") and fetch_bench.get("macroblocks_parsed_whole_run"):
                # worker grids do not come one per step: everything the run's grids moved over everything the run parsed
                f = tot[k]["FETCH_SIZE"][0] * 1024 / fetch_bench["macroblocks_parsed_whole_run"]
                w = tot[k]["WRITE_SIZE"][0] * 1024 / write_bench["macroblocks_parsed_whole_run"]
            else:
                steps_f = tot[k]["FETCH_SIZE"][1] / lps[k]; steps_w = tot[k]["WRITE_SIZE"][1] / lps[k]
                f = tot[k]["FETCH_SIZE"][0] * 1024 / (steps_f * units[k]); w = tot[k]["WRITE_SIZE"][0] * 1024 / (steps_w * units[k])
            t[k] = round(2 * f + w, 1)
            out.append("| %s | %d | %.0f | %.0f | %.0f | %.0f | %d | %.2f |" % (k, units[k], f, w, f + w, 2 * f + w, alg[k], (2 * f + w) / alg[k]))
        if t:
            traffic[cfg] = t
        open(os.path.join(ROOT, "profiles", "%s_%s.md" % (tag, cfg)), "w").write("\n".join(out) + "\n")
        print("wrote profiles/%s_%s.md" % (tag, cfg))
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
