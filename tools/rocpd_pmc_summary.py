"""Counters per kernel of one rocprofv3 --pmc pass (rocpd sqlite): python tools/rocpd_pmc_summary.py <results.db> [kernel substring ...]"""
import collections
import sqlite3
import sys


def per_kernel(db, wanted):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith('rocpd_pmc_event')][0][len('rocpd_pmc_event'):]
    q = f"""select d.id, ks.kernel_name, p.name, sum(e.value), d.end-d.start from rocpd_pmc_event{suf} e
      join rocpd_info_pmc{suf} p on e.pmc_id=p.id
      join rocpd_kernel_dispatch{suf} d on e.event_id=d.event_id
      join rocpd_info_kernel_symbol{suf} ks on d.kernel_id=ks.id group by 1,3 order by 1"""
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for _did, kn, pn, v, dur in c.execute(q):
        k = [x for x in wanted if x in kn]
        if k:
            by[k[0]][pn].append((v, dur))
    return by


if __name__ == "__main__":
    wanted = sys.argv[2:] or ['k_loopfilter_rows4', 'k_recon_inter4', 'k_recon_inter(', 'k_recon_intra4', 'k_parse_tokens', 'k_parse_mb_headers']
    for k, v in sorted(per_kernel(sys.argv[1], wanted).items()):
        for pn, vals in sorted(v.items()):
            vs = [x[0] for x in vals]; ds = [x[1] for x in vals]
            print(k, pn, 'n', len(vs), 'avg %.5g' % (sum(vs) / len(vs)), 'max %.5g' % max(vs), 'avg_dur_us %.1f' % (sum(ds) / len(ds) / 1e3))
