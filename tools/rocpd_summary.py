#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / avg / min / max
duration and launch geometry -- the same table `--stats` prints, kept as text under profiles/."""
import sqlite3
import sys


def summarise(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), avg(grid_x), avg(grid_y), "
        "max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count), max(scratch_size) from kernels group by name "
        "order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    # launches of one kernel overlap (worker grids live side by side, row kernels of consecutive calls do not): the UNION of a
    # kernel's intervals is the wall time during which at least one launch of it was on the GPU
    union = {}
    for name, in db.execute("select distinct name from kernels"):
        iv = sorted(db.execute("select start, end from kernels where name = ?", (name,)).fetchall())
        tot_u, cur_s, cur_e = 0, None, None
        for a, b in iv:
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    tot_u += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        if cur_e is not None:
            tot_u += cur_e - cur_s
        union[name] = tot_u
    out = ["| kernel | calls | total ms (sum) | union ms | avg us | min us | max us | % of sum | avg grid (threads x,y) | wg | LDS B | VGPR | SGPR | scratch |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| %s | %d | %.3f | %.3f | %.3f | %.3f | %.3f | %.1f | %.0f x %.0f | %d | %d | %d | %d | %d |" % (
            r[0].replace("aa::(anonymous namespace)::", ""), r[1], r[2] / 1e6, union.get(r[0], 0) / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    return "\n".join(out)


if __name__ == "__main__":
    print(summarise(sys.argv[1]))
