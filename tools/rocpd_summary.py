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
    out = ["| kernel | calls | total ms | avg us | min us | max us | % | avg grid (threads x,y) | wg | LDS B | VGPR | SGPR | scratch |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| %s | %d | %.3f | %.3f | %.3f | %.3f | %.1f | %.0f x %.0f | %d | %d | %d | %d | %d |" % (
            r[0].replace("aa::(anonymous namespace)::", ""), r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    return "\n".join(out)


if __name__ == "__main__":
    print(summarise(sys.argv[1]))
