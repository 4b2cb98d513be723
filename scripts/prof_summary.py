#!/usr/bin/env python3
"""Kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output) as CSV, one row per (kernel, grid):
launches of one kernel with different grids -- e.g. the 32-batch and the single-batch IVF list scan -- are separate rows.

  python scripts/prof_summary.py <results.db> [out.csv]
"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, grid_x, grid_y, workgroup_x, count(*), sum(end - start) / 1000.0, avg(end - start) / 1000.0, "
        "min(end - start) / 1000.0, max(end - start) / 1000.0, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name, grid_x, grid_y, workgroup_x order by 6 desc").fetchall()
    total = sum(r[5] for r in rows) or 1.0
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "grid_x_threads", "grid_y", "workgroup", "calls", "total_us", "avg_us", "min_us", "max_us", "percent",
                "vgpr", "agpr", "sgpr", "lds_bytes"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], r[4], f"{r[5]:.3f}", f"{r[6]:.3f}", f"{r[7]:.3f}", f"{r[8]:.3f}", f"{100 * r[5] / total:.3f}",
                    r[9], r[10], r[11], r[12]])


if __name__ == "__main__":
    main()
