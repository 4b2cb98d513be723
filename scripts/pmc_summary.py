#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc runs (rocpd sqlite): FETCH_SIZE in one, WRITE_SIZE in the other
(they do not fit one pass: MI355X_MICROARCH.md, "rocprofv3 PMC slots").

  python scripts/pmc_summary.py <fetch.db> <write.db> [out.csv]

Units and the gfx950 correction follow MI355X_MICROARCH.md "HBM": the counters are in KB, and FETCH_SIZE reports half of
the bytes of a wide coalesced read, so  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (calibrate on row_sqnorm_kernel,
which reads exactly rows * 512 bytes)."""
import csv
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "counter_name" if "counter_name" in cols else "name"
    kcol = "kernel_name" if "kernel_name" in cols else "kernel"
    q = (f"select {kcol}, grid_size, dispatch_id, sum(value) from counters_collection where {name_col} = ? "
         f"group by {kcol}, grid_size, dispatch_id")
    out = {}
    for k, grid, _, v in db.execute(q, (counter,)):
        a = out.setdefault((k, grid), [0, 0.0])
        a[0] += 1
        a[1] += v
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = open(sys.argv[3], "w", newline="") if len(sys.argv) > 3 else sys.stdout
    cw = csv.writer(out)
    cw.writerow(["kernel", "grid_size", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_bytes_per_launch(2*FETCH+WRITE)*1024"])
    rows = []
    for key in f:
        n, fs = f[key]
        wn, ws = w.get(key, (0, 0.0))
        fa, wa = fs / n, (ws / wn if wn else 0.0)
        rows.append((key[0], key[1], n, fa, wa, int((2 * fa + wa) * 1024)))
    for r in sorted(rows, key=lambda r: -r[5] * r[2]):
        cw.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.1f}", r[5]])


if __name__ == "__main__":
    main()
