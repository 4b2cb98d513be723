#!/usr/bin/env python3
"""Per-kernel averages of every counter in one or more rocprofv3 --pmc runs (rocpd sqlite).
  python scripts/pmc_counters.py <a.db> [<b.db> ...] [--grep substring]"""
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
pat = None
if "--grep" in sys.argv:
    pat = sys.argv[sys.argv.index("--grep") + 1]
    args = [a for a in args if a != pat]
res = {}
for path in args:
    db = sqlite3.connect(path)
    q = ("select kernel_name, grid_size, counter_name, dispatch_id, sum(value) from counters_collection "
         "group by kernel_name, grid_size, counter_name, dispatch_id")
    for k, g, c, _, v in db.execute(q):
        if pat and pat not in k:
            continue
        a = res.setdefault((k, g), {}).setdefault(c, [0, 0.0])
        a[0] += 1
        a[1] += v
for (k, g), cs in sorted(res.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
    print(f"{k[:70]}  grid={g}")
    for c, (n, tot) in sorted(cs.items()):
        print(f"    {c:32s} launches={n:4d} avg={tot / n:16.1f}")
