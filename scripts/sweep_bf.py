#!/usr/bin/env python3
"""Batch-size sweep of the brute-force CLI with a CSV in the column layout of the reference's sweep
(qidk_bruteforce/scripts/run_all.sh:55, :87-94): one `vsearch_bf` run (qidk form) per dataset and batch size,
metrics.txt parsed into

    dataset,batch_size,throughput_qps,gflops,avg_latency_ms,p95_latency_ms,p99_latency_ms

  python scripts/sweep_bf.py [--datasets siftsmall sift] [--batches 1 8 16 32 64] [--data-root .] [--top-k 5]
         [--out results] [--gpus N]

Datasets follow the reference's layout: <data-root>/<name>/<name>_base.fvecs and <name>_query.fvecs.  A missing
dataset is skipped with a message (the reference's run_all.sh would fail in its build step).  Needs an MI355X:
there is no CPU fallback.
"""
from __future__ import annotations

import argparse
import csv
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COLUMNS = ["dataset", "batch_size", "throughput_qps", "gflops", "avg_latency_ms", "p95_latency_ms", "p99_latency_ms"]

# the lines run_all.sh greps out of metrics.txt (:87-91)
PATTERNS = {
    "throughput_qps": r"Throughput:\s*([0-9.eE+-]+)",
    "gflops": r"Avg GFLOPS:\s*([0-9.eE+-]+)",
    "avg_latency_ms": r"Avg graph execute time:\s*([0-9.eE+-]+)",
    "p95_latency_ms": r"P95 graph exec time:\s*([0-9.eE+-]+)",
    "p99_latency_ms": r"P99 graph exec time:\s*([0-9.eE+-]+)",
}


def parse_metrics(text: str) -> dict:
    out = {}
    for key, pat in PATTERNS.items():
        m = re.search(pat, text)
        out[key] = m.group(1) if m else ""
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--datasets", nargs="+", default=["siftsmall", "sift"])
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 8, 16, 32, 64])
    ap.add_argument("--data-root", default=".")
    ap.add_argument("--top-k", type=int, default=5)
    ap.add_argument("--out", default="results")
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()

    cli = os.path.join(ROOT, "hai-25-rag-on-edge_amd", "vsearch_bf")
    if not os.path.exists(cli):
        raise SystemExit(f"{cli} not built: python -c 'import __graft_entry__ as g; g.build()'")
    os.makedirs(args.out, exist_ok=True)
    csv_path = os.path.join(args.out, f"benchmark_{time.strftime('%Y%m%d_%H%M%S')}.csv")
    with open(csv_path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(COLUMNS)
        for ds in args.datasets:
            base = os.path.join(args.data_root, ds, f"{ds}_base.fvecs")
            query = os.path.join(args.data_root, ds, f"{ds}_query.fvecs")
            if not (os.path.exists(base) and os.path.exists(query)):
                print(f"dataset {ds}: {base} / {query} not found, skipped")
                continue
            for batch in args.batches:
                suffix = "" if batch == 1 else f"_b{batch}"  # run_all.sh:75-79
                rdir = os.path.join(args.out, f"{ds}{suffix}")
                cmd = [cli, "-", query, rdir, "-", base, str(args.top_k), str(batch)]
                if args.gpus > 1:
                    cmd += ["--gpus", str(args.gpus)]
                print(">>>", " ".join(cmd), flush=True)
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
                with open(os.path.join(rdir, "metrics.txt")) as mf:
                    m = parse_metrics(mf.read())
                w.writerow([ds, batch] + [m[c] for c in COLUMNS[2:]])
                fh.flush()
                print(f"  {ds} batch {batch}: {m['throughput_qps']} QPS, {m['gflops']} GFLOPS, "
                      f"avg {m['avg_latency_ms']} ms / batch")
    print("CSV:", csv_path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
