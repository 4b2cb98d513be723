#!/usr/bin/env python3
"""nprobe sweep of the IVF CLI with a CSV in the column layout of the reference's sweep
(qidk_ivf/scripts/run_all_ivf.sh:62, :118-128): one `vsearch_ivf` run per nprobe, metrics.txt parsed into

    dataset,nprobe,top_k,recall,qps,avg_latency_ms,p50_latency_ms,p95_latency_ms,p99_latency_ms,avg_candidates,candidate_reduction

  python scripts/sweep_ivf.py --dataset sift --base sift_base.fvecs --queries sift_query.fvecs \
         [--groundtruth sift_groundtruth.ivecs] [--index-dir models/ivf_sift] [--nlist 1024] \
         [--nprobes 1 2 4 8 16 32 64] [--top-k 5] [--batch 32] [--out results]

If the index directory has no ivf_config.json it is built first with the GPU k-means builder (vs_ivf_build;
KMeans parameters of create_ivf_model_reordered.py:97-103: max_iter 100, random_state 42) and written in the
reference's own format.  Needs an MI355X: there is no CPU fallback.
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
sys.path.insert(0, ROOT)

COLUMNS = ["dataset", "nprobe", "top_k", "recall", "qps", "avg_latency_ms", "p50_latency_ms", "p95_latency_ms",
           "p99_latency_ms", "avg_candidates", "candidate_reduction"]

PATTERNS = {
    "recall": r"Recall@\d+:\s*([0-9.eE+-]+)%",
    "qps": r"QPS:\s*([0-9.eE+-]+)",
    "avg_latency_ms": r"Avg per query \(amortized\):\s*([0-9.eE+-]+)",
    "p50_latency_ms": r"P50:\s*([0-9.eE+-]+)",
    "p95_latency_ms": r"P95:\s*([0-9.eE+-]+)",
    "p99_latency_ms": r"P99:\s*([0-9.eE+-]+)",
    "avg_candidates": r"Avg candidates searched:\s*([0-9.eE+-]+)",
    "candidate_reduction": r"Candidate reduction:\s*([0-9.eE+-]+)x",
}


def parse_metrics(text: str) -> dict:
    out = {}
    for key, pat in PATTERNS.items():
        m = re.search(pat, text)
        out[key] = m.group(1) if m else ""
    return out


def build_index(base_fvecs: str, index_dir: str, nlist: int, max_iter: int) -> None:
    import __graft_entry__ as ge
    pkg = ge.load_package()
    base = pkg.read_fvecs(base_fvecs)
    nlist = pkg.clamp_nlist(base.shape[0], nlist)  # create_ivf_model_reordered.py:92-94
    vr, off, r2o, cents, iters = pkg.ivf_build(base, nlist, max_iter=max_iter, seed=42)
    print(f"k-means: {iters} iterations, nlist {nlist}")
    with pkg.IVFIndex(vectors_reordered=vr, centroids=cents, cluster_offsets=off, reorder_to_original=r2o) as ivf:
        ivf.save(index_dir)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="sift")
    ap.add_argument("--base", help="base .fvecs (only needed when the index has to be built)")
    ap.add_argument("--queries", required=True)
    ap.add_argument("--groundtruth", default="")
    ap.add_argument("--index-dir", default="")
    ap.add_argument("--nlist", type=int, default=1024)
    ap.add_argument("--max-iter", type=int, default=100)
    ap.add_argument("--nprobes", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32, 64])
    ap.add_argument("--top-k", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default="results")
    args = ap.parse_args()

    cli = os.path.join(ROOT, "hai-25-rag-on-edge_amd", "vsearch_ivf")
    if not os.path.exists(cli):
        raise SystemExit(f"{cli} not built: python -c 'import __graft_entry__ as g; g.build()'")
    index_dir = args.index_dir or os.path.join("models", f"ivf_{args.dataset}")
    if not os.path.exists(os.path.join(index_dir, "ivf_config.json")):
        if not args.base:
            raise SystemExit(f"no index in {index_dir} and no --base to build one from")
        os.makedirs(index_dir, exist_ok=True)
        build_index(args.base, index_dir, args.nlist, args.max_iter)

    os.makedirs(args.out, exist_ok=True)
    csv_path = os.path.join(args.out, f"ivf_benchmark_{time.strftime('%Y%m%d_%H%M%S')}.csv")
    with open(csv_path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(COLUMNS)
        for nprobe in args.nprobes:
            rdir = os.path.join(args.out, f"ivf_{args.dataset}_np{nprobe}")
            cmd = [cli, index_dir, args.queries, rdir, "-", str(args.top_k), str(nprobe), args.groundtruth, str(args.batch)]
            print(">>>", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
            with open(os.path.join(rdir, "metrics.txt")) as mf:
                m = parse_metrics(mf.read())
            w.writerow([args.dataset, nprobe, args.top_k] + [m[c] for c in COLUMNS[3:]])
            fh.flush()
            print(f"  nprobe {nprobe}: recall {m['recall']}%  QPS {m['qps']}  avg candidates {m['avg_candidates']}")
    print("CSV:", csv_path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
