#!/bin/bash
# The reference's two benchmark sweeps (qidk_bruteforce/scripts/run_all.sh, qidk_ivf/scripts/run_all_ivf.sh) on synthetic
# files of the SIFT shapes, through the CLIs (host buffers, file output and all):   scripts/gpu_sweeps.sh <tag>
# -> gpurun_out/<tag>/sweep_bf.csv, sweep_ivf.csv
set -o pipefail
tag=${1:-sweeps}
out=$PWD/gpurun_out/$tag
data=/tmp/vs_sweep_data
mkdir -p $out $data/siftsmall $data/sift
export TMPDIR=/tmp
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
for name, n, nq in (("siftsmall", 10_000, 100), ("sift", 1_000_000, 10_000)):
    base = pkg.synth_sift(n, seed=20251205)
    q = pkg.synth_sift(nq, seed=20251206)
    pkg.write_fvecs(f"$data/{name}/{name}_base.fvecs", base)
    pkg.write_fvecs(f"$data/{name}/{name}_query.fvecs", q)
    with pkg.BruteForceIndex(base) as bf:
        ids, _ = bf.search(q, 15)    # ground truth in the TEXMEX .ivecs layout (15 neighbours per query: the exact API serves k <= 15)
    pkg.write_ivecs(f"$data/{name}/{name}_groundtruth.ivecs", ids)
    print(name, base.shape, q.shape, flush=True)
PY
[ $? -eq 0 ] || exit 1
timeout -k 10 900 python3 scripts/sweep_bf.py --data-root $data --out $out/bf > $out/sweep_bf.log 2>&1 || { tail -5 $out/sweep_bf.log; exit 1; }
cp $(ls -t $out/bf/*.csv | head -1) $out/sweep_bf.csv
timeout -k 10 900 python3 scripts/sweep_ivf.py --dataset sift --base $data/sift/sift_base.fvecs --queries $data/sift/sift_query.fvecs \
    --groundtruth $data/sift/sift_groundtruth.ivecs --index-dir $data/ivf_sift --max-iter 20 --out $out/ivf > $out/sweep_ivf.log 2>&1 || { tail -5 $out/sweep_ivf.log; exit 1; }
cp $(ls -t $out/ivf/*.csv | head -1) $out/sweep_ivf.csv
rm -rf $out/bf $out/ivf
cat $out/sweep_bf.csv $out/sweep_ivf.csv
