#!/bin/bash
# The sliced (cluster-sharded) IVF pipeline on virtual ranks: its tests, the per-rank cost table, a kernel trace of the same.
#   scripts/gpu_shard.sh <tag>
set -o pipefail
tag=${1:-shard}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_ivf.py -m gpu -x -q -k "sliced or shards or super_batches" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout -k 10 600 python3 scripts/ivf_shard_bench.py > $out/shard.txt 2> $out/shard.err; echo "bench rc=$?"; grep -v "^{" $out/shard.txt
timeout -k 10 600 rocprofv3 --kernel-trace -d $out/trace -o t -- python3 scripts/ivf_shard_bench.py > /dev/null 2> $out/trace.err || exit 1
db=$(find $out/trace -name '*.db' | head -1); python3 scripts/prof_summary.py $db $out/stats.csv 2>> $out/trace.err; rm -rf $out/trace; grep "ivf_\|merge" $out/stats.csv | cut -c1-200
