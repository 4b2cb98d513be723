#!/bin/bash
# Quick check of the single-call (B = 1) path on the GPU box: the parity tests that reach it, the micro-benchmark, and its
# kernel trace.   scripts/gpu_b1.sh <tag>
set -o pipefail
tag=${1:-b1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
timeout -k 10 300 python3 scripts/b1_bench.py 1 > $out/b1.txt 2> $out/b1.err; cat $out/b1.txt
timeout -k 10 300 python3 scripts/b1_bench.py 32 > $out/b32.txt 2> $out/b32.err; cat $out/b32.txt
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o t -- python3 scripts/b1_bench.py 1 > /dev/null 2> $out/trace.err || exit 1
db=$(find $out/trace -name '*.db' | head -1); python3 scripts/prof_summary.py $db $out/stats.csv 2>> $out/trace.err; rm -rf $out/trace; grep "scan_one\|merge" $out/stats.csv | cut -c1-150
