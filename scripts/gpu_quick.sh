#!/bin/bash
# Quick check on the GPU box: the GPU tests, the default bench, and a kernel trace of a short bench run summarised per
# (kernel, grid).   scripts/gpu_quick.sh <tag> [pytest args]
set -o pipefail
tag=${1:-q}
shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout -k 10 600 python3 bench.py --no-cpu > $out/b.json 2> $out/err || exit 1
grep "^\[bench\]" $out/err
timeout -k 10 600 rocprofv3 --kernel-trace -d $out/trace -o t -- python3 bench.py --steps 64 --warmup 32 --no-cpu --no-int8 --no-extras --repeats 1 > /dev/null 2> $out/trace.err || exit 1
db=$(find $out/trace -name '*.db' | head -1); python3 scripts/prof_summary.py $db $out/stats.csv 2>> $out/trace.err; rm -rf $out/trace; head -25 $out/stats.csv | cut -c1-150
