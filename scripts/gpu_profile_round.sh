#!/bin/bash
# One gpurun call that produces everything profiles/ is built from (run from the repo root on the GPU box):
#   scripts/gpu_profile_round.sh <tag>
# -> gpurun_out/<tag>/{pytest.log,bench_default.json,bench_driver.json,stats/,fetch/,write/}
set -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
timeout -k 10 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "bench driver-flags rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/stats -o s -- python3 bench.py --steps 320 --warmup 32 --no-cpu --repeats 3 > $out/bench_stats.json 2> $out/stats.err; echo "stats rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o f -- python3 bench.py --steps 64 --warmup 32 --no-cpu --no-prof --repeats 1 > /dev/null 2> $out/fetch.err; echo "fetch rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o w -- python3 bench.py --steps 64 --warmup 32 --no-cpu --no-prof --repeats 1 > /dev/null 2> $out/write.err; echo "write rc=$?"
grep "^\[bench\]" $out/bench_default.err
