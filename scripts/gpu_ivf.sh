#!/bin/bash
# Quick check of the IVF paths on the GPU box: the IVF tests, then the bench's IVF legs.   scripts/gpu_ivf.sh <tag> [pytest -k expr]
set -o pipefail
tag=${1:-ivf}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ -n "$2" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_ivf.py -m gpu -x -q -k "$2" > $out/pytest.log 2>&1
else
  timeout -k 10 900 python -m pytest tests/test_gpu_ivf.py -m gpu -x -q > $out/pytest.log 2>&1
fi
echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
