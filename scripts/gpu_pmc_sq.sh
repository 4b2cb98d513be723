#!/bin/bash
# SQ counters of the brute-force scans (fp32 streaming + wide int8), two passes of 8 counters.
#   scripts/gpu_pmc_sq.sh <tag>   -> gpurun_out/<tag>/sq1, sq2
tag=${1:-sq}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 64 --warmup 32 --no-cpu --no-ivf --no-extras --no-prof --repeats 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $out/sq1 -o a -- $B > /dev/null 2> $out/sq1.err; echo "sq1 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $out/sq2 -o b -- $B > /dev/null 2> $out/sq2.err; echo "sq2 rc=$?"
