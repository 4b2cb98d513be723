#!/bin/bash
# One gpurun call that produces everything profiles/ is built from (run from the repo root on the GPU box):
#   scripts/gpu_profile_round.sh <tag>
# -> gpurun_out/<tag>/{pytest.log, bench_default.json, bench_driver.json, bench_stats.json, kernel_stats.csv,
#    pmc_traffic.csv, sq_counters.txt}.  The rocpd databases are summarised here and deleted (they exceed what gpurun
#    carries back).
set -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -3 $out/pytest.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 600 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "bench driver-flags rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/stats -o s -- python3 bench.py --steps 320 --warmup 32 --no-cpu --repeats 3 > $out/bench_stats.json 2> $out/stats.err; echo "stats rc=$?"
python3 scripts/prof_summary.py "$(db $out/stats)" $out/kernel_stats.csv; rm -rf $out/stats
PB="python3 bench.py --steps 64 --warmup 32 --no-cpu --no-extras --no-prof --repeats 1"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o f -- $PB > /dev/null 2> $out/fetch.err; echo "fetch rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o w -- $PB > /dev/null 2> $out/write.err; echo "write rc=$?"
python3 scripts/pmc_summary.py "$(db $out/fetch)" "$(db $out/write)" $out/pmc_traffic.csv; rm -rf $out/fetch $out/write
# the UFIXED_POINT_8 runner on its own (bench.py runs it among the extras, which the PMC passes above skip)
timeout -k 10 200 python3 scripts/q8_bench.py > $out/q8_bench.txt 2> $out/q8_bench.err; echo "q8 bench rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/q8f -o f -- python3 scripts/q8_bench.py > /dev/null 2> $out/q8f.err; echo "q8 fetch rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/q8w -o w -- python3 scripts/q8_bench.py > /dev/null 2> $out/q8w.err; echo "q8 write rc=$?"
python3 scripts/pmc_summary.py "$(db $out/q8f)" "$(db $out/q8w)" $out/q8_pmc_traffic.csv; rm -rf $out/q8f $out/q8w
SB="python3 bench.py --steps 64 --warmup 32 --no-cpu --no-extras --no-prof --repeats 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $out/sq1 -o a -- $SB > /dev/null 2> $out/sq1.err; echo "sq1 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $out/sq2 -o b -- $SB > /dev/null 2> $out/sq2.err; echo "sq2 rc=$?"
python3 scripts/pmc_counters.py "$(db $out/sq1)" "$(db $out/sq2)" > $out/sq_counters.txt; rm -rf $out/sq1 $out/sq2
# what the fp32 MFMA pipe delivers in the scan's issue pattern without the scan's other work
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_ceiling scripts/microbench/mfma_f32_ceiling.hip 2> $out/mfma_ceiling.err && timeout -k 5 60 /tmp/mfma_f32_ceiling > $out/mfma_ceiling.txt; echo "mfma ceiling rc=$?"
grep "^\[bench\]" $out/bench_default.err
