#!/bin/bash
# One gpurun call that produces everything profiles/ is built from (run from the repo root on the GPU box):
#   scripts/gpu_profile_round.sh <tag>
# -> gpurun_out/<tag>/{pytest.log, bench_default.json, bench_driver.json, bench_stats.json, kernel_stats.csv,
#    pmc_traffic.csv, sq_counters.txt, b1_*, ivf_*, shard.txt}.  The rocpd databases are summarised here and deleted
#    (they exceed what gpurun carries back).
set -o pipefail
tag=${1:-r03}
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
# the single-call scan (B = 1) and the IVF list scan on the fp32 rows: their own kernel statistics and HBM traffic
for leg in "b1 scripts/b1_bench.py 1" "ivf_i8 scripts/ivf_bench.py 0" "ivf_f32 scripts/ivf_bench.py 1"; do
  set -- $leg; name=$1; shift
  timeout -k 10 300 python3 "$@" > $out/$name.txt 2> $out/$name.err; echo "$name rc=$?"; cat $out/$name.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${name}_s -o s -- python3 "$@" > /dev/null 2> $out/${name}_s.err
  python3 scripts/prof_summary.py "$(db $out/${name}_s)" $out/${name}_kernel_stats.csv; rm -rf $out/${name}_s
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${name}_f -o f -- python3 "$@" > /dev/null 2> $out/${name}_f.err
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${name}_w -o w -- python3 "$@" > /dev/null 2> $out/${name}_w.err
  python3 scripts/pmc_summary.py "$(db $out/${name}_f)" "$(db $out/${name}_w)" $out/${name}_pmc_traffic.csv; rm -rf $out/${name}_f $out/${name}_w
done
# what a rank of a cluster-sharded job does per launch group (virtual ranks)
timeout -k 10 400 python3 scripts/ivf_shard_bench.py > $out/shard.txt 2> $out/shard.err; echo "shard rc=$?"
# the UFIXED_POINT_8 runner on its own (bench.py runs it among the extras, which the PMC passes above skip)
timeout -k 10 200 python3 scripts/q8_bench.py > $out/q8_bench.txt 2> $out/q8_bench.err; echo "q8 bench rc=$?"
SB="python3 bench.py --steps 64 --warmup 32 --no-cpu --no-extras --no-prof --repeats 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $out/sq1 -o a -- $SB > /dev/null 2> $out/sq1.err; echo "sq1 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $out/sq2 -o b -- $SB > /dev/null 2> $out/sq2.err; echo "sq2 rc=$?"
python3 scripts/pmc_counters.py "$(db $out/sq1)" "$(db $out/sq2)" > $out/sq_counters.txt; rm -rf $out/sq1 $out/sq2
grep "^\[bench\]" $out/bench_default.err
