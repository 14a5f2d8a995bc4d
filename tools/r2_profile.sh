#!/bin/bash
# dev helper, runs on the GPU box: rocprofv3 kernel stats + PMC passes of bench.py, summarised into gpurun_out/r2prof
# (raw traces are deleted: gpurun merges at most 64 MiB back).  usage: tools/r2_profile.sh COMMIT
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2prof; C=${1:-unknown}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc --output-format csv -- python3 $R/bench.py --steps 40 --warmup 11 --no-cpu-baseline > /dev/null 2> $O/pmc_$c.err
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace -d $O/pmc_SQ1 -o pmc --output-format csv -- python3 $R/bench.py --steps 40 --warmup 11 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_SQ2 -o pmc --output-format csv -- python3 $R/bench.py --steps 40 --warmup 11 --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 tools/summarize_pmc.py $O $C $O/pmc_FETCH_SIZE/pmc_counter_collection.csv $O/pmc_WRITE_SIZE/pmc_counter_collection.csv $O/pmc_SQ1/pmc_counter_collection.csv $O/pmc_SQ2/pmc_counter_collection.csv > $O/summary.log 2>&1
cp $O/stats/bench_kernel_stats.csv $O/r2_bench_kernel_stats.csv
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ1 $O/pmc_SQ2
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err
ls -la $O
