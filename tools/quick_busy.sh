#!/bin/bash
# dev helper, runs on the GPU box: rocpd kernel trace of bench.py, GPU busy fraction over the last 40 ms -> gpurun_out/busy_$1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=/tmp/qb_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o bench --output-format rocpd -- python3 $R/bench.py --steps 80 --warmup 20 --no-cpu-baseline > $R/gpurun_out/busy_$T.json 2> $O/err.log
cd $R
python3 tools/gpu_busy.py $(ls $O/*.db | head -1) 40 > gpurun_out/busy_$T.txt 2>&1
cat gpurun_out/busy_$T.txt
