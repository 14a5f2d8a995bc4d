#!/bin/bash
# dev helper, runs on the GPU box: per-call kernel durations over a bench.py run -> gpurun_out/series.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=/tmp/bser; S=${1:-100}; W=${2:-20}; shift; shift
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o tl --output-format rocpd -- python3 $R/bench.py --steps $S --warmup $W --no-cpu-baseline --render-frames 0 --first-iters 200 > $R/gpurun_out/series.json 2> $O/err.log
cd $R
python3 tools/kernel_series.py $(ls $O/*.db | head -1) "$@" > gpurun_out/series.txt 2>&1
cat gpurun_out/series.txt
