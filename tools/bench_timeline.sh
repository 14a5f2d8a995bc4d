#!/bin/bash
# dev helper, runs on the GPU box: rocpd kernel trace of bench.py at the driver's settings -> gpurun_out/btl.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=/tmp/btl; S=${1:-20}; W=${2:-5}
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o tl --output-format rocpd -- python3 $R/bench.py --steps $S --warmup $W --no-cpu-baseline --no-side-configs --render-frames 0 --stage-events-every 0 ${CONFIG:+--config $CONFIG} > $R/gpurun_out/btl.json 2> $O/err.log
cd $R
python3 tools/bench_timeline.py $(ls $O/*.db | head -1) $S glue > gpurun_out/btl.txt 2>&1
cat gpurun_out/btl.txt
