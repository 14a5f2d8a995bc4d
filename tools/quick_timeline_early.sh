#!/bin/bash
# dev helper, runs on the GPU box: timeline of the first 25 frames of the stream (the driver's bench window) -> gpurun_out/tle_$1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=/tmp/tle_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o tl --output-format rocpd -- python3 $R/tools/try_pipeline.py office0 26 200 > $R/gpurun_out/tle_$T.log 2>&1
cd $R
python3 tools/timeline.py $(ls $O/*.db | head -1) 5 60 > gpurun_out/tle_$T.txt 2>&1
cat gpurun_out/tle_$T.txt
