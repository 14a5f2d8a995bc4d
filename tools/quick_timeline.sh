#!/bin/bash
# dev helper, runs on the GPU box: rocpd kernel trace of tools/try_pipeline.py, steady-state timeline -> gpurun_out/tl_$1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=/tmp/tl_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o tl --output-format rocpd -- python3 $R/tools/try_pipeline.py office0 121 20 > $R/gpurun_out/tl_$T.log 2>&1
cd $R
python3 tools/timeline.py $(ls $O/*.db | head -1) 30 45 > gpurun_out/tl_$T.txt 2>&1
cat gpurun_out/tl_$T.txt
