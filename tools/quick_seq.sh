#!/bin/bash
# dev helper, runs on the GPU box: rocpd kernel trace of tools/try_pipeline.py, one mapper step kernel by kernel -> gpurun_out/seq_$1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=/tmp/seq_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o tl --output-format rocpd -- python3 $R/tools/try_pipeline.py office0 121 20 > $R/gpurun_out/seq_$T.log 2>&1 || exit 1
cd $R
python3 tools/timeline_seq.py $(ls $O/*.db | head -1) 12 260 > gpurun_out/seq_$T.txt 2>&1
python3 tools/timeline.py $(ls $O/*.db | head -1) 30 60 > gpurun_out/tl_$T.txt 2>&1
