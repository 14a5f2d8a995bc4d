#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel stats of tools/time_scatter_real.py (16 levels) for one config; $2 = r5 | new
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cfg=$1; w=${2:-new}; O=/tmp/sp_${cfg}_$w; mkdir -p $O $R/gpurun_out/r6; cd /tmp; export TMPDIR=/tmp
if [ $w = r5 ]; then export RFX_LIB_PATH=$R/build/variants/librfx_r5.so; fi
if [ $w != r5 ] && [ $w != new ]; then export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
export ONLY16=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o sp --output-format csv -- python3 $R/tools/time_scatter_real.py $cfg > $O/out.txt 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
python3 $R/tools/ks_top.py $O/sp_kernel_stats.csv 0.3 | grep -i "bin_\|scatter" > $R/gpurun_out/r6/sp_${cfg}_$w.txt
echo "== $cfg $w"; cat $R/gpurun_out/r6/sp_${cfg}_$w.txt
