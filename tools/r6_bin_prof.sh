#!/bin/bash
# dev helper (GPU box): event time per binned level + rocprofv3 kernel split of one dense and one hashed level; $1 config, $2 lib (new|r5|variant), $3 $4 levels
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cfg=$1; w=${2:-new}; O=/tmp/bp_${cfg}_$w; mkdir -p $O $R/gpurun_out/r6; cd /tmp; export TMPDIR=/tmp
if [ $w != new ]; then export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
out=$R/gpurun_out/r6/bin_${cfg}_$w.txt
timeout -k 10 200 python3 $R/tools/r6_bin_one_level.py $cfg > $out 2> $O/err0.log || { tail -5 $O/err0.log; exit 1; }
for lv in $3 $4; do
  export LEVELS=$lv
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/l$lv -o sp --output-format csv -- python3 $R/tools/r6_bin_one_level.py $cfg > $O/out$lv.txt 2> $O/err$lv.log || { tail -5 $O/err$lv.log; exit 1; }
  echo "-- kernels, level $lv only" >> $out
  python3 $R/tools/ks_last.py $O/l$lv/sp_kernel_trace.csv 10 bin_ >> $out
done
unset LEVELS
echo "== $cfg $w"; cat $out
