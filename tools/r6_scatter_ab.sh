#!/bin/bash
# dev helper (GPU box): per-level scatter cost, round-5 library (build/variants/librfx_r5.so) against the in-tree one
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O
for cfg in "$@"; do
  for w in r5 new; do
    if [ $w = r5 ]; then export RFX_LIB_PATH=$R/build/variants/librfx_r5.so; else unset RFX_LIB_PATH; fi
    ONLY16=1 timeout -k 10 300 python3 $R/tools/time_scatter_real.py $cfg > $O/scat_${cfg}_$w.txt 2> $O/scat_${cfg}_$w.err || { tail -5 $O/scat_${cfg}_$w.err; exit 1; }
    echo "== $cfg $w"; cat $O/scat_${cfg}_$w.txt
  done
done
