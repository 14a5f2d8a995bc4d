#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel durations of tools/v1_probe.py for several library variants, each run twice
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$v.so; fi
  bash $R/tools/prof_v1.sh $v > /dev/null 2>&1 || { echo "variant $v failed"; exit 1; }
  echo "== $v (run $rep): $(tail -1 $R/gpurun_out/prof_v1_$v.log)"
  grep "rfx::\|sum of" $R/gpurun_out/prof_v1_$v.txt | cut -c1-44,61-
done
done
