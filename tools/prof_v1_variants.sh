#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel stats of tools/v1_probe.py for several library variants
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$v.so; fi
  bash $R/tools/prof_v1.sh $v > /dev/null 2>&1 || { echo "variant $v failed"; exit 1; }
  echo "== $v: $(tail -1 $R/gpurun_out/prof_v1_$v.log)"
  grep "rfx::" $R/gpurun_out/prof_v1_$v.txt | cut -c1-45,100-
done
