#!/bin/bash
# dev helper (GPU box): tools/time_scatter_real.py for several library variants
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$v.so; fi
  echo "== $v"
  ONLY16=1 timeout -k 10 200 python3 $R/tools/time_scatter_real.py office0 2>&1 | grep "levels\|ray samples" || exit 1
done
