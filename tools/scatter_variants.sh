#!/bin/bash
# dev helper (GPU box): tools/time_scatter_real.py CONFIG for several library variants (first argument: the config)
R=$GRAFT_REPO_ROOT; CFG=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$v.so; fi
  echo "== $v: $(ONLY16=1 timeout -k 10 200 python3 $R/tools/time_scatter_real.py $CFG 2>&1 | grep 'levels' | sed 's/levels 0..15://' | tr '\n' '|')"
done
