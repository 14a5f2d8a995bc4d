#!/bin/bash
# dev helper (GPU box): subtractive timing builds of the sweep (results wrong on purpose): 1 = no flush, 3 = flush by plain stores, 2 = LDS atomics replaced by stores
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in new dbg1 dbg3 dbg2 new; do
  if [ $w = new ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
  ONLY16=1 timeout -k 10 300 python3 $R/tools/time_scatter_real.py office0 2>/dev/null | grep "both" | sed "s/^/$w office0 /"
done
