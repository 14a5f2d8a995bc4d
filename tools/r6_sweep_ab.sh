#!/bin/bash
# dev helper (GPU box): the sweep's schedule, in-tree library against build/variants/librfx_$1.so (default: cls): gradient tests first, then
# the merged scatter of a real iteration's points (tools/time_scatter_real.py) at three table sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; V=${1:-cls}
timeout -k 10 600 python3 -m pytest tests/test_field_gpu.py tests/test_timed_path_gpu.py -x -q 2>&1 | tail -3 || exit 1
for w in $V new $V new; do
  if [ $w = new ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
  for c in office0 scene0000 cafeteria; do
    ONLY16=1 RFX_DEBUG_SWEEP=${DBG:-} timeout -k 10 300 python3 $R/tools/time_scatter_real.py $c 2>&1 | grep -E "both|sweep" | tail -2 | sed "s/^/$w $c /"
  done
done
