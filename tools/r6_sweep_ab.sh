#!/bin/bash
# dev helper (GPU box): the sweep's new schedule against build/variants/librfx_pre.so: gradient tests, block profile, scatter times, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python3 -m pytest tests/test_field_gpu.py tests/test_timed_path_gpu.py -x -q 2>&1 | tail -3 || exit 1
RFX_DEBUG_SWEEP=1 RFX_LIB_PATH=$R/build/variants/librfx_sprof.so timeout -k 10 300 python3 tools/scatter_prof.py 2>&1 | grep -v amdgpu.ids | tail -14
for w in pre new pre new; do
  if [ $w = new ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
  for c in office0 scene0000; do
    ONLY16=1 timeout -k 10 300 python3 $R/tools/time_scatter_real.py $c 2>/dev/null | grep "both" | sed "s/^/$w $c /"
  done
done
