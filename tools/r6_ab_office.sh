#!/bin/bash
# dev helper (GPU box): office0 scatter time (tools/time_scatter_real.py, all 16 levels) and bench.py at the driver's settings, in-tree library against build/variants/librfx_$1.so
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6; mkdir -p $O; v=$1
for w in $v new $v new; do
  if [ $w = new ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
  ONLY16=1 timeout -k 10 300 python3 $R/tools/time_scatter_real.py office0 2>/dev/null | grep "both" | sed "s/^/$w /"
  timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --render-frames 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w bench', d['value'], d['ms_per_step'], d['kernels'].get('rfx_field_backward_scatter_merged'))"
done
