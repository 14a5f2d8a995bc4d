#!/bin/bash
# dev helper (GPU box): BIN_CHUNK_RECORDS (records one bin_reduce block is meant to add: sets `parts`): merged scatter of a real iteration and
# the 40-frame stream of the larger configs, in-tree library (16 384) against variants
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for w in new chunk8 chunk32 chunk64 new; do
  if [ $w = new ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
  for c in scene0000 cafeteria; do
    a=$(ONLY16=1 timeout -k 10 300 python3 $R/tools/time_scatter_real.py $c 2>/dev/null | grep "both" | sed 's/.*: *//')
    b=$(timeout -k 10 300 python3 $R/bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline --no-side-configs --render-frames 0 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'], d['kernels']['rfx_field_backward_scatter_merged']['median_ms'])")
    echo "$w $c: scatter $a | bench $b"
  done
done
