#!/bin/bash
# dev helper (GPU box): bench.py at the DRIVER's settings, in-tree library and build/variants/librfx_<name>.so ... in turn, $ROUNDS rounds
R=$GRAFT_REPO_ROOT; N=${ROUNDS:-3}
for i in $(seq $N); do
  for w in tree "$@"; do
    if [ $w = tree ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$w.so; fi
    timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --render-frames 0 > $R/gpurun_out/abv_${w}_$i.json 2>/dev/null || exit 1
    echo "$w $i: $(python3 -c "import json;d=json.loads(open('$R/gpurun_out/abv_${w}_$i.json').read().strip().splitlines()[-1]);k=d['kernels'].get('rfx_tsdf_integrate',{});print(d['value'], k.get('avg_ms'), k.get('median_ms'))")"
  done
done
