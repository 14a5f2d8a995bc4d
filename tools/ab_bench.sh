#!/bin/bash
# dev helper (GPU box): bench.py alternately with the in-tree library and build/variants/librfx_$1.so, $2 rounds
R=$GRAFT_REPO_ROOT; V=$1; N=${2:-2}
for i in $(seq $N); do
  for w in tree $V; do
    if [ $w = tree ]; then unset RFX_LIB_PATH; else export RFX_LIB_PATH=$R/build/variants/librfx_$V.so; fi
    timeout -k 10 200 python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/ab_${w}_$i.json 2>/dev/null || exit 1
    echo "$w $i: $(python3 -c "import json;d=json.loads(open('$R/gpurun_out/ab_${w}_$i.json').read().strip().splitlines()[-1]);print(d['value'])")"
  done
done
