#!/bin/bash
# dev helper (GPU box): what the live per-stage timing costs the headline.  bench.py at the driver's settings alternately with
#   events5   its default: HIP events inside the one-call iteration, every 5th iteration (rfx_ba_desc.stage_events, ABI 10)
#   events1   the same on every iteration
#   stagewise rounds 2-6's way: three iterations issued stage by stage (--stagewise-every 13)
#   none      no per-stage timing at all
R=$GRAFT_REPO_ROOT; N=${1:-3}
for i in $(seq $N); do
  for w in events5 events1 stagewise none; do
    case $w in
      events5) X="";;
      events1) X="--stage-events-every 1";;
      stagewise) X="--stagewise-every 13";;
      none) X="--stage-events-every 0";;
    esac
    timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --render-frames 0 $X > $R/gpurun_out/abs_${w}_$i.json 2>$R/gpurun_out/abs_${w}_$i.err || { tail -5 $R/gpurun_out/abs_${w}_$i.err; exit 1; }
    echo "$w $i: $(python3 -c "import json;d=json.loads(open('$R/gpurun_out/abs_${w}_$i.json').read().strip().splitlines()[-1]);k=d['kernels'].get('rfx_field_backward_scatter_merged',{});print(d['value'], k.get('calls_timed'), k.get('avg_ms'), k.get('median_ms'))")"
  done
done
