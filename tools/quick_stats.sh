#!/bin/bash
# dev helper, runs on the GPU box: rocprofv3 kernel stats of a short bench.py run, top kernels to gpurun_out/qs_$1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=/tmp/qs_$T; shift
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o bench --output-format csv -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-side-configs "$@" > $R/gpurun_out/qs_$T.json 2> $O/err.log
python3 - $O/bench_kernel_stats.csv > $R/gpurun_out/qs_$T.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
PY
cat $R/gpurun_out/qs_$T.json
