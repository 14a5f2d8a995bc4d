#!/bin/bash
# dev helper, runs on the GPU box: rocprofv3 kernel stats of tools/v1_probe.py -> gpurun_out/prof_v1_$1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; shift
O=/tmp/pv1_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O -o v1 --output-format csv -- python3 $R/tools/v1_probe.py "$@" > $R/gpurun_out/prof_v1_$T.log 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
python3 - $O/v1_kernel_stats.csv > $R/gpurun_out/prof_v1_$T.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} min_us {float(r['MinNs'])/1e3:8.2f} pct {r['Percentage']}")
PY
cp $O/v1_kernel_stats.csv $R/gpurun_out/prof_v1_${T}_kernel_stats.csv
tail -2 $R/gpurun_out/prof_v1_$T.log; cat $R/gpurun_out/prof_v1_$T.txt
