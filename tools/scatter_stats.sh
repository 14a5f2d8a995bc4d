#!/bin/bash
# dev helper, runs on the GPU box: rocprofv3 kernel stats of tools/time_scatter_real.py -> gpurun_out/ss_$1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=/tmp/ss_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
ONLY16=1 rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 $R/tools/time_scatter_real.py ${2:-office0} > $R/gpurun_out/ss_$T.log 2>&1 || exit 1
python3 - $O/s_kernel_stats.csv > $R/gpurun_out/ss_$T.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "scatter" in r["Name"] or "bin_" in r["Name"] or "fill" in r["Name"].lower():
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
cat $R/gpurun_out/ss_$T.txt
