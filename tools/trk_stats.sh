#!/bin/bash
# dev helper, runs on the GPU box: rocprofv3 kernel stats of tools/tracker_times.py -> gpurun_out/trk_stats.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=/tmp/trk_stats
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 $R/tools/tracker_times.py > $R/gpurun_out/trk_stats_run.txt 2> $O/err.log
python3 - $O/t_kernel_stats.csv > $R/gpurun_out/trk_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
PY
cat $R/gpurun_out/trk_stats_run.txt
