#!/bin/bash
# dev helper, runs on the GPU box: rocprofv3 kernel trace of tools/v1_probe.py -> gpurun_out/prof_v1_$1.txt
# (per rfx::mv_* kernel: calls, mean, MEDIAN and min duration; the mean moves by +-4 us between runs, the median by < 1)
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; shift
O=/tmp/pv1_$T
mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace -d $O -o v1 --output-format csv -- python3 $R/tools/v1_probe.py "$@" > $R/gpurun_out/prof_v1_$T.log 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
python3 - $O/v1_kernel_trace.csv > $R/gpurun_out/prof_v1_$T.txt <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "rfx::" in k:
        d[k.split("(")[0].replace("void ", "")[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:60s} calls {len(v):4d} mean_us {statistics.mean(v):8.2f} median_us {statistics.median(v):8.2f} min_us {min(v):8.2f}")
    if "mv_" in k: tot += statistics.median(v)
print(f"sum of the medians of the rfx::mv_* kernels: {tot:.2f} us")
PY
tail -2 $R/gpurun_out/prof_v1_$T.log; cat $R/gpurun_out/prof_v1_$T.txt
