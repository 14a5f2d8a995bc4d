#!/bin/bash
# GPU box: FETCH_SIZE (and the raw TCC request counters, where the profiler knows them) of tools/micro/fetch_calib
# -> gpurun_out/fetch_calib.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=/tmp/fcal; mkdir -p $O $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
B=$R/tools/micro/fetch_calib
[ -x $B ] || { echo "build tools/micro/fetch_calib first"; exit 1; }
rocprofv3 -L > $R/gpurun_out/rocprof_counters.txt 2>&1 || true
$B > $R/gpurun_out/fetch_calib_plain.txt 2>&1 || { cat $R/gpurun_out/fetch_calib_plain.txt; exit 1; }
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_BUBBLE_sum" "TCC_READ_sum"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc --output-format csv -- $B > $O/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $O/p$i.log)"
done
python3 - $R/gpurun_out/fetch_calib_plain.txt $O/p*/pmc_counter_collection.csv > $R/gpurun_out/fetch_calib.txt <<'PY'
import csv, sys, collections
print(open(sys.argv[1]).read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sys.argv[2:]:
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v)} avg {sum(v)/len(v):16.1f}  last {v[-1]:16.1f}")
PY
cat $R/gpurun_out/fetch_calib.txt
