#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE / SQ passes of tools/pmc_v1.py -> gpurun_out/r3m/r3_pmc_v1.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; T=/tmp/pv1
mkdir -p $O $T; cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $T/p$i -o pmc --output-format csv -- python3 $R/tools/pmc_v1.py > $T/p$i.log 2> $T/p$i.err || { tail -5 $T/p$i.err; exit 1; }
done
python3 - $T/p1/pmc_counter_collection.csv $T/p2/pmc_counter_collection.csv $T/p3/pmc_counter_collection.csv > $O/r3_pmc_v1.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sys.argv[1:]:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "rfx::mv_" in k:
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        last = v[-5:]          # the five launches on the bench's frame (the earlier ones filled the volume)
        print(f"   {c:24s} n={len(v):3d} last5 avg {sum(last)/len(last):16.1f}   all avg {sum(v)/len(v):16.1f}")
PY
cat $O/r3_pmc_v1.txt
