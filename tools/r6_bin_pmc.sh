#!/bin/bash
# dev helper (GPU box): PMC counters of bin_sort / bin_reduce on the group of all binned levels (tools/r6_bin_one_level.py LEVELS=all);
# separate passes per counter set (FETCH_SIZE and WRITE_SIZE alone, as the guide prescribes), --kernel-trace only beside --pmc.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cfg=${1:-cafeteria}; O=/tmp/bpmc_$cfg; mkdir -p $O $R/gpurun_out/r6m; cd /tmp; export TMPDIR=/tmp LEVELS=all
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc --output-format csv -- python3 $R/tools/r6_bin_one_level.py $cfg > $O/p$i.out 2> $O/p$i.err || { tail -5 $O/p$i.err; exit 1; }
done
python3 - $O > $R/gpurun_out/r6m/r6_pmc_bin_$cfg.txt <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(sys.argv[1] + "/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "bin_sort_kernel" in k or "bin_reduce_kernel" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc <set> --kernel-trace -- python3 tools/r6_bin_one_level.py cafeteria (LEVELS=all): the LAST 10 dispatches of each kernel =")
print("# the timed calls on the group of all 12 binned levels (391 k points); FETCH_SIZE / WRITE_SIZE in KiB (FETCH x 2 on gfx950 for 128-B lines)")
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        last = v[-10:]
        print(f"   {c:24s} n={len(v):4d}  mean of last 10: {sum(last)/len(last):16.1f}")
PY
cat $R/gpurun_out/r6m/r6_pmc_bin_$cfg.txt
