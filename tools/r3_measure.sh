#!/bin/bash
# runs on the GPU box: the round's bench lines, rocprofv3 kernel statistics and PMC traffic passes -> gpurun_out/r3m/
# (copied into profiles/ by hand afterwards).  usage: bash tools/r3_measure.sh COMMIT
set -o pipefail
R=$GRAFT_REPO_ROOT; C=${1:-HEAD}; O=$R/gpurun_out/r3m; T=/tmp/r3m
mkdir -p $O $T; cd /tmp; export TMPDIR=/tmp
echo "[r3m] bench, driver settings"; timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 > $O/r3_bench_driver_settings.json 2> $O/bench1.err || exit 1
echo "[r3m] bench, 100 steps"; timeout -k 10 300 python3 $R/bench.py --steps 100 --warmup 20 > $O/r3_bench_final.json 2> $O/bench2.err || exit 1
echo "[r3m] kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $T/ks -o bench --output-format csv -- python3 $R/bench.py --steps 40 --warmup 11 --no-cpu-baseline > $O/r3_bench_under_rocprof.json 2> $T/ks.err || { tail -5 $T/ks.err; exit 1; }
cp $T/ks/bench_kernel_stats.csv $O/r3_bench_kernel_stats.csv
python3 - $O/r3_bench_kernel_stats.csv > $O/r3_bench_kernel_stats_rfx.md <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rfx::" in r["Name"]]
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 40 --warmup 11 --no-cpu-baseline (librfx kernels)\n")
print("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
PY
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  i=$((i+1)); echo "[r3m] pmc pass $i: $set"
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $T/p$i -o pmc --output-format csv -- python3 $R/bench.py --steps 40 --warmup 11 --no-cpu-baseline > $T/p$i.json 2> $T/p$i.err || { tail -5 $T/p$i.err; exit 1; }
done
cd $R
PMC_PREFIX=r3 python3 tools/summarize_pmc.py $O $C $T/p1/pmc_counter_collection.csv $T/p2/pmc_counter_collection.csv $T/p3/pmc_counter_collection.csv $T/p4/pmc_counter_collection.csv > $O/summ.log 2>&1 || { tail $O/summ.log; exit 1; }
echo "[r3m] done"; python3 - <<PY
import json
for f in ("r3_bench_driver_settings.json", "r3_bench_final.json"):
    d = json.loads(open("$O/" + f).read().strip().split("\n")[-1]); print(f, d["value"], d["ms_per_step"], d["render_rays_per_s"], d["roofline"]["frac"], {k: (v["frac"], v.get("frac_alone")) for k, v in d["rooflines"].items()})
PY
