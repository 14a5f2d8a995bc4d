#!/bin/bash
# runs on the GPU box: the round's bench lines, rocprofv3 kernel statistics, PMC traffic passes of the bench and of V1 on the
# bench's frame -> gpurun_out/r6m/ (copied into profiles/ afterwards).  usage: bash tools/r6_measure.sh COMMIT
set -o pipefail
R=$GRAFT_REPO_ROOT; C=${1:-HEAD}; O=$R/gpurun_out/r6m; T=/tmp/r6m
mkdir -p $O $T; cd /tmp; export TMPDIR=/tmp
echo "[r6m] bench, driver settings"; timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 > $O/r6_bench_driver_settings.json 2> $O/bench1.err || exit 1
echo "[r6m] bench, 100 steps"; timeout -k 10 300 python3 $R/bench.py --steps 100 --warmup 20 --no-side-configs > $O/r6_bench_final.json 2> $O/bench2.err || exit 1
echo "[r6m] kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $T/ks -o bench --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs > $O/r6_bench_under_rocprof.json 2> $T/ks.err || { tail -5 $T/ks.err; exit 1; }
cp $T/ks/bench_kernel_stats.csv $O/r6_bench_kernel_stats.csv
python3 - $O/r6_bench_kernel_stats.csv > $O/r6_bench_kernel_stats_rfx.md <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rfx::" in r["Name"]]
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs (librfx kernels; the driver's settings)\n")
print("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
PY
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  i=$((i+1)); echo "[r6m] pmc pass $i: $set"
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $T/p$i -o pmc --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs > $T/p$i.json 2> $T/p$i.err || { tail -5 $T/p$i.err; exit 1; }
done
cd $R
PMC_PREFIX=r6 PMC_COMMAND="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs" python3 tools/summarize_pmc.py $O $C $T/p1/pmc_counter_collection.csv $T/p2/pmc_counter_collection.csv $T/p3/pmc_counter_collection.csv $T/p4/pmc_counter_collection.csv > $O/summ.log 2>&1 || { tail $O/summ.log; exit 1; }
# ---- V1 on the bench's frame (frame 25 at the driver's settings), with the two calibration sweeps
cd /tmp
j=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY"; do
  j=$((j+1)); echo "[r6m] V1 pmc pass $j: $set"
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $T/v$j -o pmc --output-format csv -- python3 $R/tools/pmc_v1.py > $T/v$j.log 2> $T/v$j.err || { tail -5 $T/v$j.err; exit 1; }
done
cd $R
python3 - $C $O $T/v1/pmc_counter_collection.csv $T/v2/pmc_counter_collection.csv $T/v3/pmc_counter_collection.csv <<'PY'
import csv, sys, collections, json, os
sys.path.insert(0, os.getcwd())
from remixfusion_amd.build import sources_digest
commit, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sys.argv[3:]:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "rfx::mv_" in k:
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines, kern, calib = [], {}, {}
for k, d in acc.items():
    lines.append(k)
    for c, v in sorted(d.items()):
        last = v[-5:]          # the five launches on the bench's frame (the earlier ones filled the volume)
        lines.append(f"   {c:24s} n={len(v):3d} last5 avg {sum(last)/len(last):16.1f}   all avg {sum(v)/len(v):16.1f}")
    f = d.get("FETCH_SIZE", [0.0])[-5:]; w = d.get("WRITE_SIZE", [0.0])[-5:]
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    if "filter" in k or "copy" in k:
        calib[k] = {"FETCH_SIZE_KiB": fa, "WRITE_SIZE_KiB": wa}
    elif any(t in k for t in ("mv_chunks", "mv_rows", "mv_frame")):
        kern[k] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa, "hbm_bytes_raw": (fa + wa) * 1024, "hbm_bytes": (2 * fa + wa) * 1024}
open(os.path.join(out, "r6_pmc_v1.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"measured_at_commit": commit, "kernel_sources_digest": sources_digest(), "frame": 25, "config": "office0",
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/pmc_v1.py (tools/r6_measure.sh; averages of five launches on frame 25, "
                      "the frame bench.py's tsdf_integrate roofline counts U and C on at the driver's --steps 20 --warmup 5)",
           "calibration": {"kernels": calib, "known_bytes": {"mv_filter (one dword per lane, coalesced)": 1536000000, "mv_copy (read and written)": 4608000000},
                           "note": "FETCH_SIZE = TCC_EA0_RDREQ x 64 B and every L2 miss is one request for a whole 128-byte line, also for masked row-misaligned dword loads (profiles/r5_fetch_calib.txt: tools/micro/fetch_calib.hip): x2 = the bytes the L2s pulled in, Infinity-Cache hits included; WRITE_SIZE is exact"},
           "kernels": kern}, open(os.path.join(out, "r6_pmc_v1_frame25.json"), "w"), indent=1, sort_keys=True)
print("\n".join(lines))
PY
# ---- BASELINE configs 3-5 at their one-GPU sizes: bench line (with the binned scatter's roofline), kernel statistics, FETCH / WRITE passes
for cfg in scene0000 cafeteria apartment; do
  echo "[r6m] $cfg: bench"; cd /tmp
  timeout -k 10 300 python3 $R/bench.py --config $cfg --no-cpu-baseline --no-side-configs > $O/r6_bench_$cfg.json 2> $O/bench_$cfg.err || { tail -5 $O/bench_$cfg.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $T/ks_$cfg -o bench --output-format csv -- python3 $R/bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-side-configs --render-frames 1 > /dev/null 2> $T/ks_$cfg.err || { tail -5 $T/ks_$cfg.err; exit 1; }
  ( echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-side-configs --render-frames 1 (librfx kernels above 0.4 % of GPU time; round-6 kernels)"; python3 $R/tools/ks_top.py $T/ks_$cfg/bench_kernel_stats.csv 0.4 ) > $O/r6_kernel_stats_$cfg.txt
  k=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    k=$((k+1)); echo "[r6m] $cfg pmc pass $k: $set"
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $T/q${cfg}$k -o pmc --output-format csv -- python3 $R/bench.py --config $cfg --steps 20 --warmup 10 --no-cpu-baseline --no-side-configs --render-frames 0 > /dev/null 2> $T/q${cfg}$k.err || { tail -5 $T/q${cfg}$k.err; exit 1; }
  done
  cd $R
  mkdir -p $T/s_$cfg
  PMC_PREFIX=r6_$cfg PMC_COMMAND="python3 bench.py --config $cfg --steps 20 --warmup 10 --no-cpu-baseline --no-side-configs --render-frames 0" python3 tools/summarize_pmc.py $T/s_$cfg $C $T/q${cfg}1/pmc_counter_collection.csv $T/q${cfg}2/pmc_counter_collection.csv > $O/summ_$cfg.log 2>&1 || { tail $O/summ_$cfg.log; exit 1; }
  cp $T/s_$cfg/r6_${cfg}_pmc_traffic.json $O/r6_pmc_traffic_$cfg.json
done
echo "[r6m] done"; python3 - <<PY
import json
for f in ("r6_bench_driver_settings.json", "r6_bench_final.json", "r6_bench_scene0000.json", "r6_bench_cafeteria.json", "r6_bench_apartment.json"):
    d = json.loads(open("$O/" + f).read().strip().split("\n")[-1]); print(f, d["value"], d["ms_per_step"], d["render_rays_per_s"], d["roofline"]["frac"], {k: (v["frac"], v.get("frac_alone")) for k, v in d["rooflines"].items()})
PY
