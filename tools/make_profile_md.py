"""profiles/rN_bench_kernel_stats_rfx.md from the rocprofv3 --stats kernel csv (librfx kernels only).
usage: python tools/make_profile_md.py [SRC.csv [DST.md [ROUND]]]"""
import csv, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "profiles/r1_bench_kernel_stats.csv")
dst = sys.argv[2] if len(sys.argv) > 2 else src.replace(".csv", "_rfx.md")
rnd = sys.argv[3] if len(sys.argv) > 3 else "1"
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline (round " + rnd + "; the process also runs the warm-up pipeline, the 200 first-frame iterations and the closing full-frame renders)", "",
       "librfx kernels only (full table: " + os.path.basename(src) + "). Durations in microseconds.", "",
       f"All GPU kernel time in the run: {tot / 1e6:.1f} ms; librfx share: "
       f"{sum(float(r['TotalDurationNs']) for r in rows if 'rfx::' in r['Name']) / tot:.1%}.", "",
       "| kernel | calls | avg us | min us | max us | % of all GPU time |", "|---|---|---|---|---|---|"]
for r in rows:
    if "rfx::" not in r["Name"]:
        continue
    name = r["Name"].split("(")[0].replace("void ", "")
    out.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
               f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
