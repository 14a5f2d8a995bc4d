"""Summarise rocprofv3 --pmc passes into profiles/<PREFIX>_pmc_traffic.json and a per-kernel SQ counter table.

usage: [PMC_PREFIX=r3] python tools/summarize_pmc.py OUT_DIR COMMIT FETCH_CSV WRITE_CSV [SQ_CSV ...]

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950
tallies the 128-byte requests of wide coalesced reads at 64 B: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; the raw
sum is recorded next to it (gathers of 8/16-byte elements are closer to the raw figure)."""
import collections, csv, json, os, sys
out_dir, commit, fetch_csv, write_csv = sys.argv[1:5]
prefix = os.environ.get("PMC_PREFIX", "r3")
command = os.environ.get("PMC_COMMAND", "python3 bench.py --steps 40 --warmup 11 --no-cpu-baseline")
sq_csvs = sys.argv[5:]


def avg_by_kernel(path, want=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "rfx::" not in r["Kernel_Name"]:
            continue
        if want and r["Counter_Name"] not in want:
            continue
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


kern = {}
for tag, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
    for k, d in avg_by_kernel(path, {tag}).items():
        v = d[tag]
        kern.setdefault(k, {})[tag + "_KiB_avg"] = sum(v) / len(v)
        kern[k]["dispatches"] = len(v)
for k, v in kern.items():
    f, w = v.get("FETCH_SIZE_KiB_avg", 0.0), v.get("WRITE_SIZE_KiB_avg", 0.0)
    v["hbm_bytes_raw"] = (f + w) * 1024
    v["hbm_bytes"] = (2 * f + w) * 1024
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remixfusion_amd.build import sources_digest
json.dump({"measured_at_commit": commit, "kernel_sources_digest": sources_digest(), "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- {command} (one pass per counter)",
           "unit_note": "KiB counters; hbm_bytes = (2*FETCH + WRITE)*1024 (gfx950 FETCH correction), hbm_bytes_raw = (FETCH + WRITE)*1024; per launch averages",
           "kernels": kern}, open(os.path.join(out_dir, prefix + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
lines = [f"# HBM traffic per launch (rocprofv3 --pmc, commit {commit}); MiB", "", "| kernel | launches | FETCH raw | FETCH x2 | WRITE |", "|---|---|---|---|---|"]
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes"]):
    lines.append(f"| `{k}` | {v['dispatches']} | {v.get('FETCH_SIZE_KiB_avg', 0) / 1024:.2f} | {2 * v.get('FETCH_SIZE_KiB_avg', 0) / 1024:.2f} | {v.get('WRITE_SIZE_KiB_avg', 0) / 1024:.2f} |")
if sq_csvs:
    lines += ["", f"# SQ counters per launch (averages; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles)", ""]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in sq_csvs:
        for k, d in avg_by_kernel(p).items():
            for c, v in d.items():
                acc[k][c] += v
    names = sorted({c for d in acc.values() for c in d})
    lines += ["| kernel | " + " | ".join(names) + " |", "|---|" + "---|" * len(names)]
    for k, d in sorted(acc.items()):
        lines.append(f"| `{k}` | " + " | ".join(f"{sum(d[c]) / len(d[c]):.3g}" if d.get(c) else "" for c in names) + " |")
open(os.path.join(out_dir, prefix + "_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
