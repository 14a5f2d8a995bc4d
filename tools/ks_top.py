"""dev helper: the librfx rows of a rocprofv3 --stats kernel csv above a share of GPU time.  usage: ks_top.py CSV [MIN_PCT]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
for r in rows:
    if "rfx::" in r["Name"] and float(r["Percentage"]) > thr:
        print(f"{r['Name'][:72]:72s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:8.2f} pct {r['Percentage']}")
