"""dev helper: from a rocprofv3 --kernel-trace csv, the median duration of the LAST n dispatches of every kernel whose name
contains one of the patterns (the warm, timed calls at the end of a tool's run).  usage: ks_last.py TRACE_CSV N pattern..."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
n, pats = int(sys.argv[2]), sys.argv[3:]
by = {}
for r in rows:
    nm = r["Kernel_Name"]
    if any(p in nm for p in pats):
        by.setdefault(nm, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")))
for nm, v in by.items():
    v.sort()
    last = v[-n:]
    d = np.array([x[1] for x in last]) / 1e3
    print(f"{nm[:70]:70s} last {len(last):3d} of {len(v):5d}: median {np.median(d):8.2f} us  min {d.min():8.2f}  max {d.max():8.2f}  grid {last[-1][2]} wg {last[-1][3]}")
