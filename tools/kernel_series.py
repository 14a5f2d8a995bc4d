"""dev helper: durations of one kernel, call by call, from a rocpd trace: python tools/kernel_series.py trace.db name [name2 ...]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
for pat in sys.argv[2:]:
    d = [(e - s) / 1e3 for n, s, e in rows if pat in n]
    print(pat, len(d), "calls; us:", " ".join(f"{v:.0f}" for v in d))
