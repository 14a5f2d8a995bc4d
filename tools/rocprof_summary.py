"""dev helper: per-kernel summary of a rocprofv3 rocpd sqlite file."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 5 desc"
print(f"{'kernel':80s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'total_us':>10s}")
for r in list(c.execute(q))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r[0][:80]:80s} {r[1]:6d} {r[2]:9.1f} {r[3]:9.1f} {r[4]:10.1f}")
