"""dev helper: GPU busy fraction over the last `window_ms` of a rocprofv3 rocpd trace."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); window = float(sys.argv[2]) * 1e6
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
tend = c.execute(f"select max(end) from {kd}").fetchone()[0]
t0 = tend - window
busy, n = c.execute(f"select sum(end-start), count(*) from {kd} where start >= ?", (t0,)).fetchone()
print(f"window {window/1e6:.1f} ms: {n} kernels, busy {busy/1e6:.1f} ms = {busy/window:.2%}")
q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where d.start >= ? group by s.kernel_name order by 3 desc"
rows = list(c.execute(q, (t0,)))
rfx = sum(r[2] for r in rows if 'rfx' in r[0]); other = sum(r[2] for r in rows if 'rfx' not in r[0])
print(f"librfx kernels {rfx/1e3:.1f} ms ({sum(r[1] for r in rows if 'rfx' in r[0])} launches), other (ATen/hipBLAS/copies) {other/1e3:.1f} ms ({sum(r[1] for r in rows if 'rfx' not in r[0])} launches)")
for r in rows[:14]: print(f"  {r[0][:90]:90s} {r[1]:6d} {r[2]/1e3:8.2f} ms")
