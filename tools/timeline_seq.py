"""dev helper: kernel-by-kernel sequence of one mapper step from a rocprofv3 rocpd sqlite trace (start offset, duration,
gap since the previous kernel on the same queue ended)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12; count = int(sys.argv[3]) if len(sys.argv) > 3 else 260
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
q = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else '0')
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.{q} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if 'gbv_integrate_kernel' in r[0]]          # one per mapper step
i0 = marks[skip]
t0 = rows[i0][1]
last_end = {}
for n, s, e, qu in rows[i0:i0 + count]:
    gap = (s - last_end[qu]) / 1e3 if qu in last_end else 0.0
    last_end[qu] = max(e, last_end.get(qu, 0))
    name = n.replace('void ', '').replace('rfx::', '')[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  q{qu!s:4s} dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {name}")
