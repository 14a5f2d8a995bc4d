"""dev helper: steady-state timeline of the mapping loop from a rocprofv3 rocpd sqlite trace.

Window = from the `skip`-th mv_integrate launch to the last one (so first-frame mapping and the trailing render are
excluded).  Prints busy fraction, idle-gap histogram and per-frame kernel counts/time grouped by name."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); skip = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
v1 = [i for i, r in enumerate(rows) if 'mv_rows_kernel' in r[0]]
i0, i1 = v1[skip], v1[-1]
frames = len(v1) - 1 - skip
win = rows[i0:i1]
t0, t1 = win[0][1], rows[i1][1]
busy = sum(r[2] - r[1] for r in win)
print(f"window {(t1 - t0) / 1e6:.2f} ms over {frames} frames = {(t1 - t0) / 1e6 / frames:.3f} ms/frame; kernels {len(win)} "
      f"({len(win) / frames:.1f}/frame); busy {busy / (t1 - t0):.1%}")
gaps = [max(0, win[i + 1][1] - max(r[2] for r in win[max(0, i - 3):i + 1])) for i in range(len(win) - 1)]
edges = [0, 1e3, 3e3, 10e3, 30e3, 100e3, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    g = [x for x in gaps if lo <= x < hi]
    print(f"  gaps {lo / 1e3:6.0f}-{hi / 1e3:8.0f} us: {len(g):6d}  total {sum(g) / 1e6:8.2f} ms ({sum(g) / (t1 - t0):.1%})")
big = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:6]
for i in sorted(big):
    if gaps[i] > 200e3:
        print(f"  gap {gaps[i] / 1e3:7.0f} us at {(win[i][2] - t0) / 1e6:7.2f} ms: after {win[i][0][:60]} -> before {win[i + 1][0][:60]}")
agg = {}
for n, s, e in win:
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
rfx = sum(v[1] for k, v in agg.items() if 'rfx' in k)
print(f"librfx {rfx / 1e6 / frames:.3f} ms/frame ({sum(v[0] for k, v in agg.items() if 'rfx' in k) / frames:.1f} launches/frame); "
      f"other {(busy - rfx) / 1e6 / frames:.3f} ms/frame ({sum(v[0] for k, v in agg.items() if 'rfx' not in k) / frames:.1f} launches/frame)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"  {k[:100]:100s} {v[0] / frames:7.2f}/frame {v[1] / v[0] / 1e3:8.1f} us  {v[1] / 1e6 / frames:7.4f} ms/frame")
