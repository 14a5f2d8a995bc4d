"""dev helper: timeline of bench.py's TIMED region from a rocprofv3 rocpd trace (bench.py ... --render-frames 0): the window from the
(steps)-th last mv_rows launch to the last kernel.  Prints busy time (union over streams), the large idle gaps and what ran around them."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
v1 = [i for i, r in enumerate(rows) if 'mv_frame_kernel' in r[0]]
i0 = v1[-steps]
win = rows[i0:]
t0, t1 = win[0][1], max(r[2] for r in win)
# union of busy intervals
busy, cur_e = 0, t0
gaps = []
for n, s, e in win:
    if s > cur_e:
        gaps.append((s - cur_e, cur_e, n))
        busy += e - s
        cur_e = e
    elif e > cur_e:
        busy += e - cur_e
        cur_e = e
print(f"window {(t1 - t0) / 1e6:.3f} ms, {len(win)} kernels, busy (union) {busy / 1e6:.3f} ms = {busy / (t1 - t0):.1%}")
edges = [0, 2e3, 5e3, 20e3, 100e3, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    g = [x[0] for x in gaps if lo <= x[0] < hi]
    print(f"  gaps {lo / 1e3:6.0f}-{hi / 1e3:8.0f} us: {len(g):6d}  total {sum(g) / 1e6:8.3f} ms")
for g, at, n in sorted(gaps, key=lambda x: -x[0])[:12]:
    prev = [r for r in win if r[2] <= at + 1][-1][0]
    print(f"  gap {g / 1e3:8.1f} us at {(at - t0) / 1e6:7.3f} ms: after {prev[:50]} -> before {n[:50]}")
agg = {}
for n, s, e in win:
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {k[:90]:90s} {v[0]:5d} x {v[1] / v[0] / 1e3:8.1f} us = {v[1] / 1e6:7.3f} ms")
# everything that is not a librfx kernel (torch glue of the frame loop and of the mapper step), by total time
other = {k: v for k, v in agg.items() if 'rfx::' not in k}
tot = sum(v[1] for v in other.values())
print(f"non-librfx kernels in the window: {sum(v[0] for v in other.values())} launches, {tot / 1e6:.3f} ms in total")
for k, v in sorted(other.items(), key=lambda kv: -kv[1][1])[:25]:
    short = k.replace('at::native::', '').replace('void ', '')[:110]
    print(f"  {short:110s} {v[0]:5d} x {v[1] / v[0] / 1e3:7.1f} us = {v[1] / 1e6:7.3f} ms")
# the launches of ONE mapper step's glue: from the end of a step's last rba kernel to the next step's first ba_prologue
if len(sys.argv) > 3:
    pro = [i for i, r in enumerate(win) if 'ba_prologue_kernel' in r[0]]
    # first prologue of the third mapper step in the window (prologues come in runs of 10)
    k = pro[20] if len(pro) > 20 else pro[-10]
    j = k - 1
    while j > 0 and 'adam_step_kernel' not in win[j][0]:
        j -= 1
    print(f"launches between the previous step's last optimizer step and this step's first iteration ({(win[k][1] - win[j][2]) / 1e3:.1f} us):")
    for n, s, e in win[j:k + 1]:
        print(f"   +{(s - win[j][1]) / 1e3:9.1f} us {(e - s) / 1e3:7.1f} us  {n.replace('at::native::', '')[:100]}")
