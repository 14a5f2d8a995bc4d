"""dev helper: the binned scatter ONE level at a time on the points of a real BA iteration (ray samples + TV lattice, the merged
call's two sources): HIP-event time per level.  usage: [LEVELS=4,6,8] python tools/r6_bin_one_level.py [config]
(under rocprofv3 --kernel-trace --stats with one level in LEVELS: the split between the kernels of that level)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd import _lib as L
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "cafeteria"
cfg = synthetic_config(name); cfg["mapping"]["first_iters"] = 20
pipe = MappingPipeline(cfg, n_frames=40)
frames = pipe.prefetch(list(range(32)))
pipe.start(frames[0])
for i in range(1, 26): pipe.step(i, frames[i])
d = pipe.mapper._direct_iterations(); d.stagewise_every = 1
for i in range(26, 32): pipe.step(i, frames[i])
torch.cuda.synchronize()
B = [v for k, v in d._cache.items() if k[0] == "stage"][0]
src = os.environ.get("SRC", "both")           # both | rays | lattice
x = (B.t.x01.clone() if src == "rays" else B.t.pts.clone() if src == "lattice" else torch.cat([B.t.x01.clone(), B.t.pts.clone()])).contiguous()
n = x.shape[0]
lib = L.load(); enc = pipe.model.embed_res_fn; st = L.stream_ptr(x.device)
g = torch.Generator(device="cuda").manual_seed(0)
sizes = list(enc.desc.size)[:16]
binned = [l for l in range(16) if -(-sizes[l] // 8192) >= 12]
sel = os.environ.get("LEVELS", "")
levels = [] if sel == "all" else [int(v) for v in sel.split(",")] if sel else binned
print(name, "source", src, "points", n, "(ray samples", B.t.x01.shape[0], "+ lattice", B.t.pts.shape[0], ")")
df = torch.randn((n, 2), device="cuda", generator=g)
dt = torch.zeros_like(enc.params)
for l in levels:
    desc = type(enc.desc).from_buffer_copy(enc.desc)
    desc.n_levels = 1
    for f in ("scale", "res", "size", "offset", "hashed"):
        getattr(desc, f)[0] = getattr(enc.desc, f)[l]
    nb = int(lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(desc), n))
    ws = torch.empty(nb // 4, device="cuda")
    call = lambda: lib.rfx_grid_encode_backward(desc, L.ptr(enc.params), L.ptr(x), n, L.ptr(df), L.ptr(dt), None, L.ptr(ws), ws.numel() * 4, st)
    for _ in range(3): L.check(call(), "b")
    evs = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    t = float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
    print(f"level {l:2d} res {enc.desc.res[l]:5d} size {sizes[l]:8d} segs {-(-sizes[l] // 8192):4d} {'hashed' if enc.desc.hashed[l] else 'dense '}: {t:7.1f} us", flush=True)

if sel in ("", "all"):          # all binned levels as ONE sub-grid: one group of launches
    k = len(binned)
    desc = type(enc.desc).from_buffer_copy(enc.desc)
    desc.n_levels = k
    for i, l in enumerate(binned):
        for f in ("scale", "res", "size", "offset", "hashed"):
            getattr(desc, f)[i] = getattr(enc.desc, f)[l]
    dfk = torch.randn((n, 2 * k), device="cuda", generator=g)
    nb = int(lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(desc), n))
    ws = torch.empty(nb // 4, device="cuda")
    call = lambda: lib.rfx_grid_encode_backward(desc, L.ptr(enc.params), L.ptr(x), n, L.ptr(dfk), L.ptr(dt), None, L.ptr(ws), ws.numel() * 4, st)
    for _ in range(3): L.check(call(), "b")
    evs = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    t = float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
    print(f"all {k} binned levels in one group: {t:7.1f} us = {t / k:6.1f} us per level", flush=True)
