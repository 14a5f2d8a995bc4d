"""dev helper: per-block durations of the binned scatter's two kernels (needs a -DSCATTER_PROF build: python tools/build_variant.py sprof -DSCATTER_PROF).
usage: RFX_DEBUG_BINS=1 RFX_LIB_PATH=build/variants/librfx_sprof.so python tools/bin_prof.py cafeteria"""
import sys, os, ctypes as C
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
x, pts = B.t.x01.clone(), B.t.pts.clone()
lib = L.load(); enc = pipe.model.embed_res_fn; st = L.stream_ptr(x.device)
raw = C.CDLL(L.LIB_PATH)
g = torch.Generator(device="cuda").manual_seed(0)
xx = torch.cat([x, pts]); n = xx.shape[0]
df = torch.randn((n, 32), device="cuda", generator=g)
dt = torch.zeros_like(enc.params)
ws = torch.empty(int(lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(enc.desc), n)) // 4, device="cuda")
for _ in range(3):
    L.check(lib.rfx_grid_encode_backward(enc.desc, L.ptr(enc.params), L.ptr(xx), n, L.ptr(df), L.ptr(dt), None, L.ptr(ws), ws.numel() * 4, st), "b")
torch.cuda.synchronize()
sizes = list(enc.desc.size)[:16]
print(name, "points", n, "level sizes", sizes, "hashed", list(enc.desc.hashed)[:16])
for which, label in ((0, "bin_sort"), (1, "bin_reduce")):
    buf = (C.c_ulonglong * (2 * 8192))()
    assert raw.rfx_debug_bin_prof(which, buf, 2 * 8192) == 0
    a = np.array(buf[:], dtype=np.uint64).reshape(-1, 2)
    live = a[:, 1] > 0
    idx = np.nonzero(live)[0]
    t0 = a[idx, 0].min()
    start = (a[idx, 0] - t0) / 100.0
    dur = (a[idx, 1] - a[idx, 0]) / 100.0
    end = start + dur
    print(f"{label}: {len(idx)} blocks (ids up to {idx.max()}), span {end.max():.1f} us, sum of block times {dur.sum():.0f} us = {dur.sum() / 256:.1f} per CU")
    step = max(64, len(idx) // 24 // 64 * 64)
    for lo in range(0, len(idx), step):
        m = slice(lo, min(len(idx), lo + step))
        print(f"   blocks {idx[lo]:5d}..: duration mean {dur[m].mean():6.1f} max {dur[m].max():6.1f}; start mean {start[m].mean():6.1f} max {start[m].max():6.1f}; end max {end[m].max():6.1f}")
    print("   running at t =", " ".join(f"{t}:{int(((start <= t) & (end > t)).sum())}" for t in range(0, int(end.max()) + 20, 20)))
