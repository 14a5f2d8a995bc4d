"""dev helper: per-block durations of the LDS scatter sweep (needs a -DSCATTER_PROF build: python tools/build_variant.py sprof -DSCATTER_PROF).
usage: RFX_LIB_PATH=build/variants/librfx_sprof.so python tools/scatter_prof.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd import _lib as L
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
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
ws = torch.empty(int(lib.rfx_grid_encode_backward_workspace_bytes(n, 16)) // 4, device="cuda")
for _ in range(3):
    L.check(lib.rfx_grid_encode_backward(enc.desc, L.ptr(enc.params), L.ptr(xx), n, L.ptr(df), L.ptr(dt), None, L.ptr(ws), ws.numel() * 4, st), "b")
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 8192))()
assert raw.rfx_debug_scatter_prof(buf, 2 * 8192) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 2)
sizes = list(enc.desc.size)[:16]
# round 6: one-dimensional grid, blocks in dispatch order (dearest level first; RFX_DEBUG_SWEEP=1 prints the parts per class)
nb = int((a[:, 1] > 0).sum())
t0 = a[:nb, 0].min()
start = (a[:nb, 0] - t0) / 100.0                        # 100 MHz wall clock -> us
dur = (a[:nb, 1] - a[:nb, 0]) / 100.0
end = start + dur
print("blocks", nb, "launch span", float(end.max()), "us; sum of block times", float(dur.sum()), "us =", float(dur.sum() / 256), "per CU")
for lo in range(0, nb, 16):
    m = slice(lo, min(nb, lo + 16))
    print(f"blocks {lo:4d}..: duration mean {dur[m].mean():6.1f} max {dur[m].max():6.1f}; start mean {start[m].mean():6.1f} max {start[m].max():6.1f}")
print("blocks running at t =", " ".join(f"{t}us:{int(((start <= t) & (end > t)).sum())}" for t in range(0, 130, 10)))
