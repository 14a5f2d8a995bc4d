"""dev helper: per-block phases of the binned scatter's reduce kernel (needs a -DBIN_PROF build: python tools/build_variant.py
bprof -DBIN_PROF).  usage: RFX_LIB_PATH=build/variants/librfx_bprof.so python tools/bin_prof.py [config]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd import _lib as L
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "cafeteria"
cfg = synthetic_config(name); cfg["mapping"]["first_iters"] = 5
pipe = MappingPipeline(cfg, n_frames=20)
frames = pipe.prefetch(list(range(12)))
pipe.start(frames[0])
for i in range(1, 7): pipe.step(i, frames[i])
d = pipe.mapper._direct_iterations(); d.stagewise_every = 1
for i in range(7, 12): pipe.step(i, frames[i])
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
buf = (C.c_ulonglong * (4 * 4096))()
assert raw.rfx_debug_bin_prof(buf, 4 * 4096) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 3] > 0]
t0 = a[:, 0].min()
z, w, f = (a[:, 1] - a[:, 0]) / 100.0, (a[:, 2] - a[:, 1]) / 100.0, (a[:, 3] - a[:, 2]) / 100.0
print(name, "points", n, "reduce blocks stamped", len(a))
print(f"per block (us): zero {z.mean():.1f}  add {w.mean():.1f}  flush {f.mean():.1f}  total {(z + w + f).mean():.1f}; span {(a[:, 3].max() - t0) / 100.0:.1f}")
start = (a[:, 0] - t0) / 100.0
print("blocks started by t =", " ".join(f"{t}us:{int((start <= t).sum())}" for t in range(0, 500, 50)))
