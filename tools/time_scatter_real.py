"""dev helper: per-level cost of the LDS hash scatter on the sample points of a real BA iteration (captured from the
stage-by-stage issue) and on its TV lattice."""
import sys, os, time
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
print(name, "ray samples", x.shape[0], "TV points", pts.shape[0])
lib = L.load(); enc = pipe.model.embed_res_fn; st = L.stream_ptr(x.device)
g = torch.Generator(device="cuda").manual_seed(0)
def run(xx, k, reps=10):
    n = xx.shape[0]
    df = torch.randn((n, 2 * k), device="cuda", generator=g)
    dt = torch.zeros_like(enc.params)
    import ctypes as C
    nb = lib.rfx_grid_encode_backward_workspace_bytes(n, 16) if os.environ.get("MIN_WS") else lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(enc.desc), n)
    ws = torch.empty(int(nb) // 4, device="cuda")
    desc = type(enc.desc).from_buffer_copy(enc.desc); desc.n_levels = k
    call = lambda: lib.rfx_grid_encode_backward(desc, L.ptr(enc.params), L.ptr(xx), n, L.ptr(df), L.ptr(dt), None, L.ptr(ws), ws.numel() * 4, st)
    for _ in range(3): L.check(call(), "b")
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
print("sizes", list(enc.desc.size)[:16])
for label, xx in (("ray samples", x), ("TV lattice", pts), ("both", torch.cat([x, pts]))):
    prev = 0.0
    for k in ((16,) if os.environ.get("ONLY16") else (1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16)):
        a = run(xx, k)
        print(f"{label:12s} levels 0..{k-1:2d}: {a:7.1f} us (+{a - prev:6.1f})")
        prev = a
