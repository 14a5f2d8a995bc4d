"""dev helper: the decoder's weight gradients and the hash scatter of one map iteration, back to back on one
stream against side by side on two streams (same buffers, captured from a stage-by-stage iteration of the office0 stream)."""
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
p, t = B.p, B.t
lib, model = L.load(), pipe.model
tr = cfg["training"]
S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
n = t.x01.shape[0] // S if t.x01.dim() == 2 else d._n_rays()
nS, n_tv = n * S, P ** 3
desc = model._field_desc(False); dref = C.byref(desc)
ws = model._workspace(lib.rfx_field_backward_workspace_bytes(nS), t.x01.device)
wsp, wb = ws.data_ptr(), ws.numel() * 4
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
def weights(st):
    L.check(lib.rfx_field_backward_weights(nS, p.d_raw, p.dws[0], p.dws[1], p.dws[2], p.dws[3], wsp, wb, st), "w")
def scatter(st):
    L.check(lib.rfx_field_backward_scatter_merged(dref, p.x01, nS, p.pts, p.dfeat, n_tv, p.dt, wsp, wb, p.ws2, B.ws2_bytes, st), "s")
def serial():
    weights(main.cuda_stream); scatter(main.cuda_stream)
def overlapped():
    e = torch.cuda.Event(); e.record(main); side.wait_event(e)
    weights(side.cuda_stream); scatter(main.cuda_stream)
    e2 = torch.cuda.Event(); e2.record(side); main.wait_event(e2)
def only(fn):
    return lambda: fn(main.cuda_stream)
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main); f(); b.record(main); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
print(name, "points", nS, "lattice", n_tv)
for rnd in range(2):
    print(f"weights alone {timeit(only(weights)):.1f} us; scatter alone {timeit(only(scatter)):.1f} us; back to back {timeit(serial):.1f} us; "
          f"two streams {timeit(overlapped):.1f} us")
