"""dev helper: time the E1 backward paths on realistic ray-major points (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd import _lib as L
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.model.scene_rep import JointEncoding
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
cfg = synthetic_config(name)
bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
m = JointEncoding(cfg, bb, num_kf=8).cuda()
lib = L.load()
enc = m.embed_res_fn
S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
n_rays = 2300
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.rand((n_rays, 1, 3), device="cuda", generator=g) * 0.2 + 0.4
d = torch.randn((n_rays, 1, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
t = torch.linspace(0.02, 0.45, S, device="cuda")[None, :, None]
x = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3).contiguous()
n = x.shape[0]
dfeat = torch.randn((n, 32), device="cuda", generator=g)
dt = torch.zeros_like(enc.params)
ws = torch.empty(int(lib.rfx_grid_encode_backward_workspace_bytes(n, 16)) // 4, device="cuda")
st = L.stream_ptr(x.device)
def run(use_ws, reps=20):
    for _ in range(3):
        L.check(lib.rfx_grid_encode_backward(enc.desc, L.ptr(enc.params), L.ptr(x), n, L.ptr(dfeat), L.ptr(dt), None,
                                             L.ptr(ws) if use_ws else None, ws.numel() * 4 if use_ws else 0, st), "b")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        L.check(lib.rfx_grid_encode_backward(enc.desc, L.ptr(enc.params), L.ptr(x), n, L.ptr(dfeat), L.ptr(dt), None,
                                             L.ptr(ws) if use_ws else None, ws.numel() * 4 if use_ws else 0, st), "b")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print(name, "points", n, "levels sizes", list(enc.desc.size)[:16])
print("direct atomics ms", run(False))
print("lds scatter    ms", run(True))
# TV lattice
P = cfg["training"]["smooth_pts"] - 1
lat = torch.stack(torch.meshgrid(*[torch.arange(P, device="cuda") * 0.004 + 0.3] * 3, indexing="ij"), -1).reshape(-1, 3).float().contiguous()
x, n = lat, lat.shape[0]
dfeat = torch.randn((n, 32), device="cuda", generator=g); ws = torch.empty(int(lib.rfx_grid_encode_backward_workspace_bytes(n, 16)) // 4, device="cuda")
print("TV lattice points", n, "direct", run(False), "lds", run(True))
# both point sets in one launch (what the merged scatter of the BA iteration sees)
xa = torch.cat([(o + d * t).clamp(0.001, 0.999).reshape(-1, 3), lat], 0).contiguous()
x, n = xa, xa.shape[0]
dfeat = torch.randn((n, 32), device="cuda", generator=g); ws = torch.empty(int(lib.rfx_grid_encode_backward_workspace_bytes(n, 16)) // 4, device="cuda")
print("concatenated points", n, "direct", run(False), "lds", run(True))
