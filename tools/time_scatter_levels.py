"""dev helper: cumulative cost of the E1-backward scatter over the first k levels, on a realistic TV lattice
and on ray-ordered samples (which levels are expensive?)."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd import _lib as L
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.model.scene_rep import JointEncoding
name = sys.argv[1] if len(sys.argv) > 1 else "scene0000"
cfg = synthetic_config(name)
tr = cfg["training"]
bb = np.array(cfg["mapping"]["bound"], dtype=np.float64)
m = JointEncoding(cfg, torch.from_numpy(bb), num_kf=8).cuda()
lib = L.load()
enc = m.embed_res_fn
P = tr["smooth_pts"] - 1
u6 = torch.rand(6, device="cuda")
pts = torch.empty((P ** 3, 3), device="cuda")
L.check(lib.rfx_tv_lattice(L.ptr(u6), P, float(tr["smooth_vox"]), float(tr["smooth_margin"]), m._bbox6, m._bbox_f64, 1, L.ptr(pts), L.stream_ptr(pts.device)), "lat")
print(name, "lattice", pts.shape[0], "range01", [(round(float(pts[:, d].min()), 3), round(float(pts[:, d].max()), 3)) for d in range(3)])
S = tr["n_range_d"] + tr["n_samples_d"]
n_rays = 2150
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.rand((n_rays, 1, 3), device="cuda", generator=g) * 0.2 + 0.4
d = torch.randn((n_rays, 1, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
t = torch.linspace(0.02, 0.45, S, device="cuda")[None, :, None]
rays = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3).contiguous()
st = L.stream_ptr(pts.device)
def run(x, k, use_ws, reps=5):
    n = x.shape[0]
    dfeat = torch.randn((n, 32), device="cuda", generator=g)
    dt = torch.zeros_like(enc.params)
    ws = torch.empty(int(lib.rfx_grid_encode_backward_workspace_bytes(n, 16)) // 4, device="cuda")
    desc = copy.copy(enc.desc)   # ctypes struct copy
    desc = type(enc.desc).from_buffer_copy(enc.desc)
    desc.n_levels = k
    def call():
        # dfeat rows stay 32 wide: pass ld through the level count of the *original* layout is not possible via this
        # entry point, so use k levels of a [n, 2k] slice instead
        return lib.rfx_grid_encode_backward(desc, L.ptr(enc.params), L.ptr(x), n, L.ptr(df), L.ptr(dt), None,
                                            L.ptr(ws) if use_ws else None, ws.numel() * 4 if use_ws else 0, st)
    df = dfeat[:, :2 * k].contiguous()
    for _ in range(2): L.check(call(), "b")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): L.check(call(), "b")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("sizes", list(enc.desc.size)[:16])
pts_in = pts.clone(); pts_in[:, 2] = pts_in[:, 2].clamp(0.0, 0.999)
pts_mod = pts.clone(); pts_mod[:, 2] = pts_mod[:, 2] - torch.floor(pts_mod[:, 2])
for label, x in (("TV lattice", pts), ("TV z clamped", pts_in), ("TV z mod 1", pts_mod)):
    prev_l = prev_d = 0.0
    for k in (4, 8, 12, 16):
        a, b = run(x, k, True), run(x, k, False)
        print(f"{label:12s} levels 0..{k-1:2d}: lds {a:7.3f} ms (+{a - prev_l:6.3f})   direct {b:7.3f} ms (+{b - prev_d:6.3f})")
        prev_l, prev_d = a, b
