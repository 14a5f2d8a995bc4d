"""dev helper: fused renderer + field forward timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.model.scene_rep import JointEncoding
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
cfg = synthetic_config(name)
m = JointEncoding(cfg, torch.from_numpy(np.array(cfg["mapping"]["bound"])), num_kf=8).cuda()
with torch.no_grad():
    m.embed_res_fn.params.uniform_(-0.5, 0.5); m.GBV.params.uniform_(0, 1)
H, W = cfg["cam"]["H"], cfg["cam"]["W"]
n = H * W
g = torch.Generator(device="cuda").manual_seed(0)
jj, ii = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
d = torch.stack([torch.ones(n, device="cuda"), ((ii.reshape(-1) - W / 2) / (0.9 * W)), ((jj.reshape(-1) - H / 2) / (0.9 * W))], -1).float().contiguous()
o = torch.tensor([0.0, -0.7, 0.2], device="cuda").repeat(n, 1)
td = (torch.rand(n, 1, device="cuda", generator=g) * 0.3 + 2.0)
def T(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
S = cfg["training"]["n_range_d"] + cfg["training"]["n_samples_d"]
ms = T(lambda: m.render_fused(o, d, td, jitter=False))
print(f"{name}: render {ms:.3f} ms  {n / ms / 1e3:.1f} Mrays/s  {n * S * 10624 / ms / 1e9:.1f} TFLOP/s ({n * S * 10624 / ms / 1e9 / 157.3 * 100:.1f}% of fp32 MFMA peak)")
x = torch.rand(1 << 21, 3, device="cuda", generator=g)
ms = T(lambda: m.query_color_sdf(x))
print(f"field_forward random pts: {ms:.3f} ms {x.shape[0] * 10624 / ms / 1e9:.1f} TFLOP/s")
