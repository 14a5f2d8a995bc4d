"""dev helper: CPU-side time of pieces of one mapping iteration (no GPU sync inside the pieces)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0"); cfg["mapping"]["first_iters"] = 20
pipe = MappingPipeline(cfg, n_frames=40)
frames = pipe.prefetch(list(range(12)))
pipe.start(frames[0])
for i in range(1, 12): pipe.step(i, frames[i])
m, slam, mp = pipe.model, pipe.slam, pipe.mapper
n = 2148
o = torch.rand(n, 3, device="cuda"); d = torch.rand(n, 3, device="cuda"); tgt = torch.rand(n, 3, device="cuda"); td = torch.rand(n, 1, device="cuda") * 2 + 0.5
m.train()
def T(f, reps=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3
print("mapping fwd        cpu/wall ms", T(lambda: m.mapping(o, d, tgt, td)))
ret = m.mapping(o, d, tgt, td)
print("get_loss (smooth)  ", T(lambda: slam.get_loss_from_ret(m.mapping(o, d, tgt, td), smooth=True)))
def fb():
    slam.map_optimizer.zero_grad()
    loss = slam.get_loss_from_ret(m.mapping(o, d, tgt, td), smooth=True)
    loss.backward()
print("fwd+loss+bwd       ", T(fb))
def fbs():
    fb(); slam.map_optimizer.step()
print("fwd+loss+bwd+adam  ", T(fbs))
og = o.clone().requires_grad_(True); dg = d.clone().requires_grad_(True)
def fbp():
    slam.map_optimizer.zero_grad()
    loss = slam.get_loss_from_ret(m.mapping(og, dg, tgt, td, clamp=True), smooth=True)
    loss.backward()
print("fwd+bwd with dx    ", T(fbp))
import torch.autograd.profiler as prof
with prof.profile() as p:
    for _ in range(5): fb()
print(p.key_averages().table(sort_by="self_cpu_time_total", row_limit=25)[:6000])
