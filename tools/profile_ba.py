import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0"); cfg["mapping"]["first_iters"] = 20
nf = 41
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
for i in range(1, 37): pipe.step(i, frames[i])
torch.cuda.synchronize()
mp = pipe.mapper
batch = pipe.dataset[35]
batch = {k: (v[None, ...] if isinstance(v, torch.Tensor) else torch.tensor([v])) for k, v in batch.items()}
for name, fn in (("global_mapping", mp.global_mapping), ("global_pose", mp.global_pose)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fn(batch, 35)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(name, "cpu ms/iter", (t1 - t0) / 15 * 1e3, "wall ms/iter", (t2 - t0) / 15 * 1e3)
import torch.autograd.profiler as prof
with prof.profile(use_device="cuda") as p:
    mp.global_pose(batch, 35)
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60)[:9000])
