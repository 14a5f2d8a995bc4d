"""dev helper: torch profiler (CPU side) of mapper.step -- which ops cost host time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0"); cfg["mapping"]["first_iters"] = 20
nf = 61
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
for i in range(1, 31): pipe.step(i, frames[i])
torch.cuda.synchronize()
t0 = time.time()
for i in range(31, 51): pipe.step(i, frames[i])
torch.cuda.synchronize(); print("fps", 20 / (time.time() - t0))
import torch.autograd.profiler as prof
with prof.profile() as p:
    for i in range(51, 61): pipe.step(i, frames[i])
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60)[:12000])
