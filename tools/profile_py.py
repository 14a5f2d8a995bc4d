"""dev helper: cProfile of the per-frame python path."""
import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0"); cfg["mapping"]["first_iters"] = 20
nf = 71
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
for i in range(1, 21): pipe.step(i, frames[i])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
for i in range(21, nf): pipe.step(i, frames[i])
torch.cuda.synchronize()
pr.disable()
print("fps", (nf - 21) / (time.time() - t0))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
