"""dev helper: cProfile of the per-frame python path."""
import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0"); cfg["mapping"]["first_iters"] = 20
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 71
start = int(sys.argv[2]) if len(sys.argv) > 2 else 21
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
for i in range(1, start): pipe.step(i, frames[i])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
for i in range(start, nf): pipe.step(i, frames[i])
host = time.time() - t0
torch.cuda.synchronize()
pr.disable()
print("fps", (nf - start) / (time.time() - t0), "host ms/frame", host / (nf - start) * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:9000])
