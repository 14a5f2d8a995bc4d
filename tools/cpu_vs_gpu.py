"""dev helper: is the frame loop CPU- or GPU-bound?  (time to issue the loop vs time until the GPU drains)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
cfg = synthetic_config(name); cfg["mapping"]["first_iters"] = 20
nf = 121
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
for i in range(1, 21): pipe.step(i, frames[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(21, nf): pipe.step(i, frames[i])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name}: issue {(t1 - t0) * 1e3 / (nf - 21):.3f} ms/frame, drained {(t2 - t0) * 1e3 / (nf - 21):.3f} ms/frame -> {(nf - 21) / (t2 - t0):.1f} fps; GPU backlog at loop end {(t2 - t1) * 1e3:.2f} ms")
