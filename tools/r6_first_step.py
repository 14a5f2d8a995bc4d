"""dev helper (GPU box): host profile of the FIRST mapper step of a pipeline (frame 6) against the second (frame 11), after a
throwaway pipeline has warmed the process like bench.py does: what the first step pays once."""
import sys, os, time, cProfile, pstats, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0")
w = copy.deepcopy(cfg); w["mapping"]["first_iters"] = 4
wp = MappingPipeline(w, n_frames=20, seed=1000)
wf = wp.prefetch(list(range(12))); wp.start(wf[0])
for i in range(1, 12): wp.step(i, wf[i])
torch.cuda.synchronize(); del wp, wf
import gc; gc.collect()
pipe = MappingPipeline(cfg, n_frames=40)
frames = pipe.prefetch(list(range(26)))
pipe.start(frames[0])
for i in range(1, 6): pipe.step(i, frames[i])
torch.cuda.synchronize()
for f in (6, 11):
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); pipe.step(f, frames[f]); pr.disable()
    print(f"frame {f}: host {1e3 * (time.perf_counter() - t0):.3f} ms")
    if f == 6:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    for i in range(f + 1, f + 5): pipe.step(i, frames[i])
    torch.cuda.synchronize()
