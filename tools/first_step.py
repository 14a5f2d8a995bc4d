"""dev helper: where the first mapper step of a pipeline spends its time (after a throwaway warm-up pipeline, like bench.py)."""
import sys, os, copy, gc, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0")
wcfg = copy.deepcopy(cfg); wcfg["mapping"]["first_iters"] = 4
wp = MappingPipeline(wcfg, n_frames=20, seed=1000)
wf = wp.prefetch(list(range(12))); wp.start(wf[0])
for i in range(1, 12): wp.step(i, wf[i])
torch.cuda.synchronize(); del wp, wf; gc.collect()
pipe = MappingPipeline(cfg, n_frames=140)
probe = [int(a) for a in sys.argv[1:]] or [6, 11]
frames = pipe.prefetch(list(range(max(probe) + 6)))
pipe.start(frames[0], first_iters=20)
for i in range(1, 6): pipe.step(i, frames[i])
torch.cuda.synchronize()
done = 6
for i in probe:
    for j in range(done, i):
        pipe.step(j, frames[j])
    done = i + 5
    for j in range(i, i + 5):
        pr = cProfile.Profile()
        torch.cuda.synchronize(); t0 = time.time()
        if j == i: pr.enable()
        pipe.step(j, frames[j])
        if j == i: pr.disable()
        host = time.time() - t0
        torch.cuda.synchronize(); tot = time.time() - t0
        if j == i:
            print(f"frame {j}: host {host * 1e3:.2f} ms, total {tot * 1e3:.2f} ms")
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
