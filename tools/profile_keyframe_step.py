"""dev helper: cProfile of ONE steady-state keyframe step (main thread) of a configuration: where the host time goes"""
import sys, os, cProfile, pstats, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "apartment"
cfg = synthetic_config(name); cfg["mapping"]["first_iters"] = 50
cfg["data"]["output"] = "/tmp/rfx_prof_out"
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe = MappingPipeline(cfg, n_frames=48)
frames = pipe.prefetch(list(range(40)))
pipe.start(frames[0])
ke = cfg["mapping"]["keyframe_every"]
for i in range(1, 30): pipe.step(i, frames[i])
pr = cProfile.Profile()
for i in range(30, 38):
    t0 = time.perf_counter()
    if i == 36:
        pr.enable(); pipe.step(i, frames[i]); pr.disable()
    else:
        pipe.step(i, frames[i])
    print(f"frame {i}: host {1e3 * (time.perf_counter() - t0):.2f} ms")
pipe.mapper.wait_meshes()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
