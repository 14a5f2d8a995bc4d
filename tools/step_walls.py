"""dev helper: host wall time of every pipeline.step (tracker on, scene0000 sizes) from frame 1 on -- where do one-off stalls fall?"""
import sys, os, warnings, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
random.seed(0)
cfg = synthetic_config("scene0000"); cfg["synthetic"].update({"tracker": os.environ.get("TRACKER", "1") == "1", "depth_noise": 0.0, "dropout": 0.0, "clutter": 48})
cfg["mapping"]["first_iters"] = 50
nf = 41
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe = MappingPipeline(cfg, n_frames=nf + 8)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
torch.cuda.synchronize()
walls = []
import cProfile, pstats
slow = None
for i in range(1, nf):
    t0 = time.perf_counter()
    if i == int(os.environ.get("PROFILE_FRAME", -1)):
        pr = cProfile.Profile(); pr.enable(); pipe.step(i, frames[i]); pr.disable(); slow = pr
    else:
        pipe.step(i, frames[i])
    walls.append(round((time.perf_counter() - t0) * 1e3, 1))
torch.cuda.synchronize()
print("step host wall ms, frames 1..:", walls)
if slow is not None:
    pstats.Stats(slow).sort_stats("cumulative").print_stats(18)
