"""dev helper: config 3 (mapping + ROTracker pose estimation) for a few frames."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("scene0000"); cfg["synthetic"]["tracker"] = True; cfg["mapping"]["first_iters"] = 50
pipe = MappingPipeline(cfg, n_frames=40)
frames = pipe.prefetch(list(range(26)))
pipe.start(frames[0])
torch.cuda.synchronize(); t0 = time.time()
for i in range(1, 26): pipe.step(i, frames[i])
torch.cuda.synchronize()
print(f"tracker pipeline: {25 / (time.time() - t0):.1f} fps, mv_stream {pipe.mv_stream}")
for i in (5, 15, 25):
    est = pipe.slam.est_c2w_data[i].cpu() if i % 5 == 0 else (pipe.slam.est_c2w_data_rel[i] @ pipe.slam.est_c2w_data[(i // 5) * 5]).cpu()
    print(i, "translation error mm", round(float((est[:3, 3] - frames[i]["c2w"][:3, 3]).norm()) * 1e3, 1))
