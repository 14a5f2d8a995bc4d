"""dev helper: per-frame tracker behaviour on scene0000 sizes (prediction error, result error, iteration successes)"""
import os, sys, random, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
from remixfusion_amd.model import ROtracker as RT
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
random.seed(0)
cfg = synthetic_config("scene0000"); cfg["synthetic"]["tracker"] = True
cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
cfg["mapping"]["first_iters"] = 30
print("RO:", cfg["RO"]); print("volume:", {k: cfg["volume"][k] for k in ("voxel_size", "trunc")})
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe = MappingPipeline(cfg, n_frames=N + 8)
ro = pipe.tracker.RO_Tracker
orig = ro.cal_transform
log = []
def spy(sv):
    ok, mt, tr = orig(sv)
    log.append((ok, float(sv[0]), float(mt), int((sv[1:] < sv[0]).sum()), ro.search_size.copy()))
    return ok, mt, tr
ro.cal_transform = spy
frames = pipe.prefetch(list(range(N)))
pipe.start(frames[0])
for i in range(1, N):
    log.clear()
    pred = None
    pipe.step(i, frames[i])
    est = pipe.slam.RO_c2w_data[i].cpu().double(); gt = frames[i]["c2w"].double()
    d = (est[:3, 3] - gt[:3, 3])
    dm = float((frames[i]["c2w"][:3, 3] - frames[i - 1]["c2w"][:3, 3]).norm())
    succ = sum(1 for l in log if l[0])
    print(f"frame {i:3d} motion {dm*100:5.2f} cm  err {float(d.norm())*100:6.2f} cm  (cam frame: {[round(float(v)*100,2) for v in gt[:3,:3].T @ d]})  iters ok {succ}/{len(log)}  "
          f"fit0 first {log[0][1]:.4f} last {log[-1][1]:.4f}  better-first {log[0][3]}  search last {np.round(log[-1][4][:3],4)}")
