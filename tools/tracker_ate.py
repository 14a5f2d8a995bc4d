"""dev helper: config 3 (scene0000 sizes, ROTracker on) over N frames: absolute trajectory error against the synthetic ground truth."""
import os, sys, time, random, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
random.seed(0)
cfg = synthetic_config("scene0000"); cfg["synthetic"]["tracker"] = True
cfg["synthetic"].update({"depth_noise": float(os.environ.get("NOISE", 0.0)), "dropout": 0.0, "clutter": int(os.environ.get("CLUTTER", 0))})
cfg["mapping"]["first_iters"] = 50
cfg["RO"]["device_search"] = os.environ.get("DEVICE_SEARCH", "1") == "1"
if os.environ.get("NO_POSE_OPT"): cfg["synthetic"]["pose_opt"] = False
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe = MappingPipeline(cfg, n_frames=N + 8)
print("PST source:", pipe.tracker.RO_Tracker.PST_source)
frames = pipe.prefetch(list(range(N)))
pipe.start(frames[0])
torch.cuda.synchronize(); t0 = time.time()
for i in range(1, N): pipe.step(i, frames[i])
torch.cuda.synchronize()
print(f"{(N - 1) / (time.time() - t0):.1f} frames/s")
ke = cfg["mapping"]["keyframe_every"]
err, along, rot = [], [], []
for i in range(1, N):
    est = pipe.slam.est_c2w_data[i] if i % ke == 0 else pipe.slam.est_c2w_data_rel[i] @ pipe.slam.est_c2w_data[(i // ke) * ke]
    est, gt = est.cpu().double(), frames[i]["c2w"].double()
    d = est[:3, 3] - gt[:3, 3]
    err.append(float(d.norm())); along.append(abs(float(d @ gt[:3, 2])))
    R = est[:3, :3].T @ gt[:3, :3]
    rot.append(float(torch.rad2deg(torch.acos(((R.trace() - 1) / 2).clamp(-1, 1)))))
err, along, rot = np.array(err), np.array(along), np.array(rot)
step = np.array([float((frames[i]["c2w"][:3, 3] - frames[i - 1]["c2w"][:3, 3]).norm()) for i in range(1, N)])
print(f"path length {step.sum():.2f} m; ATE rmse {np.sqrt((err ** 2).mean()) * 100:.2f} cm, max {err.max() * 100:.2f} cm, final {err[-1] * 100:.2f} cm; "
      f"along view axis rmse {np.sqrt((along ** 2).mean()) * 100:.2f} cm; rotation rmse {np.sqrt((rot ** 2).mean()):.3f} deg max {rot.max():.3f}")
print("every 10th frame (cm):", [round(float(e) * 100, 1) for e in err[9::10]])
mv = pipe.tracker.RO_Tracker.MV
print("volume moves:", getattr(mv, "move_count", "?"), "origin", mv.vol_origin, "dim", mv.vol_dim)
