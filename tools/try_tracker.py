import sys, os, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.ROtracker import ROTracker
random.seed(0)
cfg = synthetic_config("office0")
if len(sys.argv) > 1 and sys.argv[1] == "small":
    cfg["cam"].update({"H": 240, "W": 320, "fx": 288.0, "fy": 288.0, "cx": 159.5, "cy": 119.5})
    cfg["volume"].update({"voxel_size": 0.02, "trunc": 0.06})
cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
nf = 40
ds = get_dataset(cfg, device="cuda", n_frames=nf)
tr = ROTracker(cfg, ds)
poses = [ds[0]["c2w"].numpy()]
errs = []
t0 = time.time()
for i in range(1, nf):
    b = ds[i]
    if i == 1: pred = poses[-1]
    else:
        pred = (poses[-1] @ np.linalg.inv(poses[-2])) @ poses[-1]
    e_pred = np.linalg.norm(pred[:3, 3] - b["c2w"].numpy()[:3, 3])
    est, rgb, dep = tr.do_tracking(pred.astype(np.float32), None, b, "cuda")
    tr.post_processing(i, est, rgb, dep, None)
    poses.append(est)
    errs.append((e_pred, np.linalg.norm(est[:3, 3] - b["c2w"].numpy()[:3, 3])))
torch.cuda.synchronize()
dt = time.time() - t0
errs = np.array(errs)
print("per-frame (pred, est) cm:", [(round(a * 100, 2), round(b * 100, 2)) for a, b in errs[:14]])
print("frames", nf - 1, "tracking fps", (nf - 1) / dt)
print("pred err mean %.4f  est err mean %.4f  final err %.4f max %.4f" % (errs[:, 0].mean(), errs[:, 1].mean(), errs[-1, 1], errs[:, 1].max()))
