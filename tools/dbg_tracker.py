import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.ROtracker import ROTracker
random.seed(0)
cfg = synthetic_config("office0")
cfg["cam"].update({"H": 240, "W": 320, "fx": 288.0, "fy": 288.0, "cx": 159.5, "cy": 119.5})
cfg["volume"].update({"voxel_size": 0.02, "trunc": 0.06})
cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
ds = get_dataset(cfg, device="cuda", n_frames=5)
tr = ROTracker(cfg, ds)
b = ds[1]
gt = b["c2w"].numpy(); init = ds[0]["c2w"].numpy()
print("init err", np.linalg.norm(init[:3,3]-gt[:3,3]))
# instrument
orig = tr.cal_transform
def wrapped(sv):
    ok, m, t = orig(sv)
    better = int((sv[1:] < sv[0]).sum())
    print(f"  origin {sv[0]:.5f} min {sv[1:].min():.5f} better {better}/{len(sv)-1} success {ok} mean_t {t[:3]} ss {tr.search_size[:3]} err {np.linalg.norm(tr.current_global_T - gt[:3,3]):.4f}")
    return ok, m, t
tr.cal_transform = wrapped
orig_eval = tr.evaluate_tsdf
def weval(*a):
    r = orig_eval(*a)
    print("  counts: min", r[2][:int(a[2])].min(), "max", r[2].max(), "level", a[1], "P", a[2])
    return r
tr.evaluate_tsdf = weval
est, _, _ = tr.do_tracking(init, None, b, "cuda")
print("final err", np.linalg.norm(est[:3,3]-gt[:3,3]))
