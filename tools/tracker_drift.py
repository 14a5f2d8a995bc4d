"""dev helper: where the tracker's long-sequence drift comes from.  MODE=perfect_map: every frame is tracked from the constant-velocity
prediction of its OWN estimates, but integrated into the volume at the ground-truth pose (no map corruption).  MODE=own (default):
integrated at its own estimate (the real loop, tracker only -- no mapper).  Prints the error every 10 frames, volume moves and jumps."""
import os, sys, random, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.ROtracker import ROTracker
from remixfusion_amd.mp_slam.tracker import constant_velocity
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
mode = os.environ.get("MODE", "own")
random.seed(0)
cfg = synthetic_config("scene0000")
cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0, "clutter": int(os.environ.get("CLUTTER", 48)), "tracker": True})
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ds = get_dataset(cfg, device="cuda", n_frames=N + 8)
    tr = ROTracker(cfg, ds)
est = [ds[0]["c2w"].numpy().astype(np.float32)]
errs, rots, moves, cam_err, cam_rot = [], [], [], [], []
for i in range(1, N):
    b = ds[i]
    gt = b["c2w"].numpy()
    if i == 1:
        pred = est[0].copy()
    else:
        pred = constant_velocity(torch.from_numpy(est[-2]), torch.from_numpy(est[-1])).numpy()
    pose, rgb, depth = tr.do_tracking(pred, None, b, "cuda")
    est.append(pose.copy())
    origin_before = np.array(tr.MV.vol_origin).copy()
    tr.post_processing(i, gt if mode == "perfect_map" else pose, rgb, depth, None)
    if not np.array_equal(origin_before, np.array(tr.MV.vol_origin)):
        moves.append(i)
    d = pose[:3, 3] - gt[:3, 3]
    errs.append(float(np.linalg.norm(d)))
    cam_err.append(gt[:3, :3].T @ d)
    Rd = gt[:3, :3].T @ pose[:3, :3]
    cam_rot.append(np.degrees([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]]) / 2)
    R = pose[:3, :3].T @ gt[:3, :3]
    rots.append(float(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))))
errs, rots = np.array(errs), np.array(rots)
ce, cr = np.array(cam_err), np.array(cam_rot)
print("error in the camera frame, cm: mean", np.round(ce.mean(0) * 100, 3), "std", np.round(ce.std(0) * 100, 3), "; rotation vector deg: mean", np.round(cr.mean(0), 3), "std", np.round(cr.std(0), 3))
vel = np.array([ds[i]["c2w"].numpy()[:3, :3].T @ (ds[i]["c2w"].numpy()[:3, 3] - ds[i - 1]["c2w"].numpy()[:3, 3]) for i in range(1, N)])
print("camera-frame velocity cm/frame: mean", np.round(vel.mean(0) * 100, 3), "; correlation of error with velocity per axis", [round(float(np.corrcoef(ce[:, k], vel[:, k])[0, 1]), 3) for k in range(3)])
print("mode", mode, "frames", N, "volume moved at frames", moves)
print("err cm every 10:", [round(e * 100, 1) for e in errs[9::10]])
print("rot deg every 10:", [round(r, 2) for r in rots[9::10]])
j = np.flatnonzero(np.diff(errs) > 0.02)
print("jumps > 2 cm between consecutive frames at:", [(int(k) + 2, round(float(errs[k + 1] - errs[k]) * 100, 1)) for k in j])
