"""dev helper: spread of the device-vs-host tracker estimate over repeated runs (tests/test_tracker_gpu.py thresholds)"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
warnings.simplefilter("ignore")
import test_tracker_gpu as T
dt, dr, e, r = [], [], [], []
for rep in range(12):
    est = {}
    for dev in (True, False):
        tr, b, gt, init = T._tracker_with_a_map(dev)
        est[dev], _, _ = tr.do_tracking(init, None, b, "cuda")
    fwd = gt[:3, 2]
    e0 = abs(float((init[:3, 3] - gt[:3, 3]) @ fwd))
    for dev in (True, False):
        e.append(abs(float((est[dev][:3, 3] - gt[:3, 3]) @ fwd)) / e0)
        r.append(float(np.arccos(np.clip((np.trace(est[dev][:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1))))
    dt.append(float(np.abs(est[True][:3, 3] - est[False][:3, 3]).max()))
    dr.append(float(np.abs(est[True][:3, :3] - est[False][:3, :3]).max()))
print("max |dt| device-host", max(dt), "max |dR|", max(dr), "max e1/e0", max(e), "max rot err", max(r))
print("dt:", [round(v * 1e3, 3) for v in dt])
