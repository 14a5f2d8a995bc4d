import os
import sys

# bounded BLAS / OpenMP pools (the CPU oracle is the only heavy host work here): on a box that shows hundreds of cores behind
# a small CPU quota, default-sized pools get the test process throttled
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "16")

import numpy as np
import pytest

# the tracker tests search with the REFERENCE's particle templates: a fixture of this test suite (tests/golden/README.md), which
# the product neither ships nor looks for by itself (remixfusion_amd/model/pst.py)
os.environ.setdefault("RFX_PST_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pst_templates.npz"))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def look_at(pos, fwd, up=(0.0, 0.0, 1.0)):
    """OpenCV c2w (x right, y down, z forward) from a position and a forward vector."""
    f = np.asarray(fwd, np.float64)
    f = f / np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    r = r / np.linalg.norm(r)
    d = np.cross(f, r)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, d, f, pos
    return c2w.astype(np.float32)


def small_frame(H=120, W=160, seed=0, frame=0, name="office0"):
    """One synthetic RGB-D frame at reduced resolution (numpy): K, c2w, rgb255, depth."""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.datasets import get_dataset
    cfg = synthetic_config(name)
    cfg["cam"].update({"H": H, "W": W, "fx": 0.9 * W, "fy": 0.9 * W, "cx": (W - 1) / 2.0, "cy": (H - 1) / 2.0})
    cfg["synthetic"]["seed"] = 20251205 + seed
    ds = get_dataset(cfg, n_frames=frame + 1)
    b = ds[frame]
    rgb255 = np.floor(b["rgb"].numpy() * 255.0).astype(np.float32)      # the reference's quantisation (model/ROtracker.py:82), as pipeline.py
    return ds.K(), b["c2w"].numpy(), rgb255, b["depth"].numpy(), b["rgb"].numpy()
