"""Recorded-sequence readers on tiny sequences written here with Pillow, and the checkpoint layout."""
import os

import numpy as np
import pytest
import torch

from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset


def _write_frames(color_dir, depth_dir, names, H=12, W=16, color_hw=None, ext="jpg"):
    from PIL import Image
    rng = np.random.default_rng(0)
    depths = []
    os.makedirs(color_dir, exist_ok=True)
    os.makedirs(depth_dir, exist_ok=True)
    for cn, dn in names:
        ch, cw = color_hw or (H, W)
        Image.fromarray(rng.integers(0, 255, (ch, cw, 3), dtype=np.uint8)).save(os.path.join(color_dir, cn), quality=95)
        d = rng.integers(500, 4000, (H, W)).astype(np.uint16)
        d[0, 0] = 0
        Image.fromarray(d).save(os.path.join(depth_dir, dn))
        depths.append(d)
    return depths


def _cfg(name, datadir, H=12, W=16, **cam):
    cfg = synthetic_config("office0")
    cfg["dataset"] = name
    cfg["data"].update({"datadir": str(datadir), "sc_factor": 1.0, "downsample": 1})
    cfg["cam"].update({"H": H, "W": W, "fx": 10.0, "fy": 10.0, "cx": 7.5, "cy": 5.5, "png_depth_scale": 1000.0, "crop_edge": 0})
    cfg["cam"].update(cam)
    return cfg


def test_replica_layout(tmp_path):
    n = 3
    depths = _write_frames(tmp_path / "results", tmp_path / "results", [(f"frame{i:06d}.jpg", f"depth{i:06d}.png") for i in range(n)])
    poses = [np.eye(4) + 0.01 * i for i in range(n)]
    with open(tmp_path / "traj.txt", "w") as f:
        for p in poses:
            f.write(" ".join(f"{v:.8f}" for v in p.reshape(-1)) + "\n")
    cfg = _cfg("replica", tmp_path, crop_edge=2)
    cfg["data"]["sc_factor"] = 2.0
    ds = get_dataset(cfg)
    assert len(ds) == n and (ds.H, ds.W) == (8, 12) and (ds.cx, ds.cy) == (5.5, 3.5) and cfg["cam"]["H"] == 8
    b = ds[1]
    assert b["frame_id"] == 1 and b["rgb"].shape == (8, 12, 3) and b["depth"].shape == (8, 12) and b["direction"].shape == (8, 12, 3)
    assert torch.allclose(b["depth"], torch.from_numpy(depths[1][2:-2, 2:-2].astype(np.float32) / 1000.0 * 2.0))
    assert 0.0 <= float(b["rgb"].min()) and float(b["rgb"].max()) <= 1.0
    exp = poses[1].copy()
    exp[:3, 3] *= 2.0                                       # translation scaled by sc_factor
    assert torch.allclose(b["c2w"], torch.from_numpy(exp).float())
    assert ds.num_rays_to_save == int(8 * 12 * cfg["mapping"]["n_pixels"])


def test_scannet_layout_numeric_order_and_resize(tmp_path):
    ids = [0, 2, 10]                                        # "10" sorts after "2" numerically, not lexically
    depths = _write_frames(tmp_path / "color", tmp_path / "depth", [(f"{i}.jpg", f"{i}.png") for i in ids], color_hw=(24, 32))
    os.makedirs(tmp_path / "pose")
    for i in ids:
        np.savetxt(tmp_path / "pose" / f"{i}.txt", np.eye(4) * (i + 1), fmt="%.6f")
    ds = get_dataset(_cfg("scannet", tmp_path))
    assert len(ds) == 3
    b = ds[2]
    assert b["rgb"].shape == (12, 16, 3)                    # colour resized to the depth resolution
    assert torch.allclose(b["depth"], torch.from_numpy(depths[2].astype(np.float32) / 1000.0))
    assert float(b["c2w"][0, 0]) == 11.0
    assert float(b["depth"][0, 0]) == 0.0


def test_tum_layout_association(tmp_path):
    ts = [1.000, 1.020, 1.040, 1.500]                       # 20 ms spacing: thinned to 32 Hz
    _write_frames(tmp_path / "rgb", tmp_path / "depth", [(f"{t:.3f}.png", f"{t:.3f}.png") for t in ts], ext="png")
    with open(tmp_path / "rgb.txt", "w") as f:
        f.writelines(f"{t:.3f} rgb/{t:.3f}.png\n" for t in ts)
    with open(tmp_path / "depth.txt", "w") as f:
        f.writelines(f"{t + 0.004:.3f} depth/{t:.3f}.png\n" for t in ts)
    with open(tmp_path / "groundtruth.txt", "w") as f:
        f.write("# header\n")
        f.writelines(f"{t:.3f} {i}.0 0 0 0 0 0 1\n" for i, t in enumerate(ts))
    cfg = _cfg("tum", tmp_path, crop_size=[6, 8], png_depth_scale=5000.0)
    ds = get_dataset(cfg)
    assert [float(p[0, 3]) for p in ds.poses] == [0.0, 2.0, 3.0]          # frame at +20 ms dropped (< 1/32 s)
    assert (ds.H, ds.W) == (6, 8) and ds.fx == pytest.approx(5.0) and cfg["cam"]["W"] == 8
    b = ds[0]
    assert b["rgb"].shape == (6, 8, 3) and b["depth"].shape == (6, 8) and b["direction"].shape == (6, 8, 3)
    assert torch.equal(b["c2w"][:3, :3], torch.eye(3))


def test_bs3d_layout_crop_size_and_quaternion_poses(tmp_path):
    """BASELINE config 4's data layout (reference datasets/dataset.py:538-673)."""
    from scipy.spatial.transform import Rotation
    ids = [0, 1, 2, 10]
    depths = _write_frames(tmp_path / "color", tmp_path / "depth", [(f"{i}.jpg", f"{i}.png") for i in ids], H=24, W=32)
    quats = Rotation.from_euler("xyz", [[0.1 * i, -0.2, 0.05 * i] for i in range(4)]).as_quat()
    with open(tmp_path / "poses.txt", "w") as f:
        for i, q in enumerate(quats):
            f.write(f"{100.0 + i:.3f} {i}.5 -1.0 2.0 " + " ".join(f"{v:.9f}" for v in q) + "\n")
    cfg = _cfg("BS3D", tmp_path, H=24, W=32, crop_size=[8, 12], crop_edge=2, fx=20.0, fy=20.0, cx=15.5, cy=11.5)
    ds = get_dataset(cfg)
    assert len(ds) == 4 and (ds.H, ds.W) == (8, 12)
    # intrinsics: scaled by (crop + 2 edge) / original, then the principal point moves in by the edge (:573-584)
    assert ds.fx == pytest.approx(20.0 * 16 / 32) and ds.fy == pytest.approx(20.0 * 12 / 24)
    assert ds.cx == pytest.approx(15.5 * 16 / 32 - 2) and ds.cy == pytest.approx(11.5 * 12 / 24 - 2)
    b = ds[3]                                               # "10.jpg" is the LAST frame (numeric order)
    assert b["frame_id"] == 3 and b["rgb"].shape == (8, 12, 3) and b["depth"].shape == (8, 12) and b["direction"].shape == (8, 12, 3)
    # depth: nearest-neighbour resize of the 24x32 image to 12x16 (every second pixel), then the 2-pixel edge cut off
    exp = torch.from_numpy(depths[3].astype(np.float32) / 1000.0)[0::2, 0::2][2:-2, 2:-2]
    assert torch.equal(b["depth"], exp)
    R = Rotation.from_quat(quats[3]).as_matrix()
    assert torch.allclose(b["c2w"][:3, :3], torch.from_numpy(R).float(), atol=1e-6)
    assert torch.allclose(b["c2w"][:3, 3], torch.tensor([3.5, -1.0, 2.0]))
    assert ds.num_rays_to_save == int(8 * 12 * cfg["mapping"]["n_pixels"])


def test_uhumans_layout_lists_paired_by_index_png_and_npy_depth(tmp_path):
    """BASELINE config 5's data layout (reference datasets/dataset.py:1207-1396)."""
    n = 3
    depths = _write_frames(tmp_path / "color", tmp_path / "depth", [(f"c{i}.png", f"d{i}.png") for i in range(n)], ext="png")
    np.save(tmp_path / "depth" / "d2.npy", depths[2].astype(np.float32) / 1000.0)     # metres
    with open(tmp_path / "color.txt", "w") as f:
        f.writelines(f"{i}.0 color/c{i}.png\n" for i in range(n))
    with open(tmp_path / "depth.txt", "w") as f:
        f.writelines(f"{i + 0.5} depth/d{i}.{'npy' if i == 2 else 'png'}\n" for i in range(n))     # stamps differ: rows pair by index
    with open(tmp_path / "pose.txt", "w") as f:
        f.writelines(f"{i}.0 {i}.0 0 0 0 0 0 1\n" for i in range(n))                    # no header row
    cfg = _cfg("uhumans", tmp_path, png_depth_scale=5000.0)                            # overridden: PNG depth is millimetres
    ds = get_dataset(cfg)
    assert len(ds) == n and [float(p[0, 3]) for p in ds.poses] == [0.0, 1.0, 2.0]
    b1, b2 = ds[1], ds[2]
    assert torch.allclose(b1["depth"], torch.from_numpy(depths[1].astype(np.float32) / 1000.0))
    assert torch.allclose(b2["depth"], torch.from_numpy(depths[2].astype(np.float32) / 1000.0))      # the .npy frame, in metres
    assert b1["rgb"].shape == (12, 16, 3) and b1["direction"].shape == (12, 16, 3)
    cfg = _cfg("uhumans", tmp_path, crop_size=[6, 8], crop_edge=1)
    ds = get_dataset(cfg)
    assert (ds.H, ds.W) == (4, 6) and ds[0]["rgb"].shape == (4, 6, 3) and cfg["cam"]["W"] == 6 and ds.cx == pytest.approx(7.5 * 0.5 - 1)


def test_unknown_layout_raises():
    cfg = _cfg("azure", "/nonexistent")
    with pytest.raises(NotImplementedError):
        get_dataset(cfg)
