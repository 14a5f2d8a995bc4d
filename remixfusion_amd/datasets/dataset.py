"""Readers for recorded RGB-D sequences (reference datasets/dataset.py:12-53, 55-87, 203-299, 538-673,
675-780, 1009-1205, 1207-1396): Replica, ScanNet, TUM-RGBD, BS3D and uHumans2 layouts -- the data of BASELINE configs
2 to 5 -- returning the batch dict the Mapper / Tracker consume (`frame_id, c2w, rgb [H,W,3] 0..1, depth [H,W]
metres*sc_factor, direction [H,W,3]`).

Control-plane I/O, kept small: images are decoded with Pillow (cv2 / imageio are not in this image) and
resampled with torch (`bilinear, align_corners=False` = cv2.INTER_LINEAR, `area` = INTER_AREA, `nearest` =
INTER_NEAREST).  The other layouts of the reference (iPhone, Azure, Realsense, FastSyn, Largeindoor) differ only
in file globbing and pose-file syntax and are not reproduced.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from .synthetic import get_camera_rays


def _read_rgb(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def _read_depth_png(path: str) -> np.ndarray:
    from PIL import Image
    if not path.endswith(".png"):
        raise NotImplementedError("only 16-bit PNG depth is supported (reference raises for .exr too)")
    with Image.open(path) as im:
        return np.asarray(im).astype(np.float32)


def _resize(img: torch.Tensor, hw, mode: str) -> torch.Tensor:
    """img [H,W] or [H,W,C] -> resized, through F.interpolate."""
    x = img[None, None] if img.dim() == 2 else img.permute(2, 0, 1)[None]
    kw = {"align_corners": False} if mode == "bilinear" else {}
    y = F.interpolate(x, size=tuple(hw), mode=mode, **kw)
    return y[0, 0] if img.dim() == 2 else y[0].permute(1, 2, 0).contiguous()


class BaseDataset(torch.utils.data.Dataset):
    """camera bookkeeping shared by the readers (reference :55-87)."""

    def __init__(self, cfg: Dict):
        cam, ds = cfg["cam"], cfg["data"]["downsample"]
        self.config = cfg
        self.png_depth_scale = cam["png_depth_scale"]
        self.H, self.W = cam["H"] // ds, cam["W"] // ds
        if ds > 1:
            self.fx, self.fy, self.cx, self.cy = cam["fx"] // ds, cam["fy"] // ds, cam["cx"] // ds, cam["cy"] // ds
        else:
            self.fx, self.fy, self.cx, self.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        self.distortion = np.array(cam["distortion"]) if "distortion" in cam else None
        self.crop_edge = cam.get("crop_edge", 0)
        self.ignore_w, self.ignore_h = cfg["tracking"]["ignore_edge_W"], cfg["tracking"]["ignore_edge_H"]
        if "crop_size" in cam:
            self.total_pixels = cam["crop_size"][0] * cam["crop_size"][1]
        else:
            self.total_pixels = (self.H - self.crop_edge * 2) * (self.W - self.crop_edge * 2)
        self.num_rays_to_save = int(self.total_pixels * cfg["mapping"]["n_pixels"])
        self.rays_d = None
        self.poses: List[torch.Tensor] = []

    def _apply_crop_edge(self):
        """shrink the camera by cam.crop_edge on every side, in the config too (reference :226-232)."""
        e = self.config["cam"].get("crop_edge", 0)
        if e > 0:
            self.H -= 2 * e
            self.W -= 2 * e
            self.cx -= e
            self.cy -= e
            self.config["cam"]["H"] -= 2 * e
            self.config["cam"]["W"] -= 2 * e

    def _frame(self, color_path: str, depth_path: str, crop_size=None, depth_array=None):
        if self.distortion is not None and crop_size is None:
            raise NotImplementedError("lens undistortion is not implemented (the reference raises here too)")
        color = torch.from_numpy(_read_rgb(color_path).astype(np.float32) / 255.0)
        raw = _read_depth_png(depth_path) if depth_array is None else depth_array
        depth = torch.from_numpy(raw / self.png_depth_scale * self.sc_factor)
        H, W = depth.shape
        if color.shape[:2] != (H, W):
            color = _resize(color, (H, W), "bilinear")
        if self.downsample_factor > 1:
            H, W = H // self.downsample_factor, W // self.downsample_factor
            color = _resize(color, (H, W), "area")
            depth = _resize(depth, (H, W), "nearest")
        if crop_size is not None:       # "actually is resize" (reference :1181-1188)
            x = color.permute(2, 0, 1)[None]
            color = F.interpolate(x, tuple(crop_size), mode="bilinear", align_corners=True)[0].permute(1, 2, 0).contiguous()
            depth = F.interpolate(depth[None, None], tuple(crop_size), mode="nearest")[0, 0]
        e = self.config["cam"].get("crop_edge", 0)
        if e > 0:
            color, depth = color[e:-e, e:-e], depth[e:-e, e:-e]
        if self.rays_d is None:
            self.rays_d = get_camera_rays(self.H, self.W, self.fx, self.fy, self.cx, self.cy)
        return color.contiguous().float(), depth.contiguous().float()

    def _item(self, index, color, depth):
        return {"frame_id": self.frame_ids[index], "c2w": self.poses[index], "rgb": color, "depth": depth,
                "direction": self.rays_d}

    def __len__(self):
        return self.num_frames


class ReplicaDataset(BaseDataset):
    """<dir>/results/frame*.jpg, depth*.png, <dir>/traj.txt with one row-major 4x4 per line (reference :203-299)."""

    def __init__(self, cfg, basedir, trainskip=1, downsample_factor=1, translation=0.0, sc_factor=1.0, crop=0):
        super().__init__(cfg)
        self.basedir, self.trainskip, self.downsample_factor = basedir, trainskip, downsample_factor
        self.translation, self.sc_factor, self.crop = translation, sc_factor, crop
        self.img_files = sorted(glob.glob(f"{basedir}/results/frame*.jpg"))
        self.depth_paths = sorted(glob.glob(f"{basedir}/results/depth*.png"))
        self.load_poses(os.path.join(basedir, "traj.txt"))
        self.frame_ids = range(0, len(self.img_files))
        self.num_frames = len(self.frame_ids)
        self._apply_crop_edge()

    def load_poses(self, path):
        with open(path, "r") as f:
            lines = f.readlines()
        self.poses = []
        for i in range(len(self.img_files)):
            c2w = np.array(list(map(float, lines[i].split()))).reshape(4, 4)
            c2w[:3, 3] *= self.sc_factor
            self.poses.append(torch.from_numpy(c2w).float())

    def __getitem__(self, index):
        return self._item(index, *self._frame(self.img_files[index], self.depth_paths[index]))


class ScannetDataset(BaseDataset):
    """<dir>/color/N.jpg, depth/N.png, pose/N.txt (4 lines of 4 numbers), numeric order (reference :675-780)."""

    def __init__(self, cfg, basedir, trainskip=1, downsample_factor=1, translation=0.0, sc_factor=1.0, crop=0):
        super().__init__(cfg)
        self.basedir, self.trainskip, self.downsample_factor = basedir, trainskip, downsample_factor
        self.translation, self.sc_factor, self.crop = translation, sc_factor, crop
        num = lambda p: int(os.path.basename(p)[:-4])                                        # noqa: E731
        self.img_files = sorted(glob.glob(os.path.join(basedir, "color", "*.jpg")), key=num)
        self.depth_paths = sorted(glob.glob(os.path.join(basedir, "depth", "*.png")), key=num)
        self.load_poses(os.path.join(basedir, "pose"))
        self.frame_ids = range(0, len(self.img_files))
        self.num_frames = len(self.frame_ids)
        self._apply_crop_edge()

    def load_poses(self, path):
        self.poses = []
        for p in sorted(glob.glob(os.path.join(path, "*.txt")), key=lambda x: int(os.path.basename(x)[:-4])):
            with open(p, "r") as f:
                vals = [float(v) for line in f.readlines() for v in line.split()]
            self.poses.append(torch.from_numpy(np.array(vals).reshape(4, 4)).float())

    def __getitem__(self, index):
        if self.downsample_factor > 1:            # the reference re-divides the focal length on every read (:733-734)
            self.fx, self.fy = self.fx // self.downsample_factor, self.fy // self.downsample_factor
        return self._item(index, *self._frame(self.img_files[index], self.depth_paths[index]))


class TUMDataset(BaseDataset):
    """TUM RGB-D: rgb.txt / depth.txt / groundtruth.txt (or pose.txt) associated by time stamp within 0.08 s and
    thinned to 32 Hz; images resized to cam.crop_size (reference :1009-1205)."""

    def __init__(self, cfg, basedir, align=True, trainskip=1, downsample_factor=1, translation=0.0, sc_factor=1.0,
                 crop=0, load=True):
        super().__init__(cfg)
        self.basedir, self.trainskip, self.downsample_factor = basedir, trainskip, downsample_factor
        self.translation, self.sc_factor, self.crop = translation, sc_factor, crop
        self.color_paths, self.depth_paths, self.poses = self.loadtum(basedir, frame_rate=32)
        self.frame_ids = range(0, len(self.color_paths))
        self.num_frames = len(self.frame_ids)
        cam = cfg["cam"]
        self.crop_size = cam["crop_size"] if "crop_size" in cam else None
        if self.crop_size is not None:
            sx, sy = self.crop_size[1] / self.W, self.crop_size[0] / self.H
            self.fx, self.fy, self.cx, self.cy = sx * self.fx, sy * self.fy, sx * self.cx, sy * self.cy
            self.H, self.W = self.crop_size[0], self.crop_size[1]
            cam.update({"H": self.H, "W": self.W, "fx": self.fx, "fy": self.fy, "cx": self.cx, "cy": self.cy})
        e = cam.get("crop_edge", 0)
        if e > 0:
            self.H -= 2 * e
            self.W -= 2 * e
            self.cx -= e
            self.cy -= e
            cam.update({"H": self.H, "W": self.W, "cx": self.cx, "cy": self.cy})

    @staticmethod
    def pose_matrix_from_quaternion(pvec):
        """(tx ty tz qx qy qz qw) -> 4x4."""
        from scipy.spatial.transform import Rotation
        pose = np.eye(4)
        pose[:3, :3] = Rotation.from_quat(pvec[3:]).as_matrix()
        pose[:3, 3] = pvec[:3]
        return pose

    @staticmethod
    def associate_frames(tstamp_image, tstamp_depth, tstamp_pose, max_dt=0.08):
        out = []
        for i, t in enumerate(tstamp_image):
            j = int(np.argmin(np.abs(tstamp_depth - t)))
            if tstamp_pose is None:
                if np.abs(tstamp_depth[j] - t) < max_dt:
                    out.append((i, j))
            else:
                k = int(np.argmin(np.abs(tstamp_pose - t)))
                if np.abs(tstamp_depth[j] - t) < max_dt and np.abs(tstamp_pose[k] - t) < max_dt:
                    out.append((i, j, k))
        return out

    @staticmethod
    def parse_list(filepath, skiprows=0):
        return np.loadtxt(filepath, delimiter=" ", dtype=np.str_, skiprows=skiprows)

    def loadtum(self, datapath, frame_rate=-1):
        pose_list = os.path.join(datapath, "groundtruth.txt")
        if not os.path.isfile(pose_list):
            pose_list = os.path.join(datapath, "pose.txt")
        image_data = self.parse_list(os.path.join(datapath, "rgb.txt"))
        depth_data = self.parse_list(os.path.join(datapath, "depth.txt"))
        pose_data = self.parse_list(pose_list, skiprows=1)
        pose_vecs = pose_data[:, 1:].astype(np.float64)
        t_img, t_dep, t_pose = (d[:, 0].astype(np.float64) for d in (image_data, depth_data, pose_data))
        assoc = self.associate_frames(t_img, t_dep, t_pose)
        keep = [0]
        for i in range(1, len(assoc)):
            if t_img[assoc[i][0]] - t_img[assoc[keep[-1]][0]] > 1.0 / frame_rate:
                keep.append(i)
        images, depths, poses = [], [], []
        for ix in keep:
            i, j, k = assoc[ix]
            images.append(os.path.join(datapath, image_data[i, 1]))
            depths.append(os.path.join(datapath, depth_data[j, 1]))
            poses.append(torch.from_numpy(self.pose_matrix_from_quaternion(pose_vecs[k])).float())
        return images, depths, poses

    def __getitem__(self, index):
        if self.downsample_factor > 1:
            self.fx, self.fy = self.fx // self.downsample_factor, self.fy // self.downsample_factor
        return self._item(index, *self._frame(self.color_paths[index], self.depth_paths[index], crop_size=self.crop_size))


class BS3DDataset(BaseDataset):
    """BS3D (BASELINE config 4): <dir>/color/N.jpg, depth/N.png in numeric order, <dir>/poses.txt with one
    `stamp tx ty tz qx qy qz qw` row per frame; with cam.crop_size the images are resized to crop_size + 2 crop_edge and the
    edge is cut off, the intrinsics following (reference :538-673)."""

    def __init__(self, cfg, basedir, trainskip=1, downsample_factor=1, translation=0.0, sc_factor=1.0, crop=0):
        super().__init__(cfg)
        self.basedir, self.trainskip, self.downsample_factor = basedir, trainskip, downsample_factor
        self.translation, self.sc_factor, self.crop = translation, sc_factor, crop
        num = lambda p: int(os.path.basename(p)[:-4])                                        # noqa: E731
        self.img_files = sorted(glob.glob(os.path.join(basedir, "color", "*.jpg")), key=num)
        self.depth_paths = sorted(glob.glob(os.path.join(basedir, "depth", "*.png")), key=num)
        self.frame_ids = range(0, len(self.img_files))
        self.load_poses(os.path.join(basedir, "poses.txt"))
        self.num_frames = len(self.frame_ids)
        cam = cfg["cam"]
        self.out_hw = None
        if "crop_size" in cam:                                  # reference :569-584
            e = cam["crop_edge"]
            self.H_out, self.W_out = cam["crop_size"]
            He, We = self.H_out + 2 * e, self.W_out + 2 * e
            self.fx *= We / self.W
            self.fy *= He / self.H
            self.cx *= We / self.W
            self.cy *= He / self.H
            self.H, self.W = He - 2 * e, We - 2 * e
            self.cx -= e
            self.cy -= e
            self.out_hw = (He, We)

    def load_poses(self, path):
        rows = np.loadtxt(path, dtype=np.float64, ndmin=2)[:, 1:]
        self.poses = [torch.from_numpy(TUMDataset.pose_matrix_from_quaternion(r)).float() for r in rows]

    def __getitem__(self, index):
        if self.distortion is not None:
            raise NotImplementedError("lens undistortion is not implemented (the reference raises here too)")
        color = torch.from_numpy(_read_rgb(self.img_files[index]).astype(np.float32) / 255.0)
        depth = torch.from_numpy(_read_depth_png(self.depth_paths[index]) / self.png_depth_scale * self.sc_factor)
        H, W = depth.shape
        if self.out_hw is not None:
            color = _resize(color, self.out_hw, "bilinear")
            depth = _resize(depth, self.out_hw, "nearest")
        elif color.shape[:2] != (H, W):
            color = _resize(color, (H, W), "bilinear")
        if self.downsample_factor > 1:                          # (the reference re-divides the focal length on every read)
            H, W = H // self.downsample_factor, W // self.downsample_factor
            self.fx, self.fy = self.fx // self.downsample_factor, self.fy // self.downsample_factor
            color = _resize(color, (H, W), "area")
            depth = _resize(depth, (H, W), "nearest")
        e = self.config["cam"].get("crop_edge", 0)
        if e > 0:
            color, depth = color[e:-e, e:-e], depth[e:-e, e:-e]
        if self.rays_d is None:
            self.rays_d = get_camera_rays(self.H, self.W, self.fx, self.fy, self.cx, self.cy)
        return self._item(index, color.contiguous().float(), depth.contiguous().float())


class uhumansDataset(TUMDataset):
    """uHumans2 (BASELINE config 5): color.txt / depth.txt / pose.txt list files whose rows pair up by index (no time-stamp
    association, no header row); depth as 16-bit PNG in millimetres or as .npy in metres (reference :1207-1396)."""

    def loadtum(self, datapath, frame_rate=-1):
        image_data = self.parse_list(os.path.join(datapath, "color.txt"))
        depth_data = self.parse_list(os.path.join(datapath, "depth.txt"))
        pose_vecs = self.parse_list(os.path.join(datapath, "pose.txt"))[:, 1:].astype(np.float64)
        images = [os.path.join(datapath, image_data[i, 1]) for i in range(image_data.shape[0])]
        depths = [os.path.join(datapath, depth_data[i, 1]) for i in range(image_data.shape[0])]
        poses = [torch.from_numpy(self.pose_matrix_from_quaternion(pose_vecs[i])).float() for i in range(image_data.shape[0])]
        return images, depths, poses

    def __getitem__(self, index):
        depth_path = self.depth_paths[index]
        self.png_depth_scale = 1.0 if depth_path.endswith(".npy") else 1000.0          # reference :1327-1332
        if self.downsample_factor > 1:
            self.fx, self.fy = self.fx // self.downsample_factor, self.fy // self.downsample_factor
        if depth_path.endswith(".npy"):
            color, depth = self._frame(self.color_paths[index], depth_path, crop_size=self.crop_size,
                                       depth_array=np.load(depth_path).astype(np.float32))
        else:
            color, depth = self._frame(self.color_paths[index], depth_path, crop_size=self.crop_size)
        return self._item(index, color, depth)


_READERS = {"replica": ReplicaDataset, "scannet": ScannetDataset, "tum": TUMDataset, "BS3D": BS3DDataset, "uhumans": uhumansDataset}


def get_recorded_dataset(config: Dict):
    """reference get_dataset (:12-53) for the layouts above."""
    name = config["dataset"]
    if name not in _READERS:
        raise NotImplementedError(f"dataset {name!r}: only {sorted(_READERS)} and 'synthetic' are implemented")
    d = config["data"]
    return _READERS[name](config, d["datadir"], trainskip=d["trainskip"], downsample_factor=d["downsample"],
                          sc_factor=d["sc_factor"])
