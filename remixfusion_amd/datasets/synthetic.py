"""Seeded synthetic RGB-D streams: an analytic box room with two spheres,
exact ray-cast z-depth, procedural colour, smooth Lissajous orbit.  Produces the reference's
batch dict (datasets/dataset.py:276-282 there): frame_id, c2w [4,4], rgb [H,W,3] 0..1,
depth [H,W] metres, direction [H,W,3] = ((i-cx)/fx, (j-cy)/fy, 1).

Everything is a pure function of (config, frame id): noise and dropout come from an integer
hash of (seed, frame, pixel), so CPU and GPU generation agree bit for bit on the pattern.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch


def get_camera_rays(H, W, fx, fy=None, cx=None, cy=None, device="cpu"):
    """Pinhole ray directions, OpenCV convention (reference datasets/utils.py:24-56)."""
    if cx is None:
        cx, cy = 0.5 * W, 0.5 * H
    if fy is None:
        fy = fx
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                          torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    return torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], -1)


def _hash_u01(seed: int, frame: int, n: int, salt: int, device) -> torch.Tensor:
    """uniform (0,1) per pixel from a 32-bit integer hash (device independent)."""
    x = torch.arange(n, dtype=torch.int64, device=device)
    x = (x * 0x9E3779B1 + (seed & 0xFFFFFFFF) * 0x85EBCA77 + frame * 0xC2B2AE3D + salt * 0x27D4EB2F) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    x = (x * 0x297A2D39) & 0xFFFFFFFF
    x ^= x >> 15
    return (x.to(torch.float64) + 0.5) / 4294967296.0


class SyntheticRoom:
    """Dataset-like object with the attributes Mapper/SLAM read (H, W, fx, fy, cx, cy, poses,
    num_frames, num_rays_to_save, __getitem__, __len__)."""

    def __init__(self, cfg: Dict, device: str = "cpu", n_frames: Optional[int] = None):
        cam, syn = cfg["cam"], cfg["synthetic"]
        self.config = cfg
        self.device = torch.device(device)
        self.H, self.W = int(cam["H"]), int(cam["W"])
        self.fx, self.fy, self.cx, self.cy = float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"])
        self.num_frames = int(n_frames if n_frames is not None else syn["n_frames"])
        self.seed = int(syn["seed"])
        self.noise, self.dropout = float(syn["depth_noise"]), float(syn["dropout"])
        self.room = torch.tensor(syn["room"], dtype=torch.float32)
        self.total_pixels = self.H * self.W
        self.num_rays_to_save = int(self.total_pixels * cfg["mapping"]["n_pixels"])
        self.rays_d = get_camera_rays(self.H, self.W, self.fx, self.fy, self.cx, self.cy, device=self.device)
        self.frame_ids = list(range(self.num_frames))
        self._cache = {}
        self.poses = [self._pose(i) for i in range(self.num_frames)]
        lo, hi = self.room[:, 0], self.room[:, 1]
        ext = hi - lo
        ctr = 0.5 * (lo + hi)
        # two spheres: centre, radius
        self.spheres = [(ctr + ext * torch.tensor([0.22, 0.18, -0.30]), 0.16 * float(ext.min())),
                        (ctr + ext * torch.tensor([-0.25, -0.20, -0.35]), 0.12 * float(ext.min()))]
        # synthetic.clutter = K: K more spheres (seeded) between the camera's orbit and the walls.  The plain box room is
        # degenerate for a geometric tracker -- a translation along a flat wall changes no depth (the ROTracker slides there:
        # tools/tracker_dbg.py) -- so the tracker's sequence tests furnish the room; 0 (every other stream) changes nothing.
        k_extra = int(syn.get("clutter", 0))
        if k_extra > 0:
            rng = np.random.default_rng(self.seed + 7)
            added = 0
            while added < k_extra:
                u = rng.uniform(-0.5, 0.5, 3)
                if max(abs(u[0]), abs(u[1])) < 0.30:          # keep the orbit (|offset| <= 0.18 ext) and its surroundings free
                    continue
                r = float(rng.uniform(0.15, 0.45))
                self.spheres.append((ctr + ext * torch.tensor(u, dtype=torch.float32), r))
                added += 1

    def __len__(self):
        return self.num_frames

    def K(self) -> np.ndarray:
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=np.float32)

    # ---- trajectory: Lissajous orbit around the room centre, camera looking outward-ish
    def _pose(self, i: int) -> torch.Tensor:
        lo, hi = self.room[:, 0].numpy(), self.room[:, 1].numpy()
        ctr, ext = 0.5 * (lo + hi), hi - lo
        t = i / 30.0
        pos = ctr + np.array([0.18 * ext[0] * math.sin(0.50 * t), 0.18 * ext[1] * math.sin(0.37 * t + 0.6),
                              0.10 * ext[2] * math.sin(0.23 * t)])
        yaw = 0.35 * t + 0.4 * math.sin(0.11 * t)
        pitch = 0.18 * math.sin(0.29 * t)
        f = np.array([math.cos(yaw) * math.cos(pitch), math.sin(yaw) * math.cos(pitch), math.sin(pitch)])
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(f, up)
        right /= np.linalg.norm(right)
        down = np.cross(f, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, f, pos
        return torch.from_numpy(c2w.astype(np.float32))

    # ---- analytic ray cast
    def _render(self, c2w: torch.Tensor):
        dev = self.device
        R, o = c2w[:3, :3].to(dev), c2w[:3, 3].to(dev)
        d = torch.sum(self.rays_d[..., None, :] * R, -1).reshape(-1, 3)   # world dirs, z_cam = 1
        lo, hi = self.room[:, 0].to(dev), self.room[:, 1].to(dev)
        inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
        t_exit = torch.maximum((lo - o) * inv, (hi - o) * inv)             # camera is inside the box
        t_wall, axis = t_exit.min(dim=-1)
        t_hit = t_wall.clone()
        kind = torch.zeros_like(axis)                                      # 0 wall, 1.. sphere id
        for k, (c, r) in enumerate(self.spheres):
            oc = o - c.to(dev)
            a = (d * d).sum(-1)
            b = 2.0 * (d * oc).sum(-1)
            cc = (oc * oc).sum() - r * r
            disc = b * b - 4 * a * cc
            ts = (-b - torch.sqrt(disc.clamp_min(0))) / (2 * a)
            hit = (disc > 0) & (ts > 1e-3) & (ts < t_hit)
            t_hit = torch.where(hit, ts, t_hit)
            kind = torch.where(hit, torch.full_like(kind, k + 1), kind)
        p = o + d * t_hit[:, None]
        # procedural colour: 0.5 m checker on walls tinted by wall axis; spheres get a stripe pattern
        cell = torch.floor(p / 0.5).sum(-1)
        chk = 0.35 + 0.3 * (cell - 2 * torch.floor(cell / 2))
        tint = torch.tensor([[0.9, 0.55, 0.5], [0.5, 0.9, 0.55], [0.55, 0.5, 0.9]], device=dev)[axis]
        wall = chk[:, None] * tint + 0.1
        stripe = 0.5 + 0.4 * torch.sin(12.0 * p[:, 2:3] + 3.0 * kind[:, None].float())
        sph = torch.cat([stripe, 0.8 - 0.5 * stripe, 0.3 + 0.2 * stripe], -1)
        rgb = torch.where((kind > 0)[:, None], sph, wall).clamp(0, 1)
        return rgb.reshape(self.H, self.W, 3), t_hit.reshape(self.H, self.W)

    def prefetch(self, ids):
        """render frames once and keep them resident on ``device`` (HBM) for later __getitem__ calls."""
        for i in ids:
            if i not in self._cache:
                self._cache[i] = self._make(i)

    def __getitem__(self, index: int):
        if index in self._cache:
            return dict(self._cache[index])
        return self._make(index)

    def _make(self, index: int):
        c2w = self.poses[index]
        rgb, depth = self._render(c2w)
        n = self.total_pixels
        if self.noise > 0:
            u1 = _hash_u01(self.seed, index, n, 1, self.device)
            u2 = _hash_u01(self.seed, index, n, 2, self.device)
            g = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
            depth = depth + (self.noise * g).to(torch.float32).reshape(self.H, self.W)
        if self.dropout > 0:
            u3 = _hash_u01(self.seed, index, n, 3, self.device).reshape(self.H, self.W)
            depth = torch.where(u3 < self.dropout, torch.zeros_like(depth), depth)
        # quantise colour to 8 bit like an image file would (keeps MV colours integer valued)
        rgb = torch.floor(rgb * 255.0 + 0.5) / 255.0
        return {"frame_id": self.frame_ids[index], "c2w": c2w, "rgb": rgb.contiguous(),
                "depth": depth.to(torch.float32).contiguous(), "direction": self.rays_d}


def get_dataset(cfg: Dict, device: str = "cpu", n_frames: Optional[int] = None):
    """counterpart of datasets/dataset.py:12-53: the seeded synthetic stream, or a recorded sequence
    (`dataset: replica | scannet | tum`, see datasets/dataset.py)."""
    if cfg.get("dataset", "synthetic") != "synthetic":
        from .dataset import get_recorded_dataset
        return get_recorded_dataset(cfg)
    return SyntheticRoom(cfg, device=device, n_frames=n_frames)
