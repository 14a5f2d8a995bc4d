"""RBA pose-refinement MLP (reference model/rba.py:1-100) without kornia: 7->256->256->256->6 ELU
MLP predicting per-keyframe (axis-angle, translation) residuals.  Host-side PyTorch (north_star
keeps the optimizer / pose graph in Python); angle-axis conversions restate kornia 0.6.12's
formulas (parity unpinned: kornia is not available here)."""
from __future__ import annotations

import torch
import torch.nn as nn


def angle_axis_to_rotation_matrix(aa: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Rodrigues with a first-order branch below theta^2 = eps. aa [N,3] -> [N,3,3]."""
    theta2 = (aa * aa).sum(-1)
    theta = torch.sqrt(theta2)
    w = aa / (theta[:, None] + eps)
    wx, wy, wz = w[:, 0], w[:, 1], w[:, 2]
    c, s = torch.cos(theta), torch.sin(theta)
    k = 1.0 - c
    R = torch.stack([c + wx * wx * k, wx * wy * k - wz * s, wy * s + wx * wz * k,
                     wz * s + wx * wy * k, c + wy * wy * k, -wx * s + wy * wz * k,
                     -wy * s + wx * wz * k, wx * s + wy * wz * k, c + wz * wz * k], -1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0], aa[:, 1], aa[:, 2]
    one = torch.ones_like(rx)
    T = torch.stack([one, -rz, ry, rz, one, -rx, -ry, rx, one], -1).view(-1, 3, 3)
    return torch.where((theta2 > eps)[:, None, None], R, T)


def rotation_matrix_to_angle_axis(R: torch.Tensor) -> torch.Tensor:
    """[N,3,3] -> [N,3] via the quaternion (w>=0 branch selection), like kornia's composition."""
    m = R
    tr = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    qw = torch.sqrt(torch.clamp(1.0 + tr, min=1e-12)) / 2.0
    qx = torch.sqrt(torch.clamp(1.0 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2], min=1e-12)) / 2.0
    qy = torch.sqrt(torch.clamp(1.0 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2], min=1e-12)) / 2.0
    qz = torch.sqrt(torch.clamp(1.0 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2], min=1e-12)) / 2.0
    qx = torch.copysign(qx, m[:, 2, 1] - m[:, 1, 2])
    qy = torch.copysign(qy, m[:, 0, 2] - m[:, 2, 0])
    qz = torch.copysign(qz, m[:, 1, 0] - m[:, 0, 1])
    sin_half = torch.sqrt(qx * qx + qy * qy + qz * qz)
    angle = 2.0 * torch.atan2(sin_half, qw)
    scale = torch.where(sin_half > 1e-8, angle / sin_half.clamp_min(1e-12), torch.full_like(angle, 2.0))
    return torch.stack([qx * scale, qy * scale, qz * scale], -1)


def make_c2w(r, t):
    c2w = torch.eye(4, dtype=r.dtype, device=r.device).unsqueeze(0).repeat(r.shape[0], 1, 1)
    c2w[:, :3, :3] = angle_axis_to_rotation_matrix(r)
    c2w[:, :3, 3] = t
    return c2w


class RBA(nn.Module):
    def __init__(self, num_cams, init_c2w=None, layers=2, scale=1e-2, out_dir=None, device=None):
        super().__init__()
        self.num_cams, self.scale, self.out_dir = num_cams, scale, out_dir
        dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        # plain tensors (not buffers), like the reference: they follow .to() only via _apply below
        self.init_r = torch.zeros((num_cams, 3), dtype=torch.float32, device=dev)
        self.init_t = torch.zeros((num_cams, 3), dtype=torch.float32, device=dev)
        self.init_c2w = torch.eye(4, dtype=torch.float32, device=dev).unsqueeze(0).repeat(num_cams, 1, 1)
        if init_c2w is not None:
            for i in range(num_cams):
                self.update_init_pose(i, init_c2w[i])
        act = nn.ELU(inplace=True)
        seq = nn.Sequential(nn.Linear(7, 256), act)
        for _ in range(layers):
            seq.append(nn.Sequential(nn.Linear(256, 256), act))
        seq.append(nn.Linear(256, 6))
        self.layers = nn.Sequential(*seq)

    def _apply(self, fn):
        super()._apply(fn)
        self.init_r, self.init_t, self.init_c2w = fn(self.init_r), fn(self.init_t), fn(self.init_c2w)
        return self

    def get_init_pose(self, cam_id):
        return self.init_c2w[cam_id]

    def update_init_pose(self, cam_id, c2w):
        c2w = c2w.detach().to(self.init_c2w)
        self.init_c2w[cam_id] = c2w
        self.init_r[cam_id] = rotation_matrix_to_angle_axis(c2w[:3, :3].reshape(1, 3, 3)).reshape(-1)
        self.init_t[cam_id] = c2w[:3, 3]

    def forward(self, cam_id):
        if not isinstance(cam_id, torch.Tensor):
            if cam_id == 0:
                return self.init_c2w[0]
            cam_id = torch.tensor([[cam_id]])
        cam_id = cam_id.to(self.init_c2w.device)
        x = (cam_id.type_as(self.init_c2w) / self.num_cams) * 2 - 1
        idx = cam_id.reshape(-1)
        init_r, init_t = self.init_r[idx], self.init_t[idx]
        out = self.layers(torch.cat([x.reshape(-1, 1), init_r, init_t], dim=-1)) * self.scale
        keep = (idx != 0).to(out.dtype)[:, None]     # camera 0 is the gauge: no correction (reference :90-91)
        out = out * keep
        return make_c2w(out[:, :3] + init_r, out[:, 3:] + init_t)
