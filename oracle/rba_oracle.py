"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's pose-refinement MLP (RBA), independent of the product code.

Reference: ``model/rba.py:1-100`` (RBA: 7 -> 256 -> 256 -> 256 -> 6 ELU MLP, outputs scaled by ``scale`` and added to the
initial (angle-axis, translation) of each camera; camera 0 is the gauge; ``make_c2w`` :7-20).  The two conversions it imports
(``rba.py:3-4``) live in **kornia** (``kornia==0.6.12``, reference requirements.txt; absent from /root/reference and not
installed here): ``kornia.geometry.conversions.angle_axis_to_rotation_matrix`` and ``rotation_matrix_to_angle_axis``
(= ``quaternion_to_angle_axis(rotation_matrix_to_quaternion(R))``).  This file restates their published algorithm of that
release, element by element (the nine Rodrigues entries with ``w = aa / (theta + 1e-6)``, the first-order branch below
``theta^2 = 1e-6``; the four-branch matrix -> quaternion map with ``eps = 1e-8`` and the clamped divisions; the
``2 atan2`` quaternion -> angle-axis map).  **Parity unpinned** against kornia itself (no fixture of it exists in the
reference); pinned by analytic properties in tests/test_oracle_rba.py (orthogonality, known rotations, round trips, the
first-order limit).  Only tests/ may import this module.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn.functional as F


def angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    """kornia 0.6.12 geometry/conversions.py ``angle_axis_to_rotation_matrix``: [N,3] -> [N,3,3]."""
    eps = 1e-6
    aa = angle_axis
    theta2 = (aa.unsqueeze(1) @ aa.unsqueeze(1).transpose(1, 2)).squeeze(1)          # [N,1]
    theta = torch.sqrt(theta2)
    wxyz = aa / (theta + eps)
    wx, wy, wz = torch.chunk(wxyz, 3, dim=1)
    c, s = torch.cos(theta), torch.sin(theta)
    one = 1.0
    r00 = c + wx * wx * (one - c)
    r10 = wz * s + wx * wy * (one - c)
    r20 = -wy * s + wx * wz * (one - c)
    r01 = wx * wy * (one - c) - wz * s
    r11 = c + wy * wy * (one - c)
    r21 = wx * s + wy * wz * (one - c)
    r02 = wy * s + wx * wz * (one - c)
    r12 = -wx * s + wy * wz * (one - c)
    r22 = c + wz * wz * (one - c)
    normal = torch.cat([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)
    rx, ry, rz = torch.chunk(aa, 3, dim=1)
    k1 = torch.ones_like(rx)
    taylor = torch.cat([k1, -rz, ry, rz, k1, -rx, -ry, rx, k1], dim=1).view(-1, 3, 3)
    mask = (theta2 > eps).view(-1, 1, 1)
    mask_pos, mask_neg = mask.type_as(theta2), (~mask).type_as(theta2)
    return mask_pos * normal + mask_neg * taylor


def rotation_matrix_to_quaternion_wxyz(R: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """kornia 0.6.12 ``rotation_matrix_to_quaternion(..., order=WXYZ)``: [N,3,3] -> [N,4] (w, x, y, z)."""
    tiny = torch.finfo(R.dtype).tiny

    def sdiv(num, den):
        return num / torch.clamp(den, min=tiny)

    m = R.reshape(*R.shape[:-2], 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(m, 9, dim=-1)
    trace = m00 + m11 + m22

    def trace_positive():
        sq = torch.sqrt(trace + 1.0 + eps) * 2.0                       # 4 qw
        return torch.cat([0.25 * sq, sdiv(m21 - m12, sq), sdiv(m02 - m20, sq), sdiv(m10 - m01, sq)], dim=-1)

    def cond_1():
        sq = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0             # 4 qx
        return torch.cat([sdiv(m21 - m12, sq), 0.25 * sq, sdiv(m01 + m10, sq), sdiv(m02 + m20, sq)], dim=-1)

    def cond_2():
        sq = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0             # 4 qy
        return torch.cat([sdiv(m02 - m20, sq), sdiv(m01 + m10, sq), 0.25 * sq, sdiv(m12 + m21, sq)], dim=-1)

    def cond_3():
        sq = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0             # 4 qz
        return torch.cat([sdiv(m10 - m01, sq), sdiv(m02 + m20, sq), sdiv(m12 + m21, sq), 0.25 * sq], dim=-1)

    where_2 = torch.where(m11 > m22, cond_2(), cond_3())
    where_1 = torch.where((m00 > m11) & (m00 > m22), cond_1(), where_2)
    return torch.where(trace > 0.0, trace_positive(), where_1)


def quaternion_wxyz_to_angle_axis(q: torch.Tensor) -> torch.Tensor:
    """kornia 0.6.12 ``quaternion_to_angle_axis(..., order=WXYZ)``: [N,4] -> [N,3]."""
    cos_theta, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    sin2 = q1 * q1 + q2 * q2 + q3 * q3
    sin_theta = torch.sqrt(sin2)
    two_theta = 2.0 * torch.where(cos_theta < 0.0, torch.atan2(-sin_theta, -cos_theta), torch.atan2(sin_theta, cos_theta))
    k = torch.where(sin2 > 0.0, two_theta / sin_theta, 2.0 * torch.ones_like(sin_theta))
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)


def rotation_matrix_to_angle_axis(R: torch.Tensor) -> torch.Tensor:
    """kornia 0.6.12 ``rotation_matrix_to_angle_axis``: matrix -> quaternion (WXYZ) -> angle-axis."""
    return quaternion_wxyz_to_angle_axis(rotation_matrix_to_quaternion_wxyz(R))


def make_c2w(r: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """model/rba.py:7-20."""
    c2w = torch.eye(4, dtype=r.dtype).unsqueeze(0).repeat(r.shape[0], 1, 1)
    c2w[:, :3, :3] = angle_axis_to_rotation_matrix(r)
    c2w[:, :3, 3] = t
    return c2w


def rba_forward(params: Sequence[torch.Tensor], init_r: torch.Tensor, init_t: torch.Tensor, cam_id: torch.Tensor, num_cams: int,
                scale: float) -> torch.Tensor:
    """model/rba.py:71-100.  params = (W0, b0, W1, b1, W2, b2, W3, b3) of the four Linear layers (torch layout [out, in]);
    cam_id [K] int64.  Returns c2w [K,4,4]."""
    x = (cam_id.to(init_r.dtype).reshape(-1, 1) / num_cams) * 2 - 1                   # :83
    r0, t0 = init_r[cam_id], init_t[cam_id]
    h = torch.cat([x, r0, t0], dim=-1)                                                # :91
    n_lin = len(params) // 2
    for i in range(n_lin):
        h = F.linear(h, params[2 * i], params[2 * i + 1])
        if i + 1 < n_lin:
            h = F.elu(h)
    out = h * scale                                                                   # :93
    out = torch.where((cam_id == 0).reshape(-1, 1), torch.zeros_like(out), out)       # :95-96 (`if 0 in cam_id: out[0] = 0`,
    #                                                                                   camera 0 leads the list in every caller)
    return make_c2w(out[:, :3] + r0, out[:, 3:] + t0)
