"""RBA pose-refinement MLP (reference model/rba.py:1-100) without kornia: 7->256->256->256->6 ELU
MLP predicting per-keyframe (axis-angle, translation) residuals.  ``forward`` runs the fused librfx
kernels (``rfx_rba_forward/backward``: 3 launches instead of ~80 ATen ops per bundle-adjustment
iteration); ``forward_torch`` is the same arithmetic as plain tensor ops (any layer count, and the
reference formulation the tests compare against).  Angle-axis conversions restate kornia 0.6.12's
formulas (parity unpinned: kornia is not available here)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, ptr, stream_ptr


class _RbaFn(torch.autograd.Function):
    """poses [K,4,4] = RBA(cam_ids) through librfx; gradients for the eight Linear tensors."""

    @staticmethod
    def forward(ctx, mod, idx, *params):
        lib = _lib.load()
        K = idx.shape[0]
        dev = idx.device
        prm = [p.detach() for p in params]
        desc = _lib.RbaParams(*[ptr(p) for p in prm], 256)
        poses = torch.empty((K, 4, 4), dtype=torch.float32, device=dev)
        acts = torch.empty(int(lib.rfx_rba_acts_floats(K)), dtype=torch.float32, device=dev)
        check(lib.rfx_rba_forward(C.byref(desc), ptr(mod.init_r), ptr(mod.init_t), idx.data_ptr(), K, mod.num_cams,
                                  float(mod.scale), ptr(poses), ptr(acts), stream_ptr(dev)), "rfx_rba_forward")
        ctx.save_for_backward(acts, *prm)
        ctx.scale, ctx.K = float(mod.scale), K
        return poses

    @staticmethod
    def backward(ctx, dposes):
        lib = _lib.load()
        acts, *prm = ctx.saved_tensors
        dev = acts.device
        desc = _lib.RbaParams(*[ptr(p) for p in prm], 256)
        grads = [torch.empty_like(p) if need else None for p, need in zip(prm, ctx.needs_input_grad[2:])]
        gdesc = _lib.RbaGrads(*[ptr(g) for g in grads])
        ws = torch.empty(int(lib.rfx_rba_grads_floats(ctx.K)), dtype=torch.float32, device=dev)
        dp = dposes.to(torch.float32).contiguous()
        check(lib.rfx_rba_backward(C.byref(desc), ptr(acts), ctx.K, ptr(dp), ctx.scale, C.byref(gdesc), ptr(ws),
                                   stream_ptr(dev)), "rfx_rba_backward")
        return (None, None, *grads)


_LEVI = None


def _skew(v: torch.Tensor) -> torch.Tensor:
    """[N,3] -> [N,3,3] cross-product matrices with one matmul against the Levi-Civita tensor."""
    global _LEVI
    if _LEVI is None or _LEVI.device != v.device or _LEVI.dtype != v.dtype:
        e = torch.zeros(3, 3, 3, dtype=v.dtype)
        e[0, 1, 2] = e[1, 2, 0] = e[2, 0, 1] = -1.0      # skew(v)[i,j] = -eps_ijk v_k
        e[0, 2, 1] = e[2, 1, 0] = e[1, 0, 2] = 1.0
        _LEVI = e.to(v.device)
    return torch.matmul(v, _LEVI.reshape(9, 3).t()).reshape(-1, 3, 3)


def angle_axis_to_rotation_matrix(aa: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Rodrigues R = cos(t) I + sin(t) [w]x + (1-cos(t)) w w^T with w = aa/(t+eps), and the first-order
    branch I + [aa]x below t^2 = eps (kornia 0.6.12 semantics), in a dozen tensor ops.  aa [N,3]."""
    theta2 = (aa * aa).sum(-1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = aa / (theta + eps)
    c, s = torch.cos(theta)[..., None], torch.sin(theta)[..., None]
    eye = torch.eye(3, dtype=aa.dtype, device=aa.device)
    R = c * eye + s * _skew(w) + (1.0 - c) * (w[:, :, None] * w[:, None, :])
    T = eye + _skew(aa)
    return torch.where((theta2 > eps)[..., None], R, T)


def rotation_matrix_to_angle_axis(R: torch.Tensor) -> torch.Tensor:
    """[N,3,3] -> [N,3].  Same map as kornia's (matrix -> quaternion -> angle-axis) but evaluated from
    the antisymmetric part, which stays accurate for small rotations; the quaternion/diagonal form is
    only used next to pi where sin(theta) vanishes."""
    m = R
    v = 0.5 * torch.stack([m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1]], -1)
    s = v.norm(dim=-1)
    c = 0.5 * (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1.0)
    theta = torch.atan2(s, c)
    small = s < 1e-4
    scale = torch.where(small, 1.0 + theta * theta / 6.0, theta / s.clamp_min(1e-12))
    aa = v * scale[:, None]
    near_pi = small & (c < 0)
    # evaluated for every row and selected with where(): a host-side `if near_pi.any()` would synchronise the stream
    d = torch.stack([m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]], -1)
    axis = torch.sqrt(torch.clamp((d + 1.0) / 2.0, min=0.0))          # |w_i| from R = 2 w w^T - I at theta = pi
    k = axis.argmax(-1)
    sign = torch.sign(torch.gather(m, 1, k[:, None, None].expand(-1, 1, 3)).squeeze(1) + 1e-20)   # row k fixes the signs
    axis = axis * sign
    axis = axis / axis.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    aa = torch.where(near_pi[:, None], axis * theta[:, None], aa)
    return aa


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
        if c2w.is_cuda:              # one launch instead of ~25 tiny tensor ops per keyframe (the same expressions)
            c2w = c2w.contiguous()
            check(_lib.load().rfx_rba_set_init_pose(ptr(c2w), int(cam_id), int(self.num_cams), ptr(self.init_r), ptr(self.init_t),
                                               ptr(self.init_c2w), stream_ptr(c2w.device)), "rfx_rba_set_init_pose")
            return
        self.init_c2w[cam_id] = c2w
        self.init_r[cam_id] = rotation_matrix_to_angle_axis(c2w[:3, :3].reshape(1, 3, 3)).reshape(-1)
        self.init_t[cam_id] = c2w[:3, 3]

    def _ids(self, cam_id):
        if not isinstance(cam_id, torch.Tensor):
            cam_id = torch.tensor([[cam_id]], device=self.init_c2w.device)
        if cam_id.device != self.init_c2w.device:
            cam_id = cam_id.to(self.init_c2w.device)
        return cam_id

    def _linears(self):
        return [m for m in self.layers.modules() if isinstance(m, nn.Linear)]

    def forward(self, cam_id):
        if not isinstance(cam_id, torch.Tensor) and cam_id == 0:
            return self.init_c2w[0]
        lin = self._linears()
        if len(lin) != 4 or lin[1].in_features != 256 or not self.init_c2w.is_cuda:
            return self.forward_torch(cam_id)             # other widths / layer counts: plain tensor ops
        idx = self._ids(cam_id).reshape(-1).to(torch.int64).contiguous()
        params = [t for m in lin for t in (m.weight, m.bias)]
        return _RbaFn.apply(self, idx, *params)

    def forward_torch(self, cam_id):
        """the same map as ATen ops (reference rba.py:79-100)."""
        if not isinstance(cam_id, torch.Tensor) and cam_id == 0:
            return self.init_c2w[0]
        cam_id = self._ids(cam_id)
        x = (cam_id.type_as(self.init_c2w) / self.num_cams) * 2 - 1
        idx = cam_id.reshape(-1)
        init_r, init_t = self.init_r[idx], self.init_t[idx]
        out = self.layers(torch.cat([x.reshape(-1, 1), init_r, init_t], dim=-1)) * self.scale
        keep = (idx != 0).to(out.dtype)[:, None]     # camera 0 is the gauge: no correction (reference :90-91)
        out = out * keep
        return make_c2w(out[:, :3] + init_r, out[:, 3:] + init_t)
