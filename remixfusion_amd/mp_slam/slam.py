"""SLAM: shared pose tensors, optimizers, loss weighting and the TV smoothness term -- the caller
contract the Mapper reads (reference mp_slam/slam.py:23-286).  Orchestration only; mesh export,
pose evaluation and image dumps of the reference (:288-527, open3d / trimesh / matplotlib) are
out of scope."""
from __future__ import annotations

import os
import random
from typing import Dict

import ctypes as C

import numpy as np
import torch
import torch.optim as optim

from .. import _lib
from .._lib import check, ptr, stream_ptr
from ..model.keyframe import KeyFrameDatabase


class _SmoothFn(torch.autograd.Function):
    """TV of the hash features on a lattice as one autograd node: grid lookup + TV reduction forward,
    TV gradient + hash scatter backward (reference mp_slam/slam.py:209-215)."""

    @staticmethod
    def forward(ctx, table, pts01, enc, P, denom, weight=1.0):
        lib = _lib.load()
        denom = float(denom) / float(weight)          # weight * TV / denom, formed inside the node
        x = pts01.detach().reshape(-1, 3).to(torch.float32).contiguous()
        n = x.shape[0]
        feat = torch.empty((n, enc.n_output_dims), dtype=torch.float32, device=x.device)
        st = stream_ptr(x.device)
        check(lib.rfx_grid_encode_forward(enc.desc, ptr(table), ptr(x), n, ptr(feat), st), "rfx_grid_encode_forward")
        acc = torch.empty(1, dtype=torch.float64, device=x.device)
        check(lib.rfx_tv_forward(ptr(feat), P, enc.n_output_dims, acc.data_ptr(), st), "rfx_tv_forward")
        ctx.save_for_backward(table, x, feat)
        ctx.enc, ctx.P, ctx.denom = enc, P, denom
        return (acc[0] / denom).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        table, x, feat = ctx.saved_tensors
        st = stream_ptr(x.device)
        dfeat = torch.empty_like(feat)
        gs = g.reshape(1).to(torch.float32).contiguous()
        check(lib.rfx_tv_backward(ptr(feat), ctx.P, ctx.enc.n_output_dims, 1.0 / ctx.denom, ptr(gs), ptr(dfeat), st), "rfx_tv_backward")
        dt = torch.zeros_like(table)
        nb = int(lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(ctx.enc.desc), x.shape[0]))
        ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)      # staging of the LDS-privatised scatter
        check(lib.rfx_grid_encode_backward(ctx.enc.desc, ptr(table), ptr(x), x.shape[0], ptr(dfeat), ptr(dt), None,
                                           ptr(ws), ws.numel() * 4, st), "rfx_grid_encode_backward")
        return dt, None, None, None, None, None


class SLAM:
    def __init__(self, config, dataset, model, device):
        self.config, self.device, self.dataset, self.model = config, device, dataset, model
        self.create_bounds()
        self.create_share_data()
        self.keyframeDatabase = self.create_kf_database(config)
        self.create_optimizer()
        self.vis_dir = os.path.join(config["data"]["output"], config["data"]["exp_name"])
        self._tv_coords = {}

    # ---- shared state (reference :48-54, :80-90).  One process here, so plain device tensors.
    def create_share_data(self):
        self.create_pose_data()
        self.mapping_first_frame = torch.zeros((1)).int()
        self.mapping_idx = torch.zeros((1))
        self.tracking_idx = torch.zeros((1))
        self.tracking_stop_flag = torch.zeros((1)).int()
        self.update_local_MV = torch.zeros((1))

    def seed_everything(self, seed):
        random.seed(seed)
        os.environ["PYTHONHASHSEED"] = str(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)

    def create_pose_data(self):
        n = self.dataset.num_frames
        self.est_c2w_data = torch.zeros((n, 4, 4), device=self.device)
        self.est_c2w_data_rel = torch.zeros((n, 4, 4), device=self.device)
        self.RO_c2w_data = torch.zeros((n, 4, 4), device=self.device)
        self.load_gt_pose()

    def create_bounds(self):
        self.bounding_box = torch.from_numpy(np.array(self.config["mapping"]["bound"])).to(self.device)
        self.marching_cube_bound = torch.from_numpy(np.array(self.config["mapping"]["marching_cubes_bound"])).to(self.device)

    def create_kf_database(self, config):
        num_kf = int(self.dataset.num_frames // config["mapping"]["keyframe_every"] + 1)
        return KeyFrameDatabase(config, self.dataset.H, self.dataset.W, num_kf, self.dataset.num_rays_to_save,
                                self.device, len(self.dataset))

    def load_gt_pose(self):
        self.pose_gt = torch.zeros((self.dataset.num_frames, 4, 4))
        for i, pose in enumerate(self.dataset.poses):
            self.pose_gt[i] = pose

    def save_state_dict(self, save_path):
        torch.save(self.model.state_dict(), save_path)

    def load(self, load_path):
        self.model.load_state_dict(torch.load(load_path))

    def load_ckpt(self, load_path):
        """checkpoint layout of Mapper.save_ckpt (reference :128-135)."""
        d = torch.load(load_path)
        self.model.load_state_dict(d["model"])
        self.est_c2w_data = d["pose"]
        self.est_c2w_data_rel = d["pose_rel"]

    # ---- sampling / losses
    def select_samples(self, H, W, samples):
        return torch.tensor(random.sample(range(H * W), int(samples)))

    def get_loss_from_ret(self, ret, rgb=True, sdf=True, depth=True, fs=True, smooth=False, tracking=False, iter=0):
        """weighted sum of the four mapping losses (+ smooth_weight * TV) (reference :145-190)."""
        tr = self.config["training"]
        if rgb and depth and sdf and fs and "loss_weighted" in ret:
            loss = ret["loss_weighted"]             # the same weighted sum, formed inside the fused mapping node
        else:
            loss = 0
            if rgb:
                loss += tr["rgb_weight"] * ret["rgb_res_loss"]
            if depth:
                loss += tr["depth_weight"] * ret["depth_res_loss"]
            if sdf:
                loss += tr["sdf_weight"] * ret["sdf_res_loss"]
            if fs:
                loss += tr["fs_weight"] * ret["fs_res_loss"]
        if smooth and tr["smooth_weight"] > 0:
            loss = loss + self.smoothness(tr["smooth_pts"], tr["smooth_vox"], margin=tr["smooth_margin"],
                                          weight=tr["smooth_weight"])
        return loss

    def smoothness(self, sample_points=256, voxel_size=0.1, margin=0.05, color=False, weight=None):
        """Total variation of the raw hash features on a randomly placed lattice (reference :193-217);
        ``weight`` (optional) scales the result inside the autograd node.  The lattice is laid out by
        ``rfx_tv_lattice`` from six device uniforms (the reference builds it on the CPU every call)."""
        P = sample_points - 1
        dev = self.device
        u6 = torch.rand(6, device=dev)
        pts = torch.empty((P * P * P, 3), dtype=torch.float32, device=dev)
        m = self.model
        check(_lib.load().rfx_tv_lattice(ptr(u6), P, float(voxel_size), float(margin), m._bbox6, m._bbox_f64,
                                         1 if self.config["grid"]["tcnn_encoding"] else 0, ptr(pts), stream_ptr(dev)),
              "rfx_tv_lattice")
        return _SmoothFn.apply(m.embed_res_fn.params, pts, m.embed_res_fn, P, float(sample_points ** 3),
                               1.0 if weight is None else float(weight))

    def smoothness_points_torch(self, u6, sample_points, voxel_size, margin):
        """the lattice of the reference as tensor ops (:198-207), from the same six uniforms (tests)."""
        bb = self.bounding_box
        volume = bb[:, 1] - bb[:, 0]
        grid_size = (sample_points - 1) * voxel_size
        offset_max = bb[:, 1] - bb[:, 0] - grid_size - 2 * margin
        # torch.rand is fp32 and then cast like the reference's `.to(offset_max)` / `.to(volume)`: with an
        # all-integer mapping.bound `volume` is int64 and the lattice jitter truncates to 0 (reference quirk)
        offset = u6[:3].to(offset_max.dtype) * offset_max + margin
        P = sample_points - 1
        ar = torch.arange(0, P, dtype=torch.long, device=self.device)
        coords = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), dim=-1).float().to(volume)
        pts = (coords + u6[3:].reshape(1, 1, 1, 3).to(volume.dtype)) * voxel_size + bb[:, 0] + offset
        return (pts - bb[:, 0]) / (bb[:, 1] - bb[:, 0]) if self.config["grid"]["tcnn_encoding"] else pts

    def smoothness_unfused(self, pts_tcnn, sample_points):
        """reference formulation on top of query_sdf_res(embed=True) (kept for tests)."""
        sdf_res = self.model.query_sdf_res(pts_tcnn, embed=True)
        tv_x = torch.pow(sdf_res[1:, ...] - sdf_res[:-1, ...], 2).sum()
        tv_y = torch.pow(sdf_res[:, 1:, ...] - sdf_res[:, :-1, ...], 2).sum()
        tv_z = torch.pow(sdf_res[:, :, 1:, ...] - sdf_res[:, :, :-1, ...], 2).sum()
        return (tv_x + tv_y + tv_z) / (sample_points ** 3)

    def get_rays_from_batch(self, batch, c2w_est, indices):
        """reference :219-247."""
        rays_d_cam = batch["direction"].reshape(-1, 3)[indices].to(self.device)
        target_s = batch["rgb"].reshape(-1, 3)[indices].to(self.device)
        target_d = batch["depth"].reshape(-1, 1)[indices].to(self.device)
        rays_d = torch.sum(rays_d_cam[..., None, :] * c2w_est[:3, :3], -1)
        rays_o = c2w_est[None, :3, -1].repeat(rays_d.shape[0], 1)
        return rays_o, rays_d, target_s, target_d, batch["c2w"][0].to(self.device)

    def convert_relative_pose(self):
        """reference :258-269."""
        ke = self.config["mapping"]["keyframe_every"]
        poses = {}
        for i in range(len(self.est_c2w_data)):
            if i % ke == 0:
                poses[i] = self.est_c2w_data[i]
            else:
                poses[i] = self.est_c2w_data_rel[i] @ self.est_c2w_data[(i // ke) * ke]
        return poses

    def create_optimizer(self):
        """two Adams: map (decoder wd 1e-6, hash eps 1e-15) and pose MLP (reference :271-286)."""
        m = self.config["mapping"]
        trainable = [
            {"params": list(self.model.decoder_res.parameters()), "weight_decay": 1e-6, "lr": m["lr_decoder"]},
            {"params": list(self.model.embed_res_fn.parameters()), "eps": 1e-15, "lr": m["lr_embed_res"]},
        ]
        rba = [{"params": list(self.model.rba.parameters()), "weight_decay": 1e-6, "eps": 1e-15, "lr": m["lr_pose"]}]
        # same Adam as the reference (torch.optim.Adam, betas (0.9, 0.99)); on the device its step is one librfx launch
        # over all tensors of the optimizer (remixfusion_amd/optim.py); mapping.fused_adam=False keeps torch's own step
        if self.config["mapping"].get("fused_adam", True) and all(p.is_cuda for g in trainable + rba for p in g["params"]):
            from ..optim import Adam
            self.map_optimizer = Adam(trainable, betas=(0.9, 0.99))
            self.rba_optimizer = Adam(rba, betas=(0.9, 0.99))
        else:
            self.map_optimizer = optim.Adam(trainable, betas=(0.9, 0.99))
            self.rba_optimizer = optim.Adam(rba, betas=(0.9, 0.99))

    @torch.no_grad()
    # ---- mesh export (reference slam.py:348-414; marching cubes runs on the device, see mesh.py)
    def _mesh_path(self, name):
        return os.path.join(self.config["data"]["output"], self.config["data"]["exp_name"], name)

    def _save_mesh(self, path, sdf_fn, color_fn, voxel_size):
        from ..mesh import extract_mesh, write_ply
        mesh = extract_mesh(sdf_fn, self.model.query_w_res, self.config, self.bounding_box, color_func=color_fn,
                            marching_cube_bound=self.marching_cube_bound, voxel_size=voxel_size)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        write_ply(path, mesh)
        return mesh

    def save_mesh_async(self, i, voxel_size=0.05):
        """save_mesh without blocking the caller: the field is copied now, swept and written by a worker thread on its own
        stream (mesh.AsyncMeshExporter); the mesh is that of the field at the moment of the call.  Returns the exporter
        (``.result()`` joins)."""
        from ..mesh import AsyncMeshExporter
        ex = getattr(self, "_mesh_exporter", None)
        if ex is None:
            ex = self._mesh_exporter = AsyncMeshExporter(self.model, self.config, self.bounding_box, self.marching_cube_bound)
        ex.submit(self._mesh_path("mesh_track{}.ply".format(int(i))), voxel_size)
        return ex

    def save_mesh(self, i, voxel_size=0.05):
        return self._save_mesh(self._mesh_path("mesh_track{}.ply".format(int(i))), self.model.query_sdf_res,
                               self.model.query_color_residual, voxel_size)

    def save_mesh_final(self, voxel_size=0.05):
        return self._save_mesh(self._mesh_path("mesh.ply"), self.model.query_sdf_res, self.model.query_color_residual,
                               voxel_size)

    def save_mesh_explicit(self, i, voxel_size=0.05):
        return self._save_mesh(self._mesh_path("mesh_track{}_ex.ply".format(int(i))), self.model.query_sdf_ex,
                               self.model.query_color_ex, voxel_size)

    def render_single(self, frame_id, gt_depth, gt_color, cam_pose, ray_d, prefix=None, gap=1):
        """Full-frame rgb/depth prediction (reference :288-344) through the fused renderer."""
        gt_color = gt_color.squeeze(0)[::gap, ::gap, :]
        gt_depth = gt_depth.squeeze(0)[::gap, ::gap]
        ray_d = ray_d.squeeze()[::gap, ::gap, ...].to(self.device)
        c2w = cam_pose.squeeze().detach().to(self.device) if isinstance(cam_pose, torch.Tensor) \
            else torch.from_numpy(cam_pose).to(self.device)
        target_d = gt_depth.reshape(-1, 1).to(self.device)
        rays_d = torch.sum(ray_d.reshape(-1, 3).unsqueeze(1) * c2w[None, :3, :3], -1).reshape(-1, 3)
        rays_o = c2w[:3, -1].repeat(rays_d.shape[0], 1)
        rgb, depth = self.model.render_fused(rays_o, rays_d, target_d)
        h, w = gt_depth.shape[0], gt_depth.shape[1]
        return rgb.reshape(h, w, 3), depth.reshape(h, w)
