"""Mapper: keyframe integration into the global volume + the map / pose optimisation schedule.

Host-side mirror of the reference ``mp_slam/mapper.py:191-950`` (same class, method names,
argument meaning).  The two module-level PyCUDA kernels (:36-188) are replaced by
``rfx_gbv_integrate`` / ``rfx_gbv_clear``; everything the schedule calls on the model is a librfx
kernel.  ``step()`` is the body of the reference's ``run()`` loop (:884-906) so a single process
can drive it; ``run()`` keeps the polling form for a separate tracker thread.
Evaluation / mesh helpers (``calc_2d_metric*``, ``post_process_mesh``, :626-821) are out of scope.
"""
from __future__ import annotations

import random
import time

import numpy as np
import torch

from .. import _lib
from .._lib import _F6, _F9, check, farr, ptr, stream_ptr


class _WorldRaysFn(torch.autograd.Function):
    """rays_o = t[ids], rays_d = R[ids] @ d_cam (reference mapper.py:407-409); the backward adds the
    per-ray gradients per pose with index_add_ instead of autograd's sort-based index_put."""

    @staticmethod
    def forward(ctx, poses_all, ids, d_cam):
        K = poses_all.shape[0]
        ids = torch.remainder(ids, K)                       # -1 = the current frame = last pose
        P = poses_all[ids]
        ctx.save_for_backward(ids, d_cam, P)
        ctx.K = K
        return P[:, :3, 3].contiguous(), torch.einsum("nij,nj->ni", P[:, :3, :3], d_cam)

    @staticmethod
    def backward(ctx, g_o, g_d):
        ids, d_cam, P = ctx.saved_tensors
        g = torch.zeros((ids.shape[0], 4, 4), dtype=g_d.dtype, device=g_d.device)
        g[:, :3, :3] = g_d[:, :, None] * d_cam[:, None, :]
        g[:, :3, 3] = g_o
        gp = torch.zeros((ctx.K, 4, 4), dtype=g_d.dtype, device=g_d.device)
        gp.index_add_(0, ids, g)
        return gp, None, None


class _RayBatchFn(torch.autograd.Function):
    """sampling + gather + posing of one iteration's rays in ONE launch (``rfx_gather_rays``; reference
    mapper.py:394-409), with the pose gradient as one more (``rfx_pose_grad``).  Draws the same indices
    as ``_sample_rays`` + ``_world_rays`` for the same state of python's ``random``."""

    @staticmethod
    def forward(ctx, poses_all, kf, current_rays, n_kf_samples, n_cur, keyframe_every):
        lib = _lib.load()
        dev = poses_all.device
        n = int(n_kf_samples) + int(n_cur)
        P = poses_all.detach().to(torch.float32).contiguous()
        K = P.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        rays_o, rays_d, tgt, d_cam = (torch.empty((n, 3), **f32) for _ in range(4))
        tgt_d = torch.empty((n, 1), **f32)
        pose_idx = torch.empty(n, dtype=torch.int32, device=dev)
        seed_kf, seed_cur = random.getrandbits(64), random.getrandbits(64)        # same draw order as the unfused path
        check(lib.rfx_gather_rays(ptr(kf.rays), kf.num_rays_to_save, len(kf), kf.frame_ids_dev.data_ptr(), int(keyframe_every),
                                  ptr(current_rays), current_rays.shape[0], int(n_kf_samples), int(n_cur), seed_kf, seed_cur,
                                  ptr(P), K, ptr(rays_o), ptr(rays_d), ptr(tgt), ptr(tgt_d), ptr(d_cam), pose_idx.data_ptr(),
                                  stream_ptr(dev)), "rfx_gather_rays")
        ctx.save_for_backward(d_cam, pose_idx)
        ctx.K = K
        ctx.mark_non_differentiable(tgt, tgt_d)
        return rays_o, rays_d, tgt, tgt_d

    @staticmethod
    def backward(ctx, g_o, g_d, _gt, _gtd):
        d_cam, pose_idx = ctx.saved_tensors
        gp = torch.empty((ctx.K, 4, 4), dtype=torch.float32, device=d_cam.device)
        go = g_o.to(torch.float32).contiguous() if g_o is not None else None
        gd = g_d.to(torch.float32).contiguous() if g_d is not None else None
        check(_lib.load().rfx_pose_grad(ptr(go), ptr(gd), ptr(d_cam), pose_idx.data_ptr(), d_cam.shape[0], ctx.K, ptr(gp),
                                        stream_ptr(d_cam.device)), "rfx_pose_grad")
        return gp, None, None, None, None, None


class Mapper:
    def __init__(self, config, SLAM, model) -> None:
        self.config, self.slam, self.model = config, SLAM, model
        self.tracking_idx, self.mapping_idx = SLAM.tracking_idx, SLAM.mapping_idx
        self.mapping_first_frame, self.tracking_stop_flag = SLAM.mapping_first_frame, SLAM.tracking_stop_flag
        self.keyframe = SLAM.keyframeDatabase
        self.map_optimizer, self.rba_optimizer = SLAM.map_optimizer, SLAM.rba_optimizer
        self.device, self.dataset = SLAM.device, SLAM.dataset
        self.est_c2w_data, self.RO_c2w_data, self.est_c2w_data_rel = SLAM.est_c2w_data, SLAM.RO_c2w_data, SLAM.est_c2w_data_rel
        self.update_local_MV = SLAM.update_local_MV
        self.first_BA = True
        self.shard = None     # set by MappingPipeline when the scene is partitioned across GPUs
        self.create_global_volume(config["globalV"]["base_resolution"])

    def create_global_volume(self, base_resolution):
        """reference :213-255."""
        self.vol_dim = np.array([base_resolution] * 3)
        self.map_box = self.config["mapping"]["bound"]
        self.voxel_size = 1.0 / base_resolution
        self.vol_origin = np.array([b[0] for b in self.map_box])
        self.box_length = np.array([b[1] - b[0] for b in self.map_box])
        d = self.dataset
        self.K = np.array([[d.fx, 0, d.cx], [0, d.fy, d.cy], [0, 0, 1]])
        self.trunc_margin = self.config["training"]["c_trunc"]

    def save_ckpt(self, save_path):
        # the collective first: wait_meshes() can raise on rank 0 alone (only rank 0 exports), and a rank that raises before a
        # collective leaves its peers waiting inside it
        self.sync_field()               # (a sharded scene: the table whole on every rank; collective)
        self.wait_meshes()              # a failed in-loop export raises here, like the reference's blocking export would have
        torch.save({"pose": self.est_c2w_data, "pose_rel": self.est_c2w_data_rel, "model": self.model.state_dict()}, save_path)

    # ---- GBV kernels --------------------------------------------------------------------
    def init_mapvolume(self):
        """GBV <- (1,0,0,0) (reference :267-282)."""
        check(_lib.load().rfx_gbv_clear(ptr(self.model.GBV.params), int(np.prod(self.vol_dim)), stream_ptr(self.device)),
              "rfx_gbv_clear")

    def integrate_kf(self, batch, pose, obs_weight=1.0):
        """fuse one RGB-D keyframe into GBV/GBW (reference :823-872)."""
        color_im = batch["rgb"].squeeze().to(self.device).float().contiguous()
        depth_im = batch["depth"].squeeze().to(self.device).float().contiguous()
        im_h, im_w = depth_im.shape
        pose_dev = pose.to(self.device).float().reshape(-1).contiguous()
        box = [v for b in self.map_box for v in b]
        check(_lib.load().rfx_gbv_integrate(ptr(self.model.GBV.params), ptr(self.model.GBW.params), int(self.vol_dim[0]),
                                            farr(_F6, box), farr(_F9, self.K.reshape(-1)), ptr(pose_dev), ptr(color_im),
                                            ptr(depth_im), im_h, im_w, float(self.trunc_margin), float(obs_weight),
                                            stream_ptr(self.device)), "rfx_gbv_integrate")
        if self.shard is not None:
            self.shard.exchange_halo(self.model.GBV.params, self.model.GBW.params, int(self.vol_dim[0]))

    def update_GBV(self, cur_id):
        """reset and re-integrate every keyframe with its current pose (reference :523-534)."""
        with torch.no_grad():
            self.model.GBV.params[:] = 0.0
            self.init_mapvolume()
            self.model.GBW.params[:] = 0.0
        for i in range(0, cur_id, self.config["mapping"]["keyframe_every"]):
            self.integrate_kf(self.dataset[i], self.est_c2w_data[i])

    # ---- schedule -----------------------------------------------------------------------
    def first_frame_mapping(self, batch, n_iters=100):
        """reference :284-364."""
        if batch["frame_id"] != 0:
            raise ValueError("First frame mapping must be the first frame!")
        c2w = batch["c2w"].to(self.device)
        self.init_mapvolume()
        self.integrate_kf(batch, c2w)
        self.est_c2w_data[0] = c2w
        self.est_c2w_data_rel[0] = c2w
        self.model.rba.update_init_pose(0, c2w)
        self.model.train()
        H, W, n_s = self.slam.dataset.H, self.slam.dataset.W, self.config["mapping"]["sample"]
        direction, rgb, depth = (batch[k].to(self.device) for k in ("direction", "rgb", "depth"))
        ret = loss = None
        for _ in range(n_iters):
            self.map_optimizer.zero_grad()
            indice = (_lib.random_subset(H * W, n_s, self.device) if self.keyframe.device_sampling
                      else self.slam.select_samples(H, W, n_s).to(self.device))
            indice_h, indice_w = indice % H, indice // H        # (sic) reference :338
            rays_d_cam = direction[indice_h, indice_w, :]
            target_s = rgb[indice_h, indice_w, :]
            target_d = depth[indice_h, indice_w].unsqueeze(-1)
            rays_o = c2w[None, :3, -1].repeat(n_s, 1)
            rays_d = torch.sum(rays_d_cam[..., None, :] * c2w[:3, :3], -1)
            ret = self.model.mapping(rays_o, rays_d, target_s, target_d)
            loss = self.slam.get_loss_from_ret(ret)
            loss.backward()
            self.map_optimizer.step()
        self.keyframe.add_keyframe(batch, filter_depth=self.config["mapping"]["filter_depth"])
        self.mapping_first_frame[0] = 1
        self._prepare_steps()
        return ret, loss

    def _prepare_steps(self):
        """what the FIRST mapper step would otherwise set up on its way (0.5-0.8 ms of host time with nothing queued on the GPU
        yet -- the step's kernels wait for it): the no-op `model.to(device)` of the loop's first pass (reference :884-890), the
        direct iterations' buffers, descriptors and optimizer state.  Nothing here changes a result."""
        if self.first_BA:
            self.model = self.model.to(self.device)
            self.first_BA = False
        try:
            direct = self._direct_iterations()
        except Exception:      # noqa: BLE001 -- e.g. a scene shard that is attached later: the first step sets it up as before
            direct = None
        if direct is not None and hasattr(direct, "prepare"):
            direct.prepare()

    def _sample_rays(self, current_rays):
        m = self.config["mapping"]
        rays, ids = self.keyframe.sample_global_rays(m["sample"])
        n_cur = max(m["sample"] // len(self.keyframe.frame_ids), m["min_pixels_cur"])
        hw = self.slam.dataset.H * self.slam.dataset.W
        if self.keyframe.device_sampling:
            idx_cur = _lib.random_subset(hw, n_cur, current_rays.device)
        else:
            idx_cur = torch.as_tensor(random.sample(range(0, hw), n_cur), device=current_rays.device)
        cur = current_rays[idx_cur, :]
        rays = torch.cat([rays.to(self.device), cur], dim=0)
        ids_kf = (ids // m["keyframe_every"]).to(device=self.device, dtype=torch.int64)
        ids_all = torch.cat([ids_kf, torch.full((n_cur,), -1, dtype=torch.int64, device=self.device)])
        return rays, ids_all

    @staticmethod
    def _world_rays(rays, ids_all, poses_all):
        rays_o, rays_d = _WorldRaysFn.apply(poses_all, ids_all, rays[..., :3].contiguous())
        return rays_o, rays_d, rays[..., 3:6], rays[..., 6:7]

    def _ray_batch(self, current_rays, poses_all):
        """rays of one iteration: mapping.sample keyframe rays + the current frame's share, in world space."""
        m = self.config["mapping"]
        if not self.keyframe.device_sampling:          # reference sampling (python random.sample on the host)
            rays, ids_all = self._sample_rays(current_rays)
            return self._world_rays(rays, ids_all, poses_all)
        n_cur = max(m["sample"] // len(self.keyframe.frame_ids), m["min_pixels_cur"])
        return _RayBatchFn.apply(poses_all, self.keyframe, current_rays, m["sample"], n_cur, m["keyframe_every"])

    def _current_rays(self, batch):
        """[H*W, 7] rays of the frame (camera direction | rgb | depth); step() builds them once for both phases and the
        keyframe store (batch["_rays7"])."""
        r = batch.get("_rays7")
        if r is None:
            r = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1)
            r = r.reshape(-1, r.shape[-1]).to(self.device)
        return r

    def global_mapping(self, batch, cur_frame_id):
        """map update over all keyframes + the current frame (reference :366-423)."""
        m = self.config["mapping"]
        poses = self.est_c2w_data[0:cur_frame_id + 1:m["keyframe_every"]].clone()
        self.map_optimizer.zero_grad()
        self.rba_optimizer.zero_grad()
        current_rays = self._current_rays(batch)
        with torch.no_grad():
            k_last = cur_frame_id // m["keyframe_every"]
            last_kf_id = self._camera_ids(k_last + 1)[k_last:]          # [[k_last]]: a view, no fill launch
            poses_all = poses
            poses_all[-1, :, :] = self.model.rba(last_kf_id).squeeze()   # (the assignment is the copy)
        direct = self._direct_iterations()
        if direct is not None:          # same kernels and random draws, launched without an autograd graph
            for i in range(m["iters"]):
                direct.map_iteration(current_rays, poses_all)
            return
        for i in range(m["iters"]):
            rays_o, rays_d, target_s, target_d = self._ray_batch(current_rays, poses_all)
            ret = self.model.mapping(rays_o, rays_d, target_s, target_d)
            loss = self.slam.get_loss_from_ret(ret, smooth=True, iter=i)
            loss.backward(retain_graph=True)
            if (i + 1) % m["map_accum_step"] == 0:
                if (i + 1) > m["map_wait_step"]:
                    self.map_optimizer.step()
                self.map_optimizer.zero_grad()
                self.rba_optimizer.zero_grad()

    def global_pose(self, batch, cur_frame_id):
        """pose (RBA-MLP) update with the map frozen (reference :425-520)."""
        m = self.config["mapping"]
        n_kf = len(range(0, cur_frame_id, m["keyframe_every"]))          # poses = est_c2w_data[0:cur_frame_id:keyframe_every]
        frame_ids_all = list(range(0, cur_frame_id + 1, m["keyframe_every"]))
        self.map_optimizer.zero_grad()
        self.rba_optimizer.zero_grad()
        current_rays = self._current_rays(batch)
        all_index = self._camera_ids(n_kf + 1)                          # arange(0, n_kf + 1)[:, None], cached
        direct = self._direct_iterations() if m["opt_pose"] else None
        if direct is not None:
            idx = all_index.reshape(-1).contiguous()
            for i in range(m["BA_iters"]):
                direct.pose_iteration(current_rays, idx)
            with torch.no_grad():
                poses_all = self.model.rba(all_index)
            self._write_back_poses(poses_all, frame_ids_all, cur_frame_id)
            return
        poses_all = self.model.rba(all_index)
        if not m["opt_pose"]:
            # no optimizer consumes pose gradients in this phase (the reference still back-propagates into the
            # RBA MLP, :489-497, and discards the result): cut the graph at the poses
            poses_all = poses_all.detach()
        for i in range(m["BA_iters"]):
            rays_o, rays_d, target_s, target_d = self._ray_batch(current_rays, poses_all)
            ret = self.model.mapping(rays_o, rays_d, target_s, target_d, clamp=True)
            loss = self.slam.get_loss_from_ret(ret, fs=True, smooth=True, iter=i)
            loss.backward(retain_graph=True)
            if (i + 1) % m["pose_accum_step"] == 0 and m["opt_pose"]:
                self.rba_optimizer.step()
                poses_all = self.model.rba(all_index)
                self.map_optimizer.zero_grad()
                self.rba_optimizer.zero_grad()
        self._write_back_poses(poses_all, frame_ids_all, cur_frame_id)

    def _camera_ids(self, n):
        """arange(0, n)[:, None] on the device: a view of one cached tensor (the reference builds it, and the `kfupid` list below,
        with two or three tiny launches per mapper step)"""
        c = getattr(self, "_cam_ids", None)
        if c is None or c.shape[0] < n:
            c = self._cam_ids = torch.arange(0, max(n, 64), device=self.device).unsqueeze(-1)
        return c[:n]

    def _write_back_poses(self, poses_all, frame_ids_all, cur_frame_id):
        """refined keyframe poses -> est_c2w_data (reference :507-520)."""
        m = self.config["mapping"]
        if len(frame_ids_all) > 1 and m["opt_pose"]:
            n = len(frame_ids_all) - 1
            ke = m["keyframe_every"]
            if m["optim_cur"]:
                self.est_c2w_data[cur_frame_id] = poses_all[-1].detach()
            # est_c2w_data[arange(n) * keyframe_every] = poses_all[:-1]: the index list is a stride -- one strided copy
            self.est_c2w_data[0:(n - 1) * ke + 1:ke] = poses_all[:-1].detach()

    def _direct_iterations(self):
        """graph-free issue of the BA iterations (mp_slam/direct.py) when the configuration allows it."""
        if getattr(self, "_direct", None) is None:
            from .direct import DirectIterations
            sh = getattr(self, "scene_shard", None)        # dist.ShardedPipeline: ONE scene over several GPUs
            if sh is not None:
                from .sharded import LevelShardedIterations, ShardedIterations
                if not DirectIterations.supported(self):
                    raise _lib.RfxError("the sharded scene needs the configuration DirectIterations supports "
                                        "(device ray sampling, accumulation steps of 1, TV term on)")
                # mapping.shard_field: "levels" (hash table partitioned by level: per-point rows travel), "replicas" (table
                # replicated, dense gradient all-reduced), "auto" = levels whenever the table has a level per rank
                mode = str(self.config["mapping"].get("shard_field", "auto"))
                if mode not in ("auto", "levels", "replicas"):
                    raise _lib.RfxError(f"mapping.shard_field: {mode!r}")
                if mode == "auto":          # the cheaper of the two by the time model of dist.choose_field_mode
                    from ..dist import choose_field_mode
                    m, tr = self.config["mapping"], self.config["training"]
                    n_rays = int(m["sample"]) + max(int(m["sample"]) // 8, int(m["min_pixels_cur"]))
                    S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
                    self.field_mode_choice = choose_field_mode(self.model.embed_res_fn.desc, n_rays * S, P ** 3, sh.world)
                    mode = self.field_mode_choice["mode"]
                if mode == "levels":
                    self._direct = LevelShardedIterations(self, sh.dist, sh.rank, sh.world)
                else:
                    self._direct = ShardedIterations(self, sh.dist, sh.rank, sh.world)
            else:
                self._direct = DirectIterations(self) if DirectIterations.supported(self) else False
        return self._direct or None

    def convert_relative_pose(self, idx=None):
        """absolute pose per frame: keyframes as stored, others = delta @ keyframe (reference :580-624)."""
        ke = self.config["mapping"]["keyframe_every"]
        n = len(self.est_c2w_data) if idx is None else len(self.est_c2w_data[:idx + 1])
        poses = {}
        for i in range(n):
            if i % ke == 0:
                poses[i] = self.est_c2w_data[i]
            else:
                poses[i] = self.est_c2w_data_rel[i] @ self.est_c2w_data[(i // ke) * ke]
        return poses

    def convert_relative_pose_npy(self, idx=None):
        """reference :536-577."""
        d = self.convert_relative_pose(idx)
        poses = torch.zeros((len(self.dataset), 4, 4), device=self.device)
        for i, p in d.items():
            poses[i] = p
        return poses.detach().cpu().numpy()

    def step(self, current_map_id: int):
        """one pass of the reference's mapping loop body (:884-906) for frame ``current_map_id``."""
        ke = self.config["mapping"]["keyframe_every"]
        if self.first_BA:
            self.model = self.model.to(self.device)
            self.first_BA = False
        batch = self.dataset[current_map_id]
        batch = {k: (v[None, ...] if isinstance(v, torch.Tensor) else torch.tensor([v])) for k, v in batch.items()}
        if int(self.mapping_idx[0]) % ke == 0:
            self.model.rba.update_init_pose(int(current_map_id // ke), self.est_c2w_data[current_map_id])
            self.integrate_kf(batch, self.est_c2w_data[current_map_id])
        batch["_rays7"] = self._current_rays(batch)
        self.global_mapping(batch, current_map_id)
        self.global_pose(batch, current_map_id)
        self.mapping_idx[0] = current_map_id
        if int(self.mapping_idx[0]) % ke == 0:
            self.keyframe.add_keyframe(batch, filter_depth=self.config["mapping"]["filter_depth"])
        self._meshes_in_loop(current_map_id, batch)

    @property
    def last_mesh(self):
        """the latest in-loop mesh (joins an export that is still running)"""
        m = getattr(self, "_last_mesh", None)
        return m.result() if hasattr(m, "result") else m

    def wait_meshes(self):
        """block until the in-loop exports issued so far are on disk"""
        m = getattr(self, "_last_mesh", None)
        if hasattr(m, "wait"):
            m.wait()

    def sync_field(self):
        """collective on a sharded scene (every rank calls it): make this rank's copy of the hash table whole again before
        something reads all of it -- meshing, rendering, a checkpoint.  A no-op elsewhere."""
        d = getattr(self, "_direct", None)
        if d and getattr(d, "stale", False):
            d.sync_table()

    def _meshes_in_loop(self, idx: int, batch):
        """the mesh exports the reference makes inside its loop (:908-918): a mesh per `video.save_freq` frames when a video is
        recorded, and one per `mesh.vis` frames unless mesh.only_final (BASELINE config 5: a marching-cubes mesh per
        keyframe = mesh.vis: keyframe_every, only_final: 0).  On a sharded scene every rank holds the same field: rank 0 writes."""
        cfg = self.config
        sh = getattr(self, "scene_shard", None)
        mesh_now = (cfg["video"]["save"] and idx % cfg["video"]["save_freq"] == 0) or (idx % cfg["mesh"]["vis"] == 0 and not cfg["mesh"]["only_final"])
        if sh is not None:
            if mesh_now:
                self.sync_field()           # every rank: rank 0 is about to read the whole table
            if sh.rank != 0:
                return
        # mesh.async_export (default on): the loop copies the field and goes on; a worker thread sweeps the copy and writes the
        # file (SLAM.save_mesh_async).  Off: the reference's blocking export.
        asyn = bool(cfg["mesh"].get("async_export", True))
        save = self.slam.save_mesh_async if asyn else self.slam.save_mesh
        if cfg["video"]["save"] and idx % cfg["video"]["save_freq"] == 0:
            self._last_mesh = save(idx, voxel_size=0.075)
        if idx % cfg["mesh"]["vis"] == 0:
            if not cfg["mesh"]["only_final"]:
                self._last_mesh = save(idx, voxel_size=0.1)
            # (mesh.render_img / pose_eval_func at this point of the reference are evaluation I/O: out of scope)

    def run(self):
        """polling form of the loop (reference :874-906), for a tracker running in another thread."""
        m = self.config["mapping"]
        while self.tracking_idx[0] < len(self.dataset) - 1:
            while self.tracking_idx[0] <= self.mapping_idx[0] + m["map_every"] and self.tracking_stop_flag[0] == 0:
                time.sleep(0.01)
            current_map_id = int(self.mapping_idx[0] + m["keyframe_every"])
            if current_map_id < len(self.dataset):
                self.step(current_map_id)
            if self.tracking_stop_flag[0] != 0:
                break
        if m["save_ckpt"]:              # (save_ckpt: the collective sync_field(), THEN wait_meshes() -- see there)
            import os
            self.save_ckpt(os.path.join(self.config["data"]["output"], self.config["data"]["exp_name"], "checkpoint.pt"))
        self.wait_meshes()              # the in-loop exports are on disk (or their error is raised) when run() returns
