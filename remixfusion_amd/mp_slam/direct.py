"""One bundle-adjustment iteration issued straight to librfx, without building an autograd graph.

The loss of an iteration (reference mp_slam/mapper.py:394-420 / :470-505: ray batch -> JointEncoding.mapping ->
get_loss_from_ret(smooth=True) -> backward -> Adam step) is a fixed chain of librfx kernels whose backward is
known in closed form, so the host can launch forward and backward back to back and hand the gradients to the
(PyTorch) optimizers itself.  Same kernels, same order, same random draws as the autograd formulation in
``Mapper.global_mapping / global_pose`` -- only the graph bookkeeping (four Function nodes, the engine's worker
thread, AccumulateGrad) is gone, and the TV term's hash gradient is accumulated into the same buffer as the
field's instead of a second one.  ``tests/test_pose_gpu.py`` checks both formulations give the same updates.
Nothing is pruned: the pose phase still produces the (unused) map gradients the reference's backward produces.
"""
from __future__ import annotations

import ctypes as C
import random

import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


class DirectIterations:
    def __init__(self, mapper):
        self.mp, self.model, self.slam = mapper, mapper.model, mapper.slam
        self.lib = _lib.load()

    @staticmethod
    def supported(mapper) -> bool:
        m, tr = mapper.config["mapping"], mapper.config["training"]
        lin = mapper.model.rba._linears()
        return (bool(m.get("direct_iterations", True)) and mapper.keyframe.device_sampling and m["map_accum_step"] == 1
                and m["pose_accum_step"] == 1 and m["map_wait_step"] == 0 and tr["smooth_weight"] > 0
                and len(lin) == 4 and lin[1].in_features == 256 and mapper.model.embed_res_fn.desc.n_feat == 2)

    # ------------------------------------------------------------------ pieces
    def _rays(self, current_rays, poses):
        mp, lib, m = self.mp, self.lib, self.mp.config["mapping"]
        kf = mp.keyframe
        dev = poses.device
        n_cur = max(m["sample"] // len(kf.frame_ids), m["min_pixels_cur"])
        n = int(m["sample"]) + int(n_cur)
        f32 = dict(dtype=torch.float32, device=dev)
        o, d, tgt, d_cam = (torch.empty((n, 3), **f32) for _ in range(4))
        td = torch.empty(n, **f32)
        pidx = torch.empty(n, dtype=torch.int32, device=dev)
        seed_kf, seed_cur = random.getrandbits(64), random.getrandbits(64)
        check(lib.rfx_gather_rays(ptr(kf.rays), kf.num_rays_to_save, len(kf), kf.frame_ids_dev.data_ptr(), int(m["keyframe_every"]),
                                  ptr(current_rays), current_rays.shape[0], int(m["sample"]), int(n_cur), seed_kf, seed_cur,
                                  ptr(poses), poses.shape[0], ptr(o), ptr(d), ptr(tgt), ptr(td), ptr(d_cam), pidx.data_ptr(),
                                  stream_ptr(dev)), "rfx_gather_rays")
        return o, d, tgt, td, d_cam, pidx

    def _forward_backward(self, o, d, tgt, td, clamp, want_ray_grads):
        """mapping objective + TV term: forward, then backward into (d_hash, dW1..4[, d rays_o, d rays_d])."""
        lib, model, slam = self.lib, self.model, self.slam
        cfg = model.config
        tr = cfg["training"]
        dev = o.device
        st = stream_ptr(dev)
        n = o.shape[0]
        S = int(tr["n_range_d"]) + int(tr["n_samples_d"])
        f32 = dict(dtype=torch.float32, device=dev)
        # ---- forward (== _MappingFn.forward)
        u = torch.rand((n, S), **f32) if tr["perturb"] > 0.0 else None
        z = torch.empty((n, S), **f32)
        sd = model._sampler_desc()
        check(lib.rfx_sample_z(C.byref(sd), ptr(td), ptr(u), n, ptr(z), st), "rfx_sample_z")
        x01 = torch.empty((n * S, 3), **f32)
        check(lib.rfx_ray_points(ptr(o), ptr(d), ptr(z), n, S, model._bbox6, model._bbox_f64, ptr(x01), st), "rfx_ray_points")
        raw = torch.empty((n * S, 4), **f32)
        desc = model._field_desc(clamp)
        check(lib.rfx_field_forward(C.byref(desc), ptr(x01), n * S, ptr(raw), st), "rfx_field_forward")
        rgb_map, depth_map = torch.empty((n, 3), **f32), torch.empty(n, **f32)
        trunc, sc = float(tr["trunc"]), float(cfg["data"]["sc_factor"])
        check(lib.rfx_composite_forward(ptr(raw), ptr(z), n, S, trunc, sc, ptr(rgb_map), ptr(depth_map), None, st),
              "rfx_composite_forward")
        sums = torch.empty(8, dtype=torch.float64, device=dev)
        lc = torch.empty(8, **f32)                       # losses[4] | coef[4]
        depth_trunc, rgb_on = float(cfg["cam"]["depth_trunc"]), int(tr["rgb_missing"] > 0)
        check(lib.rfx_mapping_loss_forward(ptr(raw), ptr(z), ptr(rgb_map), ptr(depth_map), ptr(tgt), ptr(td), n, S, trunc * sc,
                                           depth_trunc, rgb_on, sums.data_ptr(), lc.data_ptr(), lc.data_ptr() + 16, st),
              "rfx_mapping_loss_forward")
        # ---- TV term forward (== SLAM.smoothness / _SmoothFn.forward)
        enc = model.embed_res_fn
        table = enc.params
        P = int(tr["smooth_pts"]) - 1
        u6 = torch.rand(6, device=dev)
        pts = torch.empty((P * P * P, 3), **f32)
        check(lib.rfx_tv_lattice(ptr(u6), P, float(tr["smooth_vox"]), float(tr["smooth_margin"]), model._bbox6, model._bbox_f64,
                                 1 if cfg["grid"]["tcnn_encoding"] else 0, ptr(pts), st), "rfx_tv_lattice")
        feat = torch.empty((pts.shape[0], enc.n_output_dims), **f32)
        check(lib.rfx_grid_encode_forward(enc.desc, ptr(table), ptr(pts), pts.shape[0], ptr(feat), st), "rfx_grid_encode_forward")
        tv_acc = torch.empty(1, dtype=torch.float64, device=dev)
        check(lib.rfx_tv_forward(ptr(feat), P, enc.n_output_dims, tv_acc.data_ptr(), st), "rfx_tv_forward")
        # ---- backward: d(total)/d(loss_i) = training weights; d(total)/d(TV) = smooth_weight
        wvec = model._loss_weights(dev)
        d_raw = torch.empty_like(raw)
        check(lib.rfx_mapping_loss_backward(ptr(raw), ptr(z), ptr(rgb_map), ptr(depth_map), ptr(tgt), ptr(td), n, S, trunc, sc,
                                            trunc * sc, depth_trunc, rgb_on, lc.data_ptr() + 16, ptr(wvec), None, None, ptr(d_raw), st),
              "rfx_mapping_loss_backward")
        dt = torch.zeros_like(table)
        w = model.decoder_res.fused_weights()
        flat = torch.zeros(sum(t.numel() for t in w), **f32)          # one fill for the four weight gradients
        dws, off = [], 0
        for t in w:
            dws.append(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        dx = torch.empty_like(x01) if want_ray_grads else None
        ws = model._workspace(lib.rfx_field_backward_workspace_bytes(n * S), dev)
        wb = ws.numel() * 4
        check(lib.rfx_field_backward_chain(C.byref(desc), ptr(x01), n * S, ptr(d_raw), ptr(ws), wb, st), "rfx_field_backward_chain")
        check(lib.rfx_field_backward_weights(n * S, ptr(d_raw), ptr(dws[0]), ptr(dws[1]), ptr(dws[2]), ptr(dws[3]), ptr(ws), wb, st),
              "rfx_field_backward_weights")
        go = gd = None
        if want_ray_grads:
            check(lib.rfx_field_backward_scatter(C.byref(desc), ptr(x01), n * S, None, ptr(dx), ptr(ws), wb, st),
                  "rfx_field_backward_scatter")
            check(lib.rfx_field_backward_dx(C.byref(desc), ptr(x01), n * S, ptr(d_raw), ptr(dx), ptr(ws), wb, st), "rfx_field_backward_dx")
            dp = dx.view(n, S, 3) / model._extent_on(dev)
            go, gd = dp.sum(1), (dp * z[..., None]).sum(1)
        # TV backward; its hash gradient and the field's are scattered by ONE sweep over the table segments
        gs = getattr(self, "_tv_gscale", None)
        if gs is None or gs.device != dev:
            gs = self._tv_gscale = torch.ones(1, **f32)
        dfeat = torch.empty_like(feat)
        scale = float(tr["smooth_weight"]) / float(int(tr["smooth_pts"]) ** 3)
        check(lib.rfx_tv_backward(ptr(feat), P, enc.n_output_dims, scale, ptr(gs), ptr(dfeat), st), "rfx_tv_backward")
        nb = int(lib.rfx_grid_encode_backward_workspace_bytes(n * S + pts.shape[0], int(enc.desc.n_levels)))
        ws2 = torch.empty(nb // 4, **f32)
        check(lib.rfx_field_backward_scatter_merged(C.byref(desc), ptr(x01), n * S, ptr(pts), ptr(dfeat), pts.shape[0], ptr(dt),
                                                    ptr(ws), wb, ptr(ws2), ws2.numel() * 4, st), "rfx_field_backward_scatter_merged")
        return dt, dws, go, gd, lc

    def _set_map_grads(self, dt, dws):
        self.model.embed_res_fn.params.grad = dt
        for p, g in zip(self.model.decoder_res.fused_weights(), dws):
            p.grad = g

    # ------------------------------------------------------------------ the two phases
    def map_gradients(self, current_rays, poses_all):
        """forward + backward of one global_mapping iteration: leaves the gradients in .grad."""
        P = poses_all.detach().to(torch.float32).contiguous()
        o, d, tgt, td, _, _ = self._rays(current_rays, P)
        dt, dws, _, _, lc = self._forward_backward(o, d, tgt, td, False, False)
        self._set_map_grads(dt, dws)
        return lc

    def map_iteration(self, current_rays, poses_all):
        """one trip of the loop of Mapper.global_mapping (map parameters step; poses fixed)."""
        lc = self.map_gradients(current_rays, poses_all)
        self.mp.map_optimizer.step()
        self.mp.map_optimizer.zero_grad()
        self.mp.rba_optimizer.zero_grad()
        return lc

    def pose_iteration(self, current_rays, idx):
        """one trip of the loop of Mapper.global_pose with opt_pose: poses = RBA(idx), pose-MLP step."""
        lc = self.pose_gradients(current_rays, idx)
        self.mp.rba_optimizer.step()
        self.mp.map_optimizer.zero_grad()
        self.mp.rba_optimizer.zero_grad()
        return lc

    def pose_gradients(self, current_rays, idx):
        """forward + backward of one global_pose iteration: pose-MLP (and map) gradients in .grad."""
        lib, rba = self.lib, self.model.rba
        dev = idx.device
        K = idx.shape[0]
        params = [t for m in rba._linears() for t in (m.weight, m.bias)]
        prm = _lib.RbaParams(*[ptr(p.detach()) for p in params], 256)
        poses = torch.empty((K, 4, 4), dtype=torch.float32, device=dev)
        acts = torch.empty(int(lib.rfx_rba_acts_floats(K)), dtype=torch.float32, device=dev)
        st = stream_ptr(dev)
        check(lib.rfx_rba_forward(C.byref(prm), ptr(rba.init_r), ptr(rba.init_t), idx.data_ptr(), K, rba.num_cams, float(rba.scale),
                                  ptr(poses), ptr(acts), st), "rfx_rba_forward")
        o, d, tgt, td, d_cam, pidx = self._rays(current_rays, poses)
        dt, dws, go, gd, lc = self._forward_backward(o, d, tgt, td, True, True)
        self._set_map_grads(dt, dws)                 # produced by the reference's backward too; no optimizer consumes them
        dposes = torch.empty((K, 4, 4), dtype=torch.float32, device=dev)
        check(lib.rfx_pose_grad(ptr(go.contiguous()), ptr(gd.contiguous()), ptr(d_cam), pidx.data_ptr(), o.shape[0], K, ptr(dposes), st),
              "rfx_pose_grad")
        grads = [torch.empty_like(p) for p in params]
        gdesc = _lib.RbaGrads(*[ptr(g) for g in grads])
        wsr = torch.empty(int(lib.rfx_rba_grads_floats(K)), dtype=torch.float32, device=dev)
        check(lib.rfx_rba_backward(C.byref(prm), ptr(acts), K, ptr(dposes), float(rba.scale), C.byref(gdesc), ptr(wsr), st),
              "rfx_rba_backward")
        for p, g in zip(params, grads):
            p.grad = g
        return lc
