"""One bundle-adjustment iteration issued straight to librfx, without building an autograd graph.

The loss of an iteration (reference mp_slam/mapper.py:394-420 / :470-505: ray batch -> JointEncoding.mapping ->
get_loss_from_ret(smooth=True) -> backward -> Adam step) is a fixed chain of librfx kernels whose backward is
known in closed form, so forward and backward are launched back to back by ONE library call
(``rfx_ba_forward_backward``, csrc/rfx_ba.hip, which only sequences the public entry points) and the gradients are
handed to the (PyTorch) optimizers here.  Same kernels, same order, same random draws as the autograd formulation in
``Mapper.global_mapping / global_pose``: what is gone is the graph bookkeeping (four Function nodes, the engine's
worker thread, AccumulateGrad) and ~20 foreign calls per iteration; the TV term's hash gradient is scattered
together with the field's (one sweep over the table).  ``tests/test_pose_gpu.py`` checks both formulations give the
same gradients.  The pose phase computes the map gradients that the reference's backward also produces there, and
that no optimizer consumes, only with ``mapping.unused_gradients``.
"""
from __future__ import annotations

import ctypes as C
import random
from types import SimpleNamespace

import torch

from .. import _lib
from .._lib import check, stream_ptr


class _Buffers:
    """device buffers of one iteration, sized for UP TO n rays and K cameras, + their pointers.  The ray count of an
    iteration shrinks as keyframes accumulate (sample // num_kf current-frame rays) and the camera count grows:
    allocating per shape would hit the allocator (hundreds of MB of workspace) on every mapper step."""

    def __init__(self, lib, dev, n, S, P, n_feat, n_levels, table, weights, K, grid_desc=None):
        f32 = dict(dtype=torch.float32, device=dev)
        t = self.t = SimpleNamespace()
        self.cap_n, self.cap_K, self._u_views = n, K, {}
        t.u = torch.empty(n * S, **f32)
        t.u6 = torch.empty(6, **f32)
        t.lc = torch.empty(8, **f32)
        t.tv_acc = torch.empty(1, dtype=torch.float64, device=dev)
        t.dt = torch.empty_like(table)
        t.dw_flat = torch.empty(sum(w.numel() for w in weights), **f32)
        # with the grid known the scatter's share is sized so that all its binned levels (T >= 2^19) are in flight at once
        self.ws_bytes = int(lib.rfx_ba_workspace_bytes_for(n, S, P, C.byref(grid_desc)) if grid_desc is not None
                            else lib.rfx_ba_workspace_bytes(n, S, P, n_feat, n_levels))
        t.ws = torch.empty(self.ws_bytes // 4 + 64, **f32)
        self.dws, off = [], 0
        for w in weights:
            self.dws.append(t.dw_flat[off:off + w.numel()].view_as(w))
            off += w.numel()
        if K:
            t.poses = torch.empty((K, 4, 4), **f32)
            t.acts = torch.empty(int(lib.rfx_rba_acts_floats(K)), **f32)
            t.dposes = torch.empty((K, 4, 4), **f32)
            t.wsr = torch.empty(int(lib.rfx_rba_grads_floats(K)), **f32)
        self.p = SimpleNamespace(**{k: v.data_ptr() for k, v in vars(t).items()})
        self.p.ws = (self.p.ws + 255) // 256 * 256           # the library wants a 256-byte aligned workspace
        self.n, self.S, self.P = n, S, P

    def u_view(self, n):
        """[n, S] prefix of the jitter buffer: uniform_() on it makes the draw torch.rand((n, S)) makes"""
        v = self._u_views.get(n)
        if v is None:
            v = self._u_views[n] = self.t.u[:n * self.S].view(n, self.S)
        return v


class _StageBuffers:
    """buffers of the stage-by-stage issue (every intermediate is a torch tensor the host can look at): views of one
    arena sized for up to cap_n rays / cap_K cameras, re-bound when the shape of the iteration changes."""

    def __init__(self, lib, dev, cap_n, S, P, n_feat, n_levels, table, weights, cap_K, grid_desc=None):
        self.lib, self.dev, self.cap_n, self.cap_K = lib, dev, cap_n, cap_K
        self.grid_desc = grid_desc
        self.S, self.P, self.n_feat, self.n_levels = S, P, n_feat, n_levels
        f32 = dict(dtype=torch.float32, device=dev)
        self.arena = torch.empty(sum(-(-fl // 64) * 64 for _, _, _, fl in self._layout(cap_n, cap_K)), **f32)
        self.dt = torch.empty_like(table)
        self.dw_flat = torch.empty(sum(w.numel() for w in weights), **f32)
        self.ones = torch.ones(1, **f32)
        self.dws, off = [], 0
        for w in weights:
            self.dws.append(self.dw_flat[off:off + w.numel()].view_as(w))
            off += w.numel()
        self.shape = None
        self.bind(cap_n, cap_K)

    def _layout(self, n, K):
        """(name, shape, dtype, fp32 words)"""
        S, nt, F = self.S, self.P ** 3, self.n_feat
        f, i32, f64 = torch.float32, torch.int32, torch.float64
        lay = [("o", (n, 3), f), ("d", (n, 3), f), ("tgt", (n, 3), f), ("d_cam", (n, 3), f), ("td", (n,), f), ("pidx", (n,), i32),
               ("u", (n, S), f), ("z", (n, S), f), ("x01", (n * S, 3), f), ("raw", (n * S, 4), f), ("rgb_map", (n, 3), f),
               ("depth_map", (n,), f), ("sums", (_lib.LOSS_WS_DOUBLES,), f64), ("lc", (8,), f), ("u6", (6,), f), ("pts", (nt, 3), f), ("feat", (nt, F), f),
               ("tv_acc", (1,), f64), ("d_raw", (n * S, 4), f), ("dx", (n * S, 3), f), ("dfeat", (nt, F), f),
               ("ws2", (int(self.lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(self.grid_desc), n * S + nt)
                            if self.grid_desc is not None else
                            self.lib.rfx_grid_encode_backward_workspace_bytes(n * S + nt, self.n_levels)) // 4,), f)]
        if K:
            lay += [("poses", (K, 4, 4), f), ("acts", (int(self.lib.rfx_rba_acts_floats(K)),), f), ("dposes", (K, 4, 4), f),
                    ("wsr", (int(self.lib.rfx_rba_grads_floats(K)),), f)]
        out = []
        for name, shape, dt in lay:
            numel = 1
            for v in shape:
                numel *= v
            out.append((name, shape, dt, numel * (2 if dt == f64 else 1)))
        return out

    def bind(self, n, K):
        if self.shape == (n, K):
            return self
        assert n <= self.cap_n and K <= self.cap_K
        t = self.t = SimpleNamespace()
        off = 0
        for name, shape, dt, fl in self._layout(n, K):
            chunk = self.arena[off:off + fl]
            setattr(t, name, (chunk if dt == torch.float32 else chunk.view(dt)).view(shape))
            off += -(-fl // 64) * 64
        t.dt, t.dw_flat, t.ones = self.dt, self.dw_flat, self.ones
        self.p = SimpleNamespace(**{k: v.data_ptr() for k, v in vars(t).items()})
        self.p.dws = [g.data_ptr() for g in self.dws]
        self.ws2_bytes = t.ws2.numel() * 4
        self.shape = (n, K)
        return self


class DirectIterations:
    def __init__(self, mapper):
        self.mp, self.model, self.slam = mapper, mapper.model, mapper.slam
        self.lib = _lib.load()
        self._cache, self._descs = {}, {}
        self._weights = tuple(self.model.decoder_res.fused_weights())          # the Parameters themselves: stable objects
        self._rba_params = [w for m in self.model.rba._linears() for w in (m.weight, m.bias)]
        self._rba_grads = None
        self.stagewise_every = 0        # bench.py: issue every k-th iteration stage by stage (timed per entry point)
        self.before_stagewise = None    # bench.py: callable run first in such an iteration (waits for the volume's stream, so
                                        # that the per-call timings are not stretched by V1 running beside them)
        # bench.py: callable(phase, n_rays) -> address of a host array of _lib.BA_STAGE_EVENTS event handles, or None.  The
        # one-call iteration records them at its stage boundaries (rfx_ba_desc.stage_events): per-stage device times of the
        # very launches the loop runs, without the stage-by-stage issue's extra calls and un-fused kernels
        self.stage_events = None
        # mapping.unused_gradients: also compute, in the pose phase, the map gradients that no optimizer consumes
        # (what the reference's loss.backward() does); off by default, results are identical either way
        self.unused_gradients = bool(mapper.config["mapping"].get("unused_gradients", False))
        # the TV value itself (SLAM.smoothness's return) only matters through its gradient: evaluated on request
        self.report_tv = bool(mapper.config["mapping"].get("report_tv", False))
        # The sampler jitter (torch.rand((n, S))) and the TV lattice offset (torch.rand(6)) are drawn inside the iteration's
        # first kernel from a seed taken from python's generator (rfx_ba_desc.seed_u), like the ray draws; torch_draws = True
        # takes them from torch's generator in the autograd formulation's order instead (two more launches per iteration).
        self.torch_draws = bool(mapper.config["mapping"].get("torch_draws", False))
        self.last_seed_u = 0            # seed of the latest iteration's own draws (tests reproduce them: oracle/draws_oracle.py)
        self._count = 0
        self.iterations = {"map": 0, "pose": 0}          # issued so far (bench.py: launches per entry point)

    def _stagewise_now(self) -> bool:
        self._count += 1
        return self.stagewise_every > 0 and self._count % self.stagewise_every == 0

    def prepare(self):
        """allocate and fill in, ahead of the first iteration, what the first iteration of each phase would on its way: the
        buffers at their capacity, both phases' descriptors, the pose MLP's gradient buffers, the optimizers' state
        (Mapper._prepare_steps: the first mapper step of a stream then costs the host what every later one does)."""
        enc = self.model.embed_res_fn
        dev = enc.params.device
        if dev.type != "cuda":
            return
        B = self._buffers(1, 1, dev)                      # capacity: most rays an iteration can have, all cameras
        for clamp in (False, True):
            self._descriptor(B, clamp, dev)
        self.model._loss_weights(dev)
        self._rba_grad_buffers()
        for opt in (self.mp.map_optimizer, self.mp.rba_optimizer):
            if hasattr(opt, "prepare"):
                opt.prepare()

    @staticmethod
    def supported(mapper) -> bool:
        m, tr = mapper.config["mapping"], mapper.config["training"]
        lin = mapper.model.rba._linears()
        return (bool(m.get("direct_iterations", True)) and mapper.keyframe.device_sampling and m["map_accum_step"] == 1
                and m["pose_accum_step"] == 1 and m["map_wait_step"] == 0 and tr["smooth_weight"] > 0
                and len(lin) == 4 and lin[1].in_features == 256 and mapper.model.embed_res_fn.desc.n_feat == 2)

    def _buffers(self, n, K, dev):
        tr = self.model.config["training"]
        enc = self.model.embed_res_fn
        S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
        key = (S, P, str(dev), enc.params.data_ptr())
        b = self._cache.get(key)
        if b is None or b.cap_n < n or b.cap_K < K:
            cap_n, cap_K = self._capacity(n, K, b)
            b = self._cache[key] = _Buffers(self.lib, dev, cap_n, S, P, enc.n_output_dims, int(enc.desc.n_levels), enc.params,
                                            self._weights, cap_K, grid_desc=enc.desc)
        b.n = n
        return b

    def _capacity(self, n, K, old):
        """most rays an iteration can have (one keyframe: sample + sample current-frame rays) and all cameras"""
        m = self.mp.config["mapping"]
        return (max(n, int(m["sample"]) + max(int(m["sample"]), int(m["min_pixels_cur"])), old.cap_n if old else 0),
                max(K, int(self.model.rba.num_cams) + 1, old.cap_K if old else 0))

    def _n_rays(self):
        m = self.mp.config["mapping"]
        return int(m["sample"]) + int(max(m["sample"] // len(self.mp.keyframe.frame_ids), m["min_pixels_cur"]))

    def _descriptor(self, B, clamp, dev):
        """rfx_ba_desc of one phase with everything that does not change between iterations filled in; rebuilt when a
        tensor it points to is re-allocated."""
        model, mp = self.model, self.mp
        enc, kf = model.embed_res_fn, mp.keyframe
        ws = self._weights
        key = (id(B), enc.params.data_ptr(), ws[0].data_ptr(), ws[3].data_ptr(), model.GBV.params.data_ptr(), kf.rays.data_ptr(),
               kf.frame_ids_dev.data_ptr())
        c = self._descs.get(clamp)
        if c is not None and c[0] == key:
            return c[1]
        cfg = model.config
        tr, m = cfg["training"], cfg["mapping"]
        d = _lib.BaDesc()
        d.field = self._field(clamp)                    # .staged set: rfx_ba_forward_backward refreshes the image itself
        d.sampler = model._sampler_desc()
        d.bbox, d.bbox_f64 = model._bbox6, model._bbox_f64
        d.sc_factor, d.depth_trunc, d.trunc = float(cfg["data"]["sc_factor"]), float(cfg["cam"]["depth_trunc"]), float(tr["trunc"])
        d.rgb_missing_on = int(tr["rgb_missing"] > 0)
        d.tv_P, d.tv_voxel, d.tv_margin = B.P, float(tr["smooth_vox"]), float(tr["smooth_margin"])
        d.tv_scale = float(tr["smooth_weight"]) / float(int(tr["smooth_pts"]) ** 3)
        d.tv_normalise = 1 if cfg["grid"]["tcnn_encoding"] else 0
        d.kf_rays, d.rays_per_kf = kf.rays.data_ptr(), kf.num_rays_to_save
        d.kf_frame_ids, d.keyframe_every = kf.frame_ids_dev.data_ptr(), int(m["keyframe_every"])
        d.n_kf_samples = int(m["sample"])
        d.u6, d.losses8 = B.p.u6, B.p.lc
        d.hash_entries = enc.params.numel() // int(enc.desc.n_feat)
        self._descs[clamp] = (key, d, C.byref(d))
        self._perturb = tr["perturb"] > 0.0
        return d

    def _field(self, clamp):
        return self.model._field_desc(clamp)

    def _run(self, B, current_rays, poses_ptr, K, clamp, d_poses_ptr, st, map_grads=True, rba=None):
        """fill in what changes per iteration and launch it (forward + backward).  rba = (params, acts, scale, grads, workspace):
        the call carries on into the pose MLP's backward (rfx_ba_desc.rba)."""
        self._fill(B, current_rays, poses_ptr, K, clamp, d_poses_ptr, map_grads, rba)
        check(self.lib.rfx_ba_forward_backward(self._descs[clamp][2], B.p.ws, B.ws_bytes, st), "rfx_ba_forward_backward")

    def _fill(self, B, current_rays, poses_ptr, K, clamp, d_poses_ptr, map_grads=True, rba=None):
        """the iteration's rfx_ba_desc: what changes per iteration filled in (the random draws are taken here)"""
        t, p = B.t, B.p
        n = B.n
        d = self._descriptor(B, clamp, t.u.device)
        d.loss_w_dev = self.model._loss_weights(t.u.device).data_ptr()
        d.num_kf = len(self.mp.keyframe)
        d.cur_rays, d.cur_population = current_rays.data_ptr(), current_rays.shape[0]
        d.n_cur = n - d.n_kf_samples
        d.seed_kf, d.seed_cur = random.getrandbits(64), random.getrandbits(64)     # same draw order as the autograd path
        d.poses16, d.K = poses_ptr, K
        if self.torch_draws:
            d.u_z = None
            if self._perturb:
                B.u_view(n).uniform_()                   # the draw torch.rand((n, S)) makes
                d.u_z = p.u
            t.u6.uniform_()                               # the draw torch.rand(6) makes
            d.u6, d.seed_u = p.u6, 0
        else:
            d.u_z = d.u6 = None
            d.seed_u = self.last_seed_u = random.getrandbits(64) | 1
        d.d_poses16 = d_poses_ptr
        if rba is not None:
            d.rba, d.rba_acts, d.rba_scale, d.rba_grads, d.rba_ws = C.addressof(rba[0]), rba[1], rba[2], C.addressof(rba[3]), rba[4]
        else:
            d.rba = d.rba_acts = d.rba_grads = d.rba_ws = None
        if map_grads:
            d.d_hash, d.d_w, d.tv_sum = p.dt, p.dw_flat, (p.tv_acc if self.report_tv else None)
        else:                           # pose phase: only the ray/pose gradients (the u6 draw above keeps the random stream)
            d.d_hash = d.d_w = d.tv_sum = None
        d.stage_events = self.stage_events("pose" if d_poses_ptr else "map", n) if self.stage_events is not None else None
        return d

    # ------------------------------------------------------------------ stage-by-stage issue (instrumentation / cross-check)
    # The same iteration as rfx_ba_forward_backward, one foreign call per stage, so that bench.py can put HIP events
    # around the individual entry points and tests can compare the two.  Used for every `stagewise_every`-th iteration.
    def _rays(self, B, current_rays, poses_ptr, K, st):
        mp, lib, m = self.mp, self.lib, self.mp.config["mapping"]
        kf = mp.keyframe
        n_cur = B.t.o.shape[0] - int(m["sample"])
        seed_kf, seed_cur = random.getrandbits(64), random.getrandbits(64)
        p = B.p
        check(lib.rfx_gather_rays(kf.rays.data_ptr(), kf.num_rays_to_save, len(kf), kf.frame_ids_dev.data_ptr(), int(m["keyframe_every"]),
                                  current_rays.data_ptr(), current_rays.shape[0], int(m["sample"]), n_cur, seed_kf, seed_cur,
                                  poses_ptr, K, p.o, p.d, p.tgt, p.td, p.d_cam, p.pidx, st), "rfx_gather_rays")


    def _forward_backward(self, B, S, P, clamp, want_ray_grads, st, map_grads=True):
        """mapping objective + TV term on the rays in B: forward, then backward into B.dt / B.dws (and, for
        want_ray_grads, d rays_o / d rays_d, returned)."""
        lib, model = self.lib, self.model
        cfg = model.config
        tr = cfg["training"]
        t, p = B.t, B.p
        n = t.o.shape[0]
        dev = t.o.device
        # ---- forward (== _MappingFn.forward)
        u_ptr = None
        seed_u = self.last_seed_u = 0 if self.torch_draws else random.getrandbits(64) | 1      # same place in python's stream as in _run
        if tr["perturb"] > 0.0:
            if self.torch_draws:
                t.u.uniform_()                           # the draw torch.rand((n, S)) makes
            else:
                check(lib.rfx_uniform_draws(seed_u, 0, n * S, p.u, st), "rfx_uniform_draws")
            u_ptr = p.u
        sd = model._sampler_desc()
        check(lib.rfx_sample_z(C.byref(sd), p.td, u_ptr, n, p.z, st), "rfx_sample_z")
        check(lib.rfx_ray_points(p.o, p.d, p.z, n, S, model._bbox6, model._bbox_f64, p.x01, st), "rfx_ray_points")
        desc = model._field_desc(clamp)
        dref = C.byref(desc)
        # the forward leaves its hash features in the backward workspace; the chain below reads them instead of looking
        # the table up again (nothing else touches the workspace or the table in between)
        ws = model._workspace(lib.rfx_field_backward_workspace_bytes(n * S), dev)
        wsp, wb = ws.data_ptr(), ws.numel() * 4
        check(lib.rfx_field_forward_stash(dref, p.x01, n * S, p.raw, wsp, wb, st), "rfx_field_forward_stash")
        trunc, sc = float(tr["trunc"]), float(cfg["data"]["sc_factor"])
        check(lib.rfx_composite_forward(p.raw, p.z, n, S, trunc, sc, p.rgb_map, p.depth_map, None, st), "rfx_composite_forward")
        depth_trunc, rgb_on = float(cfg["cam"]["depth_trunc"]), int(tr["rgb_missing"] > 0)
        check(lib.rfx_mapping_loss_forward(p.raw, p.z, p.rgb_map, p.depth_map, p.tgt, p.td, n, S, trunc * sc, depth_trunc, rgb_on,
                                           p.sums, p.lc, p.lc + 16, st), "rfx_mapping_loss_forward")
        # ---- TV term forward (== SLAM.smoothness / _SmoothFn.forward)
        enc = model.embed_res_fn
        table_ptr = enc.params.data_ptr()
        if self.torch_draws:
            t.u6.uniform_()                               # the draw torch.rand(6) makes
        else:
            check(lib.rfx_uniform_draws(seed_u, 1, 6, p.u6, st), "rfx_uniform_draws")
        n_tv = P * P * P
        if map_grads:
            check(lib.rfx_tv_lattice(p.u6, P, float(tr["smooth_vox"]), float(tr["smooth_margin"]), model._bbox6, model._bbox_f64,
                                     1 if cfg["grid"]["tcnn_encoding"] else 0, p.pts, st), "rfx_tv_lattice")
            check(lib.rfx_grid_encode_forward(enc.desc, table_ptr, p.pts, n_tv, p.feat, st), "rfx_grid_encode_forward")
            if self.report_tv:
                check(lib.rfx_tv_forward(p.feat, P, enc.n_output_dims, p.tv_acc, st), "rfx_tv_forward")
        # ---- backward: d(total)/d(loss_i) = training weights; d(total)/d(TV) = smooth_weight
        wvec = model._loss_weights(dev)
        check(lib.rfx_mapping_loss_backward(p.raw, p.z, p.rgb_map, p.depth_map, p.tgt, p.td, n, S, trunc, sc, trunc * sc, depth_trunc,
                                            rgb_on, p.lc + 16, wvec.data_ptr(), None, None, p.d_raw, st), "rfx_mapping_loss_backward")
        chain = (lib.rfx_field_backward_chain_inputs_stashed if not map_grads else
                 lib.rfx_field_backward_chain_stashed if want_ray_grads else lib.rfx_field_backward_chain_weights_stashed)
        check(chain(dref, p.x01, n * S, p.d_raw, wsp, wb, st), "rfx_field_backward_chain")
        if map_grads:
            t.dt.zero_()
            t.dw_flat.zero_()
            dws = p.dws
            check(lib.rfx_field_backward_weights(n * S, p.d_raw, dws[0], dws[1], dws[2], dws[3], wsp, wb, st), "rfx_field_backward_weights")
        go = gd = None
        if want_ray_grads:
            check(lib.rfx_field_backward_scatter(dref, p.x01, n * S, None, p.dx, wsp, wb, st), "rfx_field_backward_scatter")
            check(lib.rfx_field_backward_dx(dref, p.x01, n * S, p.d_raw, p.dx, wsp, wb, st), "rfx_field_backward_dx")
            dp = t.dx.view(n, S, 3) / model._extent_on(dev)
            go, gd = dp.sum(1), (dp * t.z[..., None]).sum(1)
        if not map_grads:
            return go, gd
        # TV backward; its hash gradient and the field's are scattered by ONE sweep over the table segments
        scale = float(tr["smooth_weight"]) / float(int(tr["smooth_pts"]) ** 3)
        check(lib.rfx_tv_backward(p.feat, P, enc.n_output_dims, scale, p.ones, p.dfeat, st), "rfx_tv_backward")
        check(lib.rfx_field_backward_scatter_merged(dref, p.x01, n * S, p.pts, p.dfeat, n_tv, p.dt, wsp, wb, p.ws2, B.ws2_bytes, st),
              "rfx_field_backward_scatter_merged")
        return go, gd


    def _run_stagewise(self, current_rays, poses_ptr, K, clamp, want_pose_grads, dev, st, map_grads=True):
        tr = self.model.config["training"]
        enc = self.model.embed_res_fn
        n = self._n_rays()
        S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
        key = ("stage", S, P, str(dev), enc.params.data_ptr())
        B = self._cache.get(key)
        if B is None or B.cap_n < n or B.cap_K < K:
            cap_n, cap_K = self._capacity(n, K, B)
            B = self._cache[key] = _StageBuffers(self.lib, dev, cap_n, S, P, enc.n_output_dims, int(enc.desc.n_levels), enc.params,
                                                 self._weights, cap_K, grid_desc=enc.desc)
        B.bind(n, K)
        if self.before_stagewise is not None:
            self.before_stagewise()
        self._rays(B, current_rays, poses_ptr, K, st)
        go, gd = self._forward_backward(B, S, P, clamp, want_pose_grads, st, map_grads)
        return B, go, gd

    def _set_map_grads(self, B):
        self.model.embed_res_fn.params.grad = B.t.dt
        for prm, g in zip(self._weights, B.dws):
            prm.grad = g

    # ------------------------------------------------------------------ the two phases
    def map_gradients(self, current_rays, poses_all):
        """forward + backward of one global_mapping iteration: leaves the gradients in .grad."""
        Pm = poses_all.detach().to(torch.float32).contiguous()
        dev = Pm.device
        if self._stagewise_now():
            B, _, _ = self._run_stagewise(current_rays, Pm.data_ptr(), Pm.shape[0], False, False, dev, stream_ptr(dev))
        else:
            B = self._buffers(self._n_rays(), 0, dev)
            self._run(B, current_rays, Pm.data_ptr(), Pm.shape[0], False, None, stream_ptr(dev))
        self._set_map_grads(B)
        return B.t.lc

    def map_iteration(self, current_rays, poses_all):
        """one trip of the loop of Mapper.global_mapping (map parameters step; poses fixed)."""
        lc = self.map_gradients(current_rays, poses_all)
        self.iterations["map"] += 1
        self.mp.map_optimizer.step()
        self._drop_grads()
        return lc

    def pose_iteration(self, current_rays, idx):
        """one trip of the loop of Mapper.global_pose with opt_pose: poses = RBA(idx), pose-MLP step."""
        lc = self.pose_gradients(current_rays, idx, map_grads=self.unused_gradients)
        self.iterations["pose"] += 1
        self.mp.rba_optimizer.step()
        self._drop_grads()
        return lc

    def _rba_grad_buffers(self):
        params = self._rba_params
        if self._rba_grads is None or self._rba_grads[0] != params[0].data_ptr():
            grads = [torch.empty_like(w) for w in params]      # overwritten by every rfx_rba_backward
            self._rba_grads = (params[0].data_ptr(), grads, _lib.RbaParams(*[w.data_ptr() for w in params], 256),
                               _lib.RbaGrads(*[g.data_ptr() for g in grads]))
        return self._rba_grads

    def _drop_grads(self):
        """map_optimizer.zero_grad() + rba_optimizer.zero_grad() (set_to_none) for the parameters these iterations touch"""
        self.model.embed_res_fn.params.grad = None
        for w in self._weights:
            w.grad = None
        for w in self._rba_params:
            w.grad = None

    def pose_gradients(self, current_rays, idx, map_grads=True):
        """forward + backward of one global_pose iteration: pose-MLP (and, with map_grads, map) gradients in .grad.

        The reference's loss.backward() also fills the map parameters' .grad in this phase, but only the pose
        optimizer steps before both are zeroed (mp_slam/mapper.py:494-499): with map_grads=False those stages
        (weight gradients, table scatter, TV term) are not run; poses and parameters come out the same."""
        lib, rba = self.lib, self.model.rba
        dev = idx.device
        K = idx.shape[0]
        st = stream_ptr(dev)
        R = self._buffers(self._n_rays(), K, dev)     # owns the RBA buffers in both modes
        p = R.p
        params = self._rba_params
        _, grads, prm, gdesc = self._rba_grad_buffers()
        check(lib.rfx_rba_forward(C.byref(prm), rba.init_r.data_ptr(), rba.init_t.data_ptr(), idx.data_ptr(), K, rba.num_cams,
                                  float(rba.scale), p.poses, p.acts, st), "rfx_rba_forward")
        if self._stagewise_now():
            B, go, gd = self._run_stagewise(current_rays, p.poses, K, True, True, dev, st, map_grads)
            check(lib.rfx_pose_grad(go.data_ptr(), gd.data_ptr(), B.p.d_cam, B.p.pidx, B.t.o.shape[0], K, p.dposes, st), "rfx_pose_grad")
            check(lib.rfx_rba_backward(C.byref(prm), p.acts, K, p.dposes, float(rba.scale), C.byref(gdesc), p.wsr, st), "rfx_rba_backward")
        else:                                        # ... and on into the pose MLP's backward, inside the same call
            B = R
            self._run(B, current_rays, p.poses, K, True, p.dposes, st, map_grads, rba=(prm, p.acts, float(rba.scale), gdesc, p.wsr))
        if map_grads:
            self._set_map_grads(B)                   # produced by the reference's backward too; no optimizer consumes them
        for w, g in zip(params, grads):
            w.grad = g
        return B.t.lc
