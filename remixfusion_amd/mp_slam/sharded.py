"""Bundle-adjustment iterations of ONE scene on several GPUs.

Two ways of spreading the residual field, chosen by ``mapping.shard_field`` ("auto" = levels):

``LevelShardedIterations`` (round 4, the default): the hash table is PARTITIONED BY LEVEL.  Rank q keeps a contiguous range
of the 16 levels -- it alone looks them up, accumulates their gradient, evaluates their share of the TV term and takes their
Adam step -- for every sample point of the iteration; each rank runs the decoder on a contiguous share of the rays.  What
travels is per-point rows: 8 B per point and level of features before the decoder (all-to-all), the same of feature
gradients after it (all-to-all), 21 KB of decoder gradients and 64 B of loss sums (all-reduce); in the pose phase also 12 B
per point of d loss / d x and the pose gradients.  No table-sized buffer is ever exchanged, and the scatter -- the largest
stage of an iteration at T >= 2^19 -- is divided over the ranks instead of repeated on each.  The iteration is issued as four
library calls (rfx_ba_shard_lookup / _render / _scatter / _pose: the fused launches of the single-GPU call) with the
collectives between them.

``ShardedIterations`` (round 3, ``mapping.shard_field: replicas``): the table replicated, its dense gradient all-reduced.

Below: the replicated form.
The residual field, the decoder and the global explicit volume are replicated; an iteration's ray batch is the SAME
on every rank (same seeds, same device random draws, so it is the batch the single-GPU run would take) and rank r
renders the rays r, r + world, r + 2 world, ...  Two exchanges make the step identical to the single-GPU one:

  * the loss of the reference weighs its free-space / sdf terms by sample counts of the whole batch
    (model/utils.py:170-198), so the eight loss sums are all-reduced (64 bytes) between the forward and the backward
    (``rfx_mapping_loss_sums`` -> all-reduce -> ``rfx_mapping_loss_finalize``);
  * the gradients -- hash table (6.6 MB at T = 2^16 ... 166 MB at 2^21), decoder weights (21 KB), pose gradients --
    are all-reduced (sum) before the optimizers step; every rank then takes the same Adam step on the same values, so the
    replicas stay bit-identical without ever broadcasting parameters.  The TV term has no rays: rank 0 evaluates it.

Same kernels and entry points as ``DirectIterations``' stage-by-stage issue; the reference has no counterpart (single
GPU, mp_slam/mapper.py:366-520 is the loop this implements).
"""
from __future__ import annotations

import ctypes as C

import random

import torch

from .. import _lib
from .._lib import check, stream_ptr
from ..dist import (all_reduce_sum_, all_reduce_sum_start, all_to_all_rows_, all_to_all_rows_start, broadcast_, level_exchange_splits,
                    level_partition)
from .direct import DirectIterations, _StageBuffers


class ShardedIterations(DirectIterations):
    def __init__(self, mapper, dist, rank: int, world: int):
        super().__init__(mapper)
        self.dist, self.rank, self.world = dist, int(rank), int(world)
        self._loc = None
        self._tot8 = None

    def _stagewise_now(self) -> bool:
        self._count += 1
        return True                     # every iteration is issued stage by stage (the exchanges sit between stages)

    def _local_buffers(self, n_loc, K, dev, S, P):
        enc = self.model.embed_res_fn
        B = self._loc
        if B is None or B.cap_n < n_loc or B.cap_K < K or B.dt.data_ptr() == 0:
            cap_n, cap_K = self._capacity(n_loc, K, B)
            B = self._loc = _StageBuffers(self.lib, dev, cap_n, S, P, enc.n_output_dims, int(enc.desc.n_levels), enc.params,
                                          self._weights, cap_K, grid_desc=enc.desc)
        return B.bind(n_loc, K)

    def _run_stagewise(self, current_rays, poses_ptr, K, clamp, want_pose_grads, dev, st, map_grads=True):
        lib, model = self.lib, self.model
        cfg = model.config
        tr = cfg["training"]
        enc = model.embed_res_fn
        n = self._n_rays()
        S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
        key = ("stage", S, P, str(dev), enc.params.data_ptr())
        G = self._cache.get(key)                      # the whole batch (ray setup only)
        if G is None or G.cap_n < n or G.cap_K < K:
            cap_n, cap_K = self._capacity(n, K, G)
            G = self._cache[key] = _StageBuffers(lib, dev, cap_n, S, P, enc.n_output_dims, int(enc.desc.n_levels), enc.params,
                                                 self._weights, cap_K, grid_desc=enc.desc)
        G.bind(n, K)
        self._rays(G, current_rays, poses_ptr, K, st)           # same seeds on every rank: the same n rays
        if self.torch_draws:
            if tr["perturb"] > 0.0:
                G.t.u.uniform_()                                  # the draw torch.rand((n, S)) makes, on every rank
            G.t.u6.uniform_()                                     # the draw torch.rand(6) makes
        else:                                                     # the draws DirectIterations._run lets the library make
            seed_u = random.getrandbits(64) | 1
            if tr["perturb"] > 0.0:
                check(lib.rfx_uniform_draws(seed_u, 0, n * S, G.p.u, st), "rfx_uniform_draws")
            check(lib.rfx_uniform_draws(seed_u, 1, 6, G.p.u6, st), "rfx_uniform_draws")
        # ---- this rank's share: rays rank, rank + world, ...
        sel = torch.arange(self.rank, n, self.world, device=dev)
        n_loc = int(sel.numel())
        B = self._local_buffers(max(n_loc, 1), K, dev, S, P)
        t, p = B.t, B.p
        if n_loc:
            for name in ("o", "d", "tgt", "d_cam", "td", "pidx", "u"):
                getattr(t, name)[:n_loc].copy_(getattr(G.t, name).index_select(0, sel))
        t.u6.copy_(G.t.u6)
        # ---- forward on the share
        trunc, sc = float(tr["trunc"]), float(cfg["data"]["sc_factor"])
        depth_trunc, rgb_on = float(cfg["cam"]["depth_trunc"]), int(tr["rgb_missing"] > 0)
        desc = model._field_desc(clamp)
        dref = C.byref(desc)
        if n_loc:
            sd = model._sampler_desc()
            check(lib.rfx_sample_z(C.byref(sd), p.td, p.u if tr["perturb"] > 0.0 else None, n_loc, p.z, st), "rfx_sample_z")
            check(lib.rfx_ray_points(p.o, p.d, p.z, n_loc, S, model._bbox6, model._bbox_f64, p.x01, st), "rfx_ray_points")
            ws = model._workspace(lib.rfx_field_backward_workspace_bytes(n_loc * S), dev)
            wsp, wb = ws.data_ptr(), ws.numel() * 4
            check(lib.rfx_field_forward_stash(dref, p.x01, n_loc * S, p.raw, wsp, wb, st), "rfx_field_forward_stash")   # see direct.py
            check(lib.rfx_composite_forward(p.raw, p.z, n_loc, S, trunc, sc, p.rgb_map, p.depth_map, None, st), "rfx_composite_forward")
        # ---- loss of the WHOLE batch: local sums -> all-reduce -> finalize
        if self._tot8 is None or self._tot8.device != dev:
            self._tot8 = torch.zeros(8, dtype=torch.float64, device=dev)
        total8 = self._tot8
        check(lib.rfx_mapping_loss_sums(p.raw, p.z, p.rgb_map, p.depth_map, p.tgt, p.td, n_loc, S, trunc * sc, depth_trunc, rgb_on,
                                        p.sums, total8.data_ptr(), st), "rfx_mapping_loss_sums")
        all_reduce_sum_(self.dist, [total8])
        check(lib.rfx_mapping_loss_finalize(total8.data_ptr(), n, S, p.lc, p.lc + 16, st), "rfx_mapping_loss_finalize")
        # ---- TV term (rank 0)
        table_ptr = enc.params.data_ptr()
        n_tv = P * P * P
        tv_here = map_grads and self.rank == 0
        if tv_here:
            check(lib.rfx_tv_lattice(p.u6, P, float(tr["smooth_vox"]), float(tr["smooth_margin"]), model._bbox6, model._bbox_f64,
                                     1 if cfg["grid"]["tcnn_encoding"] else 0, p.pts, st), "rfx_tv_lattice")
            check(lib.rfx_grid_encode_forward(enc.desc, table_ptr, p.pts, n_tv, p.feat, st), "rfx_grid_encode_forward")
        # ---- backward on the share
        go = gd = None
        if map_grads:
            t.dt.zero_()
            t.dw_flat.zero_()
        if n_loc:
            wvec = model._loss_weights(dev)
            check(lib.rfx_mapping_loss_backward(p.raw, p.z, p.rgb_map, p.depth_map, p.tgt, p.td, n_loc, S, trunc, sc, trunc * sc,
                                                depth_trunc, rgb_on, p.lc + 16, wvec.data_ptr(), None, None, p.d_raw, st),
                  "rfx_mapping_loss_backward")
            chain = (lib.rfx_field_backward_chain_inputs_stashed if not map_grads else
                     lib.rfx_field_backward_chain_stashed if want_pose_grads else lib.rfx_field_backward_chain_weights_stashed)
            check(chain(dref, p.x01, n_loc * S, p.d_raw, wsp, wb, st), "rfx_field_backward_chain")
            if map_grads:
                dws = p.dws
                check(lib.rfx_field_backward_weights(n_loc * S, p.d_raw, dws[0], dws[1], dws[2], dws[3], wsp, wb, st),
                      "rfx_field_backward_weights")
            if want_pose_grads:
                check(lib.rfx_field_backward_scatter(dref, p.x01, n_loc * S, None, p.dx, wsp, wb, st), "rfx_field_backward_scatter")
                check(lib.rfx_field_backward_dx(dref, p.x01, n_loc * S, p.d_raw, p.dx, wsp, wb, st), "rfx_field_backward_dx")
                dp = t.dx[:n_loc * S].view(n_loc, S, 3) / model._extent_on(dev)
                go, gd = dp.sum(1), (dp * t.z[:n_loc][..., None]).sum(1)
        else:
            ws = model._workspace(lib.rfx_field_backward_workspace_bytes(1), dev)
            wsp, wb = ws.data_ptr(), ws.numel() * 4
        if map_grads:
            if tv_here:
                scale = float(tr["smooth_weight"]) / float(int(tr["smooth_pts"]) ** 3)
                check(lib.rfx_tv_backward(p.feat, P, enc.n_output_dims, scale, p.ones, p.dfeat, st), "rfx_tv_backward")
            check(lib.rfx_field_backward_scatter_merged(dref, p.x01, n_loc * S, p.pts if tv_here else None, p.dfeat if tv_here else None,
                                                        n_tv if tv_here else 0, p.dt, wsp, wb, p.ws2, B.ws2_bytes, st),
                  "rfx_field_backward_scatter_merged")
            all_reduce_sum_(self.dist, [t.dt, t.dw_flat])      # the step every rank takes is the whole batch's
        self._n_loc = n_loc
        return B, go, gd

    def pose_gradients(self, current_rays, idx, map_grads=True):
        """as DirectIterations.pose_gradients, with the pose gradients summed over the ranks before the pose MLP's backward"""
        lib, rba = self.lib, self.model.rba
        dev = idx.device
        K = idx.shape[0]
        st = stream_ptr(dev)
        R = self._buffers(self._n_rays(), K, dev)     # owns the RBA buffers
        p = R.p
        params = self._rba_params
        if self._rba_grads is None or self._rba_grads[0] != params[0].data_ptr():
            grads = [torch.empty_like(w) for w in params]
            self._rba_grads = (params[0].data_ptr(), grads, _lib.RbaParams(*[w.data_ptr() for w in params], 256),
                               _lib.RbaGrads(*[g.data_ptr() for g in grads]))
        _, grads, prm, gdesc = self._rba_grads
        check(lib.rfx_rba_forward(C.byref(prm), rba.init_r.data_ptr(), rba.init_t.data_ptr(), idx.data_ptr(), K, rba.num_cams,
                                  float(rba.scale), p.poses, p.acts, st), "rfx_rba_forward")
        self._count += 1
        B, go, gd = self._run_stagewise(current_rays, p.poses, K, True, True, dev, st, map_grads)
        R.t.dposes.zero_()
        if self._n_loc:
            check(lib.rfx_pose_grad(go.data_ptr(), gd.data_ptr(), B.p.d_cam, B.p.pidx, self._n_loc, K, p.dposes, st), "rfx_pose_grad")
        all_reduce_sum_(self.dist, [R.t.dposes[:K]])
        if map_grads:
            self._set_map_grads(B)
        check(lib.rfx_rba_backward(C.byref(prm), p.acts, K, p.dposes, float(rba.scale), C.byref(gdesc), p.wsr, st), "rfx_rba_backward")
        for w, g in zip(params, grads):
            w.grad = g
        return B.t.lc


class LevelShardedIterations(DirectIterations):
    """one scene on `world` GPUs with the hash table partitioned by level (module docstring; include/rfx.h: rfx_ba_shard).
    The reference has no counterpart (single GPU); the iteration reproduced is mp_slam/mapper.py:392-423 / :470-505."""

    def __init__(self, mapper, dist, rank: int, world: int):
        super().__init__(mapper)
        self.dist, self.rank, self.world = dist, int(rank), int(world)
        enc = self.model.embed_res_fn
        self.cuts = level_partition(enc.desc, self.world)
        F = int(enc.desc.n_feat)
        lo_l, hi_l = self.cuts[self.rank], self.cuts[self.rank + 1]
        self.k_own = hi_l - lo_l
        # the own levels' part of the flat table (and of its gradient, and of the Adam state): one contiguous range
        self.slices = [(int(enc.desc.offset[self.cuts[q]]) * F,
                        (int(enc.desc.offset[self.cuts[q + 1] - 1]) + int(enc.desc.size[self.cuts[q + 1] - 1])) * F) for q in range(self.world)]
        self.own_slice = self.slices[self.rank]
        opt = self.mp.map_optimizer
        if not hasattr(opt, "slices"):
            raise _lib.RfxError("the level-partitioned field needs remixfusion_amd.optim.Adam (it steps the own levels only)")
        opt.slices[enc.params] = self.own_slice
        self._x = None                  # exchange buffers
        self._shard = None
        self.stale = False              # other ranks' levels of the local table copy are out of date
        self.exchanged_bytes = 0        # received by this rank, summed over the iterations issued
        self.last_exchange = {}

    def _stagewise_now(self) -> bool:
        self._count += 1
        return False                    # always the four fused phases

    def _field(self, clamp):
        return self.model._field_desc(clamp, partitioned_ok=True)

    # -- exchange buffers and the rfx_ba_shard descriptor
    def _exchange(self, B, n, S, dev):
        X = self._x
        cap = B.cap_n
        if X is None or X["cap"] < cap or X["S"] != S:
            f32 = dict(dtype=torch.float32, device=dev)
            m_cap = cap // self.world + 1
            X = self._x = {"cap": cap, "S": S,
                           "feat_send": torch.empty(cap * S * 2 * self.k_own, **f32), "feat_recv": torch.empty(m_cap * S * 32, **f32),
                           "demb_send": torch.empty(m_cap * S * 32, **f32), "demb_recv": torch.empty(cap * S * 2 * self.k_own, **f32),
                           "dx_send": torch.empty(cap * S * 3, **f32), "dx_recv": torch.empty(self.world * m_cap * S * 3, **f32),
                           "sums8": torch.zeros(8, dtype=torch.float64, device=dev), "n": None}
            sh = self._shard = _lib.BaShard()
            sh.rank, sh.world = self.rank, self.world
            for q in range(self.world + 1):
                sh.level_start[q] = self.cuts[q]
            for k in ("feat_send", "feat_recv", "demb_send", "demb_recv", "dx_send", "dx_recv"):
                setattr(sh, k, X[k].data_ptr())
            sh.loss_sums8 = X["sums8"].data_ptr()
        if X["n"] != n:
            sp = level_exchange_splits(n, S, self.cuts, self.rank)      # element counts of the all-to-alls: (to rank q, from rank q)
            for q in range(self.world + 1):
                self._shard.ray_start[q] = sp["rays"][q]
            X.update(n=n, m=sp["m"], feat=sp["feat"], demb=sp["demb"], dx=sp["dx"])
        return X

    def _run(self, B, current_rays, poses_ptr, K, clamp, d_poses_ptr, st, map_grads=True, rba=None):
        lib, dist = self.lib, self.dist
        self._fill(B, current_rays, poses_ptr, K, clamp, d_poses_ptr, map_grads, None)      # (the pose MLP's backward follows the all-reduce)
        d, dref = self._descs[clamp][1], self._descs[clamp][2]
        n, S = B.n, B.S
        X = self._exchange(B, n, S, B.t.u.device)
        sref = C.byref(self._shard)
        ws, wb = B.p.ws, B.ws_bytes
        # Issue order (round 6): what an exchange does not need is launched while it is in flight.  Under RCCL a collective
        # runs on the process group's stream; the compute stream only waits where it reads what was exchanged.
        #   ray batch + own levels' features  ->  feature all-to-all  ||  TV lattice, its lookups, zero-fill of the own gradient
        check(lib.rfx_ba_shard_lookup_rays(dref, sref, ws, wb, st), "rfx_ba_shard_lookup_rays")
        feat_done = all_to_all_rows_start(dist, X["feat_recv"], X["feat"][1], X["feat_send"], X["feat"][0])
        check(lib.rfx_ba_shard_lookup_tv(dref, sref, ws, wb, st), "rfx_ba_shard_lookup_tv")
        feat_done()
        check(lib.rfx_ba_shard_render(dref, sref, ws, wb, st), "rfx_ba_shard_render")
        #   gradient rows all-to-all, then the all-reduce of the decoder gradients / loss sums (21 KB: needed by the optimizer
        #   step and the loss report only)  ||  the table scatter, which waits for the rows alone
        demb_done = all_to_all_rows_start(dist, X["demb_recv"], X["demb"][1], X["demb_send"], X["demb"][0])
        small = [X["sums8"]]
        if map_grads:
            small.append(B.t.dw_flat)
            if self.report_tv:
                small.append(B.t.tv_acc)
        small_done = all_reduce_sum_start(dist, small)
        demb_done()
        check(lib.rfx_ba_shard_scatter(dref, sref, ws, wb, st), "rfx_ba_shard_scatter")
        recv = 4 * (sum(X["feat"][1]) - X["feat"][1][self.rank] + sum(X["demb"][1]) - X["demb"][1][self.rank])      # bytes from OTHER ranks
        if d_poses_ptr:
            all_to_all_rows_(dist, X["dx_recv"], X["dx"][1], X["dx_send"], X["dx"][0])
            check(lib.rfx_ba_shard_pose(dref, sref, ws, wb, st), "rfx_ba_shard_pose")
            all_reduce_sum_(dist, [B.t.dposes[:K]])
            recv += 4 * (sum(X["dx"][1]) - X["dx"][1][self.rank])
            if rba is not None:
                check(lib.rfx_rba_backward(C.byref(rba[0]), rba[1], K, d_poses_ptr, rba[2], C.byref(rba[3]), rba[4], st), "rfx_rba_backward")
        small_done()            # (the sums are whole from here on: the optimizer step follows this call)
        # the four losses of the WHOLE batch (and the coefficients the backward used, re-derived from the summed counts)
        check(lib.rfx_mapping_loss_finalize(X["sums8"].data_ptr(), n, S, B.p.lc, B.p.lc + 16, st), "rfx_mapping_loss_finalize")
        self.last_exchange = {"recv_bytes": recv, "points": n * S, "rays_own": X["m"]}
        self.exchanged_bytes += recv
        if map_grads:
            self.stale = True
            self.model.embed_res_fn.partition_stale = True

    # -- the whole table on every rank again (meshing, rendering, checkpoints: whatever reads all 16 levels of one point)
    def sync_table(self, with_optimizer_state: bool = False):
        enc = self.model.embed_res_fn
        with torch.no_grad():
            tensors = [enc.params.data]
            if with_optimizer_state:
                stt = self.mp.map_optimizer.state.get(enc.params, {})
                tensors += [stt[k] for k in ("exp_avg", "exp_avg_sq") if k in stt]
            for t in tensors:
                for q, (lo, hi) in enumerate(self.slices):
                    broadcast_(self.dist, t[lo:hi], q)
        self.stale = False
        enc.partition_stale = False
