"""JointEncoding: the mixed scene representation (explicit GBV + residual neural field) on MI355X.

Host-side mirror of the reference ``model/scene_rep.py:13-529``: same constructor, attributes
(``decoder_res``, ``embed_res_fn``, ``embedpos_fn``, ``GBV``, ``GBW``, ``rba``, ``bounding_box``,
``config``, ``clamp``) and methods (``mapping``, ``render_rays``, ``run_network``,
``query_color_sdf``, ``query_sdf_res``, ``query_w_res``, ``query_color_residual``, ``query_sdf_ex``,
``query_color_ex``, ``sdf2weights``, ``raw2outputs``).  Every tensor op on the hot path is a librfx
HIP kernel wrapped in a ``torch.autograd.Function``; PyTorch provides memory, streams, autograd
plumbing and the Adam optimizers.  There is no CPU fallback.

Not built (dead code in the reference): ``query_sdf``,
``query_color_sdf_tracking``; ``pcwrite`` is CPU debug I/O.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import FieldDesc, SamplerDesc, _D6, check, farr, ptr, stream_ptr
from .decoder import ColorSDFNet
from .encodings import DenseGrid, get_encoder
from .rba import RBA
from .utils import batchify, compute_loss, get_sdf_loss


# ------------------------------------------------------------------------------ autograd glue
def _field_backward_staged(lib, desc, x, n, draw, dt, dws, dx, ws, st):
    """the four stages of rfx_field_backward as separate calls (identical kernels; separately timeable)."""
    wb = ws.numel() * 4
    check(lib.rfx_field_backward_chain(C.byref(desc), ptr(x), n, ptr(draw), ptr(ws), wb, st), "rfx_field_backward_chain")
    check(lib.rfx_field_backward_weights(n, ptr(draw), ptr(dws[0]), ptr(dws[1]), ptr(dws[2]), ptr(dws[3]), ptr(ws), wb, st),
          "rfx_field_backward_weights")
    check(lib.rfx_field_backward_scatter(C.byref(desc), ptr(x), n, ptr(dt), ptr(dx), ptr(ws), wb, st), "rfx_field_backward_scatter")
    check(lib.rfx_field_backward_dx(C.byref(desc), ptr(x), n, ptr(draw), ptr(dx), ptr(ws), wb, st), "rfx_field_backward_dx")


class _FieldFn(torch.autograd.Function):
    """raw4 = Q1(x01) (scene_rep.py:314-349) with grads for hash table, MLP weights and x01."""

    @staticmethod
    def forward(ctx, x01, table, w1, w2, w3, w4, model, clamp):
        lib = _lib.load()
        x = x01.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        raw = torch.empty((n, 4), dtype=torch.float32, device=x.device)
        desc = model._field_desc(clamp)
        check(lib.rfx_field_forward(C.byref(desc), ptr(x), n, ptr(raw), stream_ptr(x.device)), "rfx_field_forward")
        ctx.save_for_backward(x, table, w1, w2, w3, w4)
        ctx.model, ctx.clamp = model, clamp
        return raw

    @staticmethod
    def backward(ctx, draw):
        lib = _lib.load()
        x, table, w1, w2, w3, w4 = ctx.saved_tensors
        model = ctx.model
        n = x.shape[0]
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        dt = torch.zeros_like(table) if need[1] else None
        dws = [torch.zeros_like(w) if nd else None for w, nd in zip((w1, w2, w3, w4), need[2:6])]
        ws = model._workspace(model._backward_workspace_bytes(n), x.device)
        desc = model._field_desc(ctx.clamp)
        _field_backward_staged(lib, desc, x, n, draw.contiguous(), dt, dws, dx, ws, stream_ptr(x.device))
        return dx, dt, dws[0], dws[1], dws[2], dws[3], None, None


class _RayPointsFn(torch.autograd.Function):
    """x01 = ((o + d z) - bb_min)/(bb_max - bb_min) (scene_rep.py:443,:388)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, model):
        lib = _lib.load()
        o, d, z = (t.detach().to(torch.float32).contiguous() for t in (rays_o, rays_d, z_vals))
        n, S = z.shape
        x01 = torch.empty((n * S, 3), dtype=torch.float32, device=z.device)
        check(lib.rfx_ray_points(ptr(o), ptr(d), ptr(z), n, S, model._bbox6, model._bbox_f64, ptr(x01),
                                 stream_ptr(z.device)), "rfx_ray_points")
        ctx.save_for_backward(z)
        ctx.model = model
        return x01

    @staticmethod
    def backward(ctx, dx01):
        (z,) = ctx.saved_tensors
        n, S = z.shape
        dp = dx01.view(n, S, 3) / ctx.model._extent_on(dx01.device)
        go = dp.sum(1) if ctx.needs_input_grad[0] else None
        gd = (dp * z[..., None]).sum(1) if ctx.needs_input_grad[1] else None
        return go, gd, None, None


class _CompositeFn(torch.autograd.Function):
    """raw2outputs (scene_rep.py:156-179): (raw [n,S,4], z [n,S]) -> rgb [n,3], depth [n]."""

    @staticmethod
    def forward(ctx, raw, z_vals, trunc, sc_factor):
        lib = _lib.load()
        raw_c, z = raw.detach().contiguous(), z_vals.detach().contiguous()
        n, S = z.shape
        rgb = torch.empty((n, 3), dtype=torch.float32, device=z.device)
        depth = torch.empty((n,), dtype=torch.float32, device=z.device)
        check(lib.rfx_composite_forward(ptr(raw_c), ptr(z), n, S, trunc, sc_factor, ptr(rgb), ptr(depth), None,
                                        stream_ptr(z.device)), "rfx_composite_forward")
        ctx.save_for_backward(raw_c, z)
        ctx.trunc, ctx.sc = trunc, sc_factor
        return rgb, depth

    @staticmethod
    def backward(ctx, d_rgb, d_depth):
        lib = _lib.load()
        raw, z = ctx.saved_tensors
        n, S = z.shape
        d_raw = torch.empty_like(raw)
        check(lib.rfx_composite_backward(ptr(raw), ptr(z), n, S, ctx.trunc, ctx.sc, ptr(d_rgb.contiguous()),
                                         ptr(d_depth.contiguous()), ptr(d_raw), stream_ptr(z.device)),
              "rfx_composite_backward")
        return d_raw, None, None, None


class _MappingFn(torch.autograd.Function):
    """The whole train-mode forward of JointEncoding.mapping as ONE autograd node: S1 sampler, points,
    Q1 field, R1 compositing and the L1 losses are six kernel launches; backward is three.  Returns
    (rgb_loss, depth_loss, sdf_loss, fs_loss, rgb_map, depth_map)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, target_rgb, target_d, table, w1, w2, w3, w4, model, clamp, wvec=None):
        lib = _lib.load()
        cfg = model.config
        tr = cfg["training"]
        dev = rays_o.device
        st = stream_ptr(dev)
        o, d = rays_o.detach().to(torch.float32).contiguous(), rays_d.detach().to(torch.float32).contiguous()
        tgt = target_rgb.detach().to(torch.float32).contiguous()
        td = target_d.detach().reshape(-1).to(torch.float32).contiguous()
        n = o.shape[0]
        S = int(tr["n_range_d"]) + int(tr["n_samples_d"])
        u = torch.rand((n, S), dtype=torch.float32, device=dev) if tr["perturb"] > 0.0 else None
        z = torch.empty((n, S), dtype=torch.float32, device=dev)
        sd = model._sampler_desc()
        check(lib.rfx_sample_z(C.byref(sd), ptr(td), ptr(u), n, ptr(z), st), "rfx_sample_z")
        x01 = torch.empty((n * S, 3), dtype=torch.float32, device=dev)
        check(lib.rfx_ray_points(ptr(o), ptr(d), ptr(z), n, S, model._bbox6, model._bbox_f64, ptr(x01), st), "rfx_ray_points")
        raw = torch.empty((n * S, 4), dtype=torch.float32, device=dev)
        desc = model._field_desc(clamp)
        check(lib.rfx_field_forward(C.byref(desc), ptr(x01), n * S, ptr(raw), st), "rfx_field_forward")
        out = torch.empty((n, 4), dtype=torch.float32, device=dev)      # rgb map [n,3] | depth map [n]
        rgb_map, depth_map = torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev)
        trunc, sc = float(tr["trunc"]), float(cfg["data"]["sc_factor"])
        check(lib.rfx_composite_forward(ptr(raw), ptr(z), n, S, trunc, sc, ptr(rgb_map), ptr(depth_map), None, st),
              "rfx_composite_forward")
        sums = torch.empty(_lib.LOSS_WS_DOUBLES, dtype=torch.float64, device=dev)
        lc = torch.empty(8, dtype=torch.float32, device=dev)            # losses[4] | coef[4]
        check(lib.rfx_mapping_loss_forward(ptr(raw), ptr(z), ptr(rgb_map), ptr(depth_map), ptr(tgt), ptr(td), n, S,
                                           trunc * sc, float(cfg["cam"]["depth_trunc"]), int(tr["rgb_missing"] > 0),
                                           sums.data_ptr(), lc.data_ptr(), lc.data_ptr() + 16, st), "rfx_mapping_loss_forward")
        ctx.save_for_backward(o, d, z, x01, raw, rgb_map, depth_map, tgt, td, lc, table, w1, w2, w3, w4)
        ctx.model, ctx.clamp, ctx.dims = model, clamp, (n, S)
        ctx.mark_non_differentiable(z)
        del out
        # wvec dev [4]: also hand back sum_i w_i * loss_i so the caller's weighting costs no extra graph nodes
        ctx.wvec = wvec
        total = torch.dot(lc[:4], wvec) if wvec is not None else lc.new_zeros(())
        return lc[0], lc[1], lc[2], lc[3], rgb_map, depth_map, z, raw.view(n, S, 4), total

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_sdf, g_fs, g_rgb_map, g_depth_map, _gz, g_raw, g_total):
        lib = _lib.load()
        o, d, z, x01, raw, rgb_map, depth_map, tgt, td, lc, table, w1, w2, w3, w4 = ctx.saved_tensors
        model, (n, S) = ctx.model, ctx.dims
        cfg = model.config
        tr = cfg["training"]
        dev = o.device
        st = stream_ptr(dev)
        singles = (g_rgb, g_depth, g_sdf, g_fs)
        gout = (g_total.to(torch.float32) * ctx.wvec) if (g_total is not None and ctx.wvec is not None) else None
        if gout is None or any(g is not None for g in singles):
            zero = lc.new_zeros(())
            gs = torch.stack([g if g is not None else zero for g in singles]).to(torch.float32)
            gout = gs if gout is None else gout + gs
        gout = gout.contiguous()
        d_raw = torch.empty_like(raw)
        trunc, sc = float(tr["trunc"]), float(cfg["data"]["sc_factor"])
        check(lib.rfx_mapping_loss_backward(ptr(raw), ptr(z), ptr(rgb_map), ptr(depth_map), ptr(tgt), ptr(td), n, S, trunc, sc,
                                            trunc * sc, float(cfg["cam"]["depth_trunc"]), int(tr["rgb_missing"] > 0),
                                            lc.data_ptr() + 16, ptr(gout),
                                            ptr(g_rgb_map.contiguous()) if g_rgb_map is not None else None,
                                            ptr(g_depth_map.contiguous()) if g_depth_map is not None else None, ptr(d_raw), st),
              "rfx_mapping_loss_backward")
        if g_raw is not None:
            d_raw = d_raw + g_raw.reshape(-1, 4)
        need = ctx.needs_input_grad
        want_dx = need[0] or need[1]
        dx = torch.empty_like(x01) if want_dx else None
        dt = torch.zeros_like(table) if need[4] else None
        dws = [torch.zeros_like(w) if nd else None for w, nd in zip((w1, w2, w3, w4), need[5:9])]
        ws = model._workspace(model._backward_workspace_bytes(n * S), dev)
        desc = model._field_desc(ctx.clamp)
        _field_backward_staged(lib, desc, x01, n * S, d_raw, dt, dws, dx, ws, st)
        go = gd = None
        if want_dx:
            dp = dx.view(n, S, 3) / model._extent_on(dev)
            go = dp.sum(1) if need[0] else None
            gd = (dp * z[..., None]).sum(1) if need[1] else None
        return go, gd, None, None, dt, dws[0], dws[1], dws[2], dws[3], None, None, None


# ------------------------------------------------------------------------------ the module
class JointEncoding(nn.Module):
    def __init__(self, config, bound_box, num_kf=None):
        super().__init__()
        self.config = config
        self.bounding_box = bound_box
        self.num_kf = num_kf
        self.clamp = False
        self.get_resolution()
        self.get_encoding(config)
        self.get_decoder(config)
        self.count = 0
        self._ws_buf: Optional[torch.Tensor] = None
        self._refresh_box()

    # -- geometry helpers -----------------------------------------------------------------
    def _refresh_box(self):
        bb = self.bounding_box.detach().cpu()
        # torch promotes (fp32 pts - bound) to float64 iff the bound tensor is floating (float64 when
        # built from a yaml list with a non-integer entry, run.py:90); integer bounds stay fp32.
        self._bbox_f64 = 1 if bb.dtype == torch.float64 else 0
        self._bbox6 = farr(_D6, bb.to(torch.float64).reshape(-1).tolist())
        self._extent32 = (bb[:, 1] - bb[:, 0]).to(torch.float32)
        self._extent_dev = {}

    def _extent_on(self, device) -> torch.Tensor:
        """bound extent as an fp32 tensor on ``device`` (cached: an H2D copy here would sync every backward)."""
        key = str(device)
        if key not in self._extent_dev:
            self._extent_dev[key] = self._extent32.to(device)
        return self._extent_dev[key]

    def get_resolution(self):
        """reference :24-37."""
        dim_max = (self.bounding_box[:, 1] - self.bounding_box[:, 0]).max()
        g = self.config["grid"]
        self.resolution_sdf = g["voxel_sdf"] if g["voxel_sdf"] > 10 else int(dim_max / g["voxel_sdf"])
        self.resolution_color = g["voxel_color"] if g["voxel_color"] > 10 else int(dim_max / g["voxel_color"])

    def get_encoding(self, config, GBV=True):
        """reference :41-93: OneBlob position encoding, hash-grid residual features, GBV / GBW."""
        self.embedpos_fn, self.input_ch_pos = get_encoder(config["pos"]["enc"], n_bins=config["pos"]["n_bins"])
        if config["pos"].get("fp16_opt_in", False) and hasattr(self.embedpos_fn, "fp16"):
            self.embedpos_fn.fp16 = True      # explicit opt-in (not the reference's precision): see encodings.OneBlob
        self.embed_res_fn, self.input_ch = get_encoder(config["grid"]["enc"], log2_hashmap_size=config["grid"]["hash_size"],
                                                       desired_resolution=self.resolution_sdf)
        if GBV:
            gv = config["globalV"]
            self.device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
            self.GBV = DenseGrid(gv["base_resolution"], gv["n_features_per_level"], gv["n_levels"], gv["per_level_scale"])
            self.GBV.requires_grad_(False)
            self.GBW = DenseGrid(gv["base_resolution"], 1, gv["n_levels"], gv["per_level_scale"])
            self.GBW.requires_grad_(False)
            with torch.no_grad():
                self.GBW.params[:] = 0.0

    def get_decoder(self, config):
        """reference :96-105."""
        self.decoder_res = ColorSDFNet(config, input_ch=self.input_ch, input_ch_pos=self.input_ch_pos)
        self.color_net_res = batchify(self.decoder_res.color_net, None)
        self.sdf_net_res = batchify(self.decoder_res.sdf_net, None)
        self.rba = RBA(self.num_kf, scale=config["mapping"]["pose_scale"])

    # -- librfx plumbing --------------------------------------------------------------------
    def _field_desc(self, clamp: bool, partitioned_ok: bool = False) -> FieldDesc:
        if not partitioned_ok and getattr(self.embed_res_fn, "partition_stale", False):
            raise _lib.RfxError("the hash table is partitioned by level over several GPUs and this rank's copy of the other ranks' "
                                "levels is out of date: call Mapper.sync_field() on every rank first")
        tr = self.config["training"]
        w1, w2, w3, w4 = self.decoder_res.fused_weights()
        d = FieldDesc()
        d.hash = self.embed_res_fn.desc
        d.hash_table = ptr(self.embed_res_fn.params)
        d.gbv = ptr(self.GBV.params)
        d.gbv_res = int(self.config["globalV"]["base_resolution"])
        d.w1, d.w2, d.w3, d.w4 = ptr(w1), ptr(w2), ptr(w3), ptr(w4)
        d.c_trunc, d.trunc = float(tr["c_trunc"]), float(tr["trunc"])
        d.tsdf_scale = d.c_trunc / d.trunc
        d.clamp_mode = 1 if clamp else 0
        d.clamp_hi = float(self.config["mapping"]["clamp"]) if clamp else 1.0
        d.pos_fp16 = 1 if getattr(self.embedpos_fn, "fp16", False) else 0
        d.staged = self._staged_weights(d, (w1, w2, w3, w4))
        return d

    def _staged_weights(self, d: FieldDesc, ws) -> int:
        """decoder weights in MFMA operand order: one tiny launch per descriptor (i.e. per forward / backward
        call), after which the kernels copy the image instead of every block gathering it from the four Linear
        tensors.  Re-staged every time on purpose: optimizers update the weights in place."""
        lib = _lib.load()
        buf = getattr(self, "_staged_buf", None)
        if buf is None or buf.device != ws[0].device:
            buf = self._staged_buf = torch.empty(int(lib.rfx_field_staged_floats()), dtype=torch.float32, device=ws[0].device)
        d.staged = None
        check(lib.rfx_field_stage_weights(C.byref(d), ptr(buf), stream_ptr(buf.device)), "rfx_field_stage_weights")
        return buf.data_ptr()

    def _sampler_desc(self) -> SamplerDesc:
        tr, cam = self.config["training"], self.config["cam"]
        s = SamplerDesc()
        s.near, s.far, s.range_d = float(cam["near"]), float(cam["far"]), float(tr["range_d"])
        s.n_range_d, s.n_samples_d, s.perturb = int(tr["n_range_d"]), int(tr["n_samples_d"]), float(tr["perturb"])
        return s

    def _backward_workspace_bytes(self, n: int) -> int:
        """workspace of the field backward for n points: the library's minimum + what lets the table scatter (its last region)
        keep ALL binned levels of a large table (T >= 2^19) in one group of launches (the first-frame mapping's 500-1000
        iterations ran them one level at a time until round 6); equal to the minimum for small tables"""
        lib = _lib.load()
        enc = self.embed_res_fn
        extra = int(lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(enc.desc), n)) - int(lib.rfx_grid_encode_backward_workspace_bytes(n, int(enc.desc.n_levels)))
        return int(lib.rfx_field_backward_workspace_bytes(n)) + max(extra, 0)

    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        n = (nbytes + 3) // 4
        if self._ws_buf is None or self._ws_buf.numel() < n or self._ws_buf.device != device:
            self._ws_buf = torch.empty(int(n * 1.25), dtype=torch.float32, device=device)
        return self._ws_buf

    def _flat(self, query_points):
        return torch.reshape(query_points, [-1, query_points.shape[-1]]).to(torch.float32).contiguous()

    # -- R1 -----------------------------------------------------------------------------------
    def sdf2weights(self, sdf, z_vals, args=None):
        """Normalised bell-shaped weights with first-surface mask (reference :107-127).  Evaluated by
        the compositing kernel (weights output); kept for API parity."""
        args = args or self.config
        lib = _lib.load()
        n, S = z_vals.shape
        raw = torch.zeros((n, S, 4), dtype=torch.float32, device=z_vals.device)
        raw[..., 3] = sdf
        w = torch.empty((n, S), dtype=torch.float32, device=z_vals.device)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=z_vals.device)
        dep = torch.empty((n,), dtype=torch.float32, device=z_vals.device)
        check(lib.rfx_composite_forward(ptr(raw), ptr(z_vals.contiguous()), n, S, float(args["training"]["trunc"]),
                                        float(args["data"]["sc_factor"]), ptr(rgb), ptr(dep), ptr(w),
                                        stream_ptr(z_vals.device)), "rfx_composite_forward")
        return w

    def raw2outputs(self, raw, z_vals):
        """reference :156-179."""
        return _CompositeFn.apply(raw[..., :4], z_vals, float(self.config["training"]["trunc"]),
                                  float(self.config["data"]["sc_factor"]))

    # -- Q1 / Q2 --------------------------------------------------------------------------------
    def query_color_sdf(self, query_points, ranged_mask=None):
        """raw [.., 4] = (rgb residual + GBV rgb, sdf residual + GBV tsdf) (reference :314-349)."""
        flat = torch.reshape(query_points, [-1, query_points.shape[-1]])
        w1, w2, w3, w4 = self.decoder_res.fused_weights()
        return _FieldFn.apply(flat, self.embed_res_fn.params, w1, w2, w3, w4, self, bool(self.clamp))

    def query_sdf_res(self, query_points, return_geo=False, embed=False):
        """reference :212-248.  embed=True returns the raw hash features (TV smoothness, slam.py:209)."""
        flat = torch.reshape(query_points, [-1, query_points.shape[-1]])
        if embed:
            emb = self.embed_res_fn(flat)
            return torch.reshape(emb, list(query_points.shape[:-1]) + [emb.shape[-1]])
        if return_geo:
            raise NotImplementedError("return_geo=True has no caller in the reference")
        lib = _lib.load()
        x = self._flat(query_points)
        out = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
        desc = self._field_desc(False)
        check(lib.rfx_field_query_sdf(C.byref(desc), ptr(x), x.shape[0], ptr(out), stream_ptr(x.device)), "rfx_field_query_sdf")
        return torch.reshape(out, list(query_points.shape[:-1]))

    def query_sdf_ex(self, query_points, return_geo=False, embed=False):
        """GBV tsdf channel (reference :250-265)."""
        ex = self.GBV(self._flat(query_points))
        return torch.reshape(ex[..., 0], list(query_points.shape[:-1]))

    def query_w_res(self, query_points, return_geo=False, embed=False):
        """GBW lookup (reference :269-282)."""
        ex_w = self.GBW(self._flat(query_points))
        return torch.reshape(ex_w, list(query_points.shape[:-1]))

    def query_color_residual(self, query_points):
        """reference :285-298 (the decoder is fed the un-rescaled GBV tsdf here)."""
        lib = _lib.load()
        x = self._flat(query_points)
        out = torch.empty((x.shape[0], 3), dtype=torch.float32, device=x.device)
        desc = self._field_desc(False)
        check(lib.rfx_field_query_color(C.byref(desc), ptr(x), x.shape[0], ptr(out), stream_ptr(x.device)),
              "rfx_field_query_color")
        return out

    def query_color_ex(self, query_points):
        """GBV rgb channels (reference :300-310)."""
        return self.GBV(self._flat(query_points))[..., 1:]

    def run_network(self, inputs, flat=False):
        """normalise to [0,1]^3 by the bounding box and query (reference :370-402)."""
        inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        if self.config["grid"]["tcnn_encoding"]:
            bb = self.bounding_box.to(inputs_flat.device)
            inputs_flat = (inputs_flat - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
        outputs_flat = self.query_color_sdf(inputs_flat)
        if flat:
            return outputs_flat
        return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])

    # -- S1 + Q1 + R1 ------------------------------------------------------------------------
    def sample_z_vals(self, target_d, n_rays, device):
        """depth-guided + uniform samples, sorted, stratified jitter (reference :415-441)."""
        tr = self.config["training"]
        if target_d is None:
            raise NotImplementedError("target_d=None needs training.n_samples, absent from every reference config")
        lib = _lib.load()
        S = int(tr["n_range_d"]) + int(tr["n_samples_d"])
        td = target_d.detach().reshape(-1).to(torch.float32).contiguous()
        u = torch.rand((n_rays, S), dtype=torch.float32, device=device) if tr["perturb"] > 0.0 else None
        z = torch.empty((n_rays, S), dtype=torch.float32, device=device)
        sd = self._sampler_desc()
        check(lib.rfx_sample_z(C.byref(sd), ptr(td), ptr(u), n_rays, ptr(z), stream_ptr(device)), "rfx_sample_z")
        return z

    def render_rays(self, rays_o, rays_d, target_d=None, tracking=False, frameid=None, render_flag=False):
        """reference :407-456."""
        n_rays = rays_o.shape[0]
        z_vals = self.sample_z_vals(target_d, n_rays, rays_o.device)
        S = z_vals.shape[1]
        x01 = _RayPointsFn.apply(rays_o, rays_d, z_vals, self)
        raw = self.query_color_sdf(x01).view(n_rays, S, 4)
        rgb_res_map, depth_res_map = self.raw2outputs(raw, z_vals)
        return {"rgb_res_map": rgb_res_map, "depth_res_map": depth_res_map, "z_vals": z_vals, "raw": raw}

    @torch.no_grad()
    def render_fused(self, rays_o, rays_d, target_d, jitter: bool = True):
        """Fused eval renderer (one launch; SLAM.render_single's workload, mp_slam/slam.py:290-344):
        returns (rgb [n,3], depth [n])."""
        lib = _lib.load()
        o, d = rays_o.to(torch.float32).contiguous(), rays_d.to(torch.float32).contiguous()
        td = target_d.reshape(-1).to(torch.float32).contiguous()
        n = o.shape[0]
        tr = self.config["training"]
        S = int(tr["n_range_d"]) + int(tr["n_samples_d"])
        u = torch.rand((n, S), dtype=torch.float32, device=o.device) if (jitter and tr["perturb"] > 0.0) else None
        rgb = torch.empty((n, 3), dtype=torch.float32, device=o.device)
        depth = torch.empty((n,), dtype=torch.float32, device=o.device)
        fd, sd = self._field_desc(bool(self.clamp)), self._sampler_desc()
        check(lib.rfx_render_rays(C.byref(fd), C.byref(sd), ptr(o), ptr(d), ptr(td), ptr(u), n, self._bbox6, self._bbox_f64,
                                  float(self.config["data"]["sc_factor"]), ptr(rgb), ptr(depth), stream_ptr(o.device)),
              "rfx_render_rays")
        return rgb, depth

    def mapping(self, rays_o, rays_d, target_rgb, target_d, tracking=False, render_flag=False, clamp=False):
        """One forward of the mapping objective (reference :460-529).  Train mode: dict of the four
        losses (+ rendered rgb/depth); eval mode: the render dict."""
        self.clamp = clamp
        if not self.training:
            return self.render_rays(rays_o, rays_d, target_d=target_d, tracking=tracking, render_flag=render_flag)
        # train mode: one fused autograd node (sampler + points + field + compositing + losses)
        w1, w2, w3, w4 = self.decoder_res.fused_weights()
        rgb_l, depth_l, sdf_l, fs_l, rgb_map, depth_map, _z, _raw, total = _MappingFn.apply(
            rays_o, rays_d, target_rgb, target_d, self.embed_res_fn.params, w1, w2, w3, w4, self, bool(clamp),
            self._loss_weights(rays_o.device))
        # "loss_weighted" = rgb_weight*rgb + depth_weight*depth + sdf_weight*sdf + fs_weight*fs (training.* weights),
        # what SLAM.get_loss_from_ret assembles from the four entries above, pre-summed inside the node
        return {"rgb_res_loss": rgb_l, "depth_res_loss": depth_l, "sdf_res_loss": sdf_l, "fs_res_loss": fs_l,
                "rgb_res": rgb_map, "depth_res": depth_map, "loss_weighted": total}

    def _loss_weights(self, device) -> torch.Tensor:
        tr = self.config["training"]
        key = (str(device), float(tr["rgb_weight"]), float(tr["depth_weight"]), float(tr["sdf_weight"]), float(tr["fs_weight"]))
        if getattr(self, "_wvec_key", None) != key:
            self._wvec = torch.tensor(key[1:], dtype=torch.float32, device=device)
            self._wvec_key = key
        return self._wvec

    def mapping_unfused(self, rays_o, rays_d, target_rgb, target_d, clamp=False):
        """the same objective assembled from the individual kernels + torch ops (kept for tests: the
        fused node must agree with it).  Mirrors the reference's loss block line by line (:493-527)."""
        self.clamp = clamp
        rend = self.render_rays(rays_o, rays_d, target_d=target_d)
        cfg = self.config
        td = target_d.squeeze()
        valid_depth_mask = (td > 0.0) * (td < cfg["cam"]["depth_trunc"])
        # NOTE: like the reference (:495-498) rgb_weight stays a *bool* tensor, so assigning
        # training.rgb_missing casts it to True/False: any rgb_missing > 0 weighs invalid-depth rays by 1.
        rgb_weight = valid_depth_mask.clone().unsqueeze(-1)
        rgb_weight[rgb_weight == 0] = cfg["training"]["rgb_missing"]
        rgb_res_loss = compute_loss(rend["rgb_res_map"] * rgb_weight, target_rgb * rgb_weight)
        depth_res_loss = compute_loss(rend["depth_res_map"].squeeze()[valid_depth_mask], td[valid_depth_mask])
        truncation = cfg["training"]["trunc"] * cfg["data"]["sc_factor"]
        fs_res_loss, sdf_res_loss = get_sdf_loss(rend["z_vals"], target_d, rend["raw"][..., 3], truncation, loss_type="l2",
                                                 middle_mask=valid_depth_mask)
        return {"rgb_res_loss": rgb_res_loss, "depth_res_loss": depth_res_loss, "sdf_res_loss": sdf_res_loss,
                "fs_res_loss": fs_res_loss, "rgb_res": rend["rgb_res_map"], "depth_res": rend["depth_res_map"]}
