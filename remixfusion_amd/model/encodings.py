"""Encoders of the residual field, MI355X-native: drop-in for the reference's tinycudann factory
``get_encoder`` (model/encodings.py:6-102).  Each encoder is an ``nn.Module`` with a flat fp32
``params`` tensor in tiny-cuda-nn's layout, evaluated by librfx HIP kernels (no tinycudann, no
CPU fallback).

Supported (what the reference configs select): 'HashGrid' (:33-51), 'OneBlob' (:65-76) and the
dense ``Grid`` used for GBV/GBW (model/scene_rep.py:60-93, via ``DenseGrid``).  Spherical /
Frequency / Identity / 'dense' multi-level (:15-30,:53-62,:80-101) are not selected by any
config and raise NotImplementedError.
"""
from __future__ import annotations

from typing import Tuple

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import GridDesc, check, ptr, stream_ptr


def make_grid_desc(n_levels: int, n_feat: int, log2_hashmap_size: int, base_resolution: int,
                   per_level_scale: float, hashed_type: bool) -> Tuple[GridDesc, int]:
    """tiny-cuda-nn level geometry: scale_l = exp2(l*log2(pls))*base - 1, res_l = ceil(scale)+1,
    params_l = min(round_up(res^3, 8), 2^T) for hash grids.  Returns (desc, total entries)."""
    if n_levels > _lib.RFX_MAX_LEVELS:
        raise ValueError("too many levels")
    d = GridDesc()
    d.n_levels, d.n_feat = n_levels, n_feat
    log2_pls = np.float32(np.log2(np.float32(per_level_scale)))
    offset = 0
    for l in range(n_levels):
        scale = np.float32(np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        cap = (2 ** 32 - 1) // 2
        size = cap if float(res) ** 3 > float(cap) else res ** 3
        size = (size + 7) // 8 * 8
        if hashed_type:
            size = min(size, 1 << log2_hashmap_size)
        stride, dim = 1, 0
        while dim < 3 and stride <= size:
            stride *= res
            dim += 1
        d.scale[l], d.res[l], d.size[l], d.offset[l] = float(scale), res, size, offset
        d.hashed[l] = 1 if (hashed_type and size < stride) else 0
        offset += size
    return d, offset


class _GridEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x01, params, module):
        lib = _lib.load()
        if getattr(module, "partition_stale", False):
            raise _lib.RfxError("this grid is partitioned by level over several GPUs and the local copy is out of date: "
                                "Mapper.sync_field() first")
        x = x01.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty((n, module.n_output_dims), dtype=torch.float32, device=x.device)
        check(lib.rfx_grid_encode_forward(module.desc, ptr(params), ptr(x), n, ptr(out), stream_ptr(x.device)),
              "rfx_grid_encode_forward")
        ctx.save_for_backward(x, params)
        ctx.module = module
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, params = ctx.saved_tensors
        module = ctx.module
        if module.desc.n_feat != 2:
            raise _lib.RfxError("grid backward is only implemented for the 2-feature hash grid")
        n = x.shape[0]
        dout = dout.contiguous()
        dparams = torch.zeros_like(params) if ctx.needs_input_grad[1] else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = None
        if dparams is not None:       # staging buffer of the LDS-privatised scatter
            nb = int(lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(module.desc), n))
            ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
        check(lib.rfx_grid_encode_backward(module.desc, ptr(params), ptr(x), n, ptr(dout), ptr(dparams), ptr(dx),
                                           ptr(ws), 0 if ws is None else ws.numel() * 4, stream_ptr(x.device)),
              "rfx_grid_encode_backward")
        return dx, dparams, None


class GridEncoding(nn.Module):
    """tinycudann.Encoding(otype Grid/HashGrid) stand-in: ``params`` flat fp32, U(-1e-4, 1e-4)."""

    def __init__(self, desc: GridDesc, n_entries: int, seed: int = 1337):
        super().__init__()
        self.desc = desc
        self.n_input_dims = 3
        self.n_output_dims = desc.n_levels * desc.n_feat
        g = torch.Generator().manual_seed(seed)
        init = (torch.rand(n_entries * desc.n_feat, generator=g, dtype=torch.float32) * 2.0 - 1.0) * 1e-4
        self.params = nn.Parameter(init)

    def forward(self, x01: torch.Tensor) -> torch.Tensor:
        return _GridEncodeFn.apply(x01, self.params, self)


class HashGrid(GridEncoding):
    def __init__(self, n_levels=16, level_dim=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=2.0):
        desc, n = make_grid_desc(n_levels, level_dim, log2_hashmap_size, base_resolution, per_level_scale, True)
        super().__init__(desc, n)


class DenseGrid(GridEncoding):
    """tcnn ``Grid``/``Dense`` with n_levels=1 (GBV: F=4, GBW: F=1); x fastest, features interleaved."""

    def __init__(self, base_resolution=200, n_features=4, n_levels=1, per_level_scale=1.0):
        desc, n = make_grid_desc(n_levels, n_features, 0, base_resolution, per_level_scale, False)
        super().__init__(desc, n)
        self.base_resolution = base_resolution


class OneBlob(nn.Module):
    """tcnn OneBlob (n_bins=16).  No parameters; ``params`` is an empty tensor like tcnn's.
    The reference builds it with ``dtype=torch.float`` (model/encodings.py:65-76, line 73), so the
    48 outputs are fp32 and that is the default here.  ``fp16=True`` is an explicit opt-in that
    rounds the outputs to half precision (tinycudann's own default output type, NOT what the
    reference selects) and lets the fused kernels run these columns on the fp16 matrix pipe."""

    def __init__(self, n_bins=16, fp16=False):
        super().__init__()
        if n_bins != 16:
            raise NotImplementedError("librfx implements n_bins=16 (every reference config)")
        self.n_bins, self.fp16 = n_bins, fp16
        self.n_input_dims, self.n_output_dims = 3, 3 * n_bins
        self.params = nn.Parameter(torch.empty(0), requires_grad=False)

    def forward(self, x01: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        x = x01.detach().to(torch.float32).contiguous()
        out = torch.empty((x.shape[0], self.n_output_dims), dtype=torch.float32, device=x.device)
        check(lib.rfx_oneblob_forward(ptr(x), x.shape[0], self.n_bins, int(self.fp16), ptr(out), stream_ptr(x.device)),
              "rfx_oneblob_forward")
        return out


def get_encoder(encoding, input_dim=3, degree=4, n_bins=16, n_frequencies=12, n_levels=16, level_dim=2,
                base_resolution=16, log2_hashmap_size=19, desired_resolution=512):
    """Same signature and return value ``(module, out_dim)`` as the reference (model/encodings.py:6-10)."""
    name = encoding.lower()
    if "hash" in name or "tiled" in name:
        if "tiled" in name:
            raise NotImplementedError("tiled grids are not selected by any reference config")
        per_level_scale = np.exp2(np.log2(desired_resolution / n_levels) / (n_levels - 1)) if n_levels > 1 else 1.0
        embed = HashGrid(n_levels, level_dim, log2_hashmap_size, base_resolution, float(per_level_scale))
    elif "blob" in name:
        embed = OneBlob(n_bins)
    else:
        raise NotImplementedError(f"encoding {encoding!r}: only HashGrid and OneBlob are used by the reference configs")
    return embed, embed.n_output_dims
