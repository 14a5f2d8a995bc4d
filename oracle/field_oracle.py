"""oracle/field_oracle.py -- TEST INFRASTRUCTURE ONLY (parity oracle + cpu_baseline).

CPU (torch fp32) restatement of the residual neural field path of the reference.  Nothing in
``remixfusion_amd/`` may import this module; only ``tests/``, ``bench.py``'s ``cpu_baseline``
leg and ``__graft_entry__.smoke()`` use it, and only as the checker.

Reference lines followed (paths under /root/reference):
  sample_z_vals        model/scene_rep.py:415-441  (render_rays z sampler, S1)
  query_color_sdf      model/scene_rep.py:314-349  (Q1), run_network :370-402
  query_sdf_res & co   model/scene_rep.py:212-310  (Q2)
  sdf2weights          model/scene_rep.py:107-127  (R1)
  raw2outputs          model/scene_rep.py:156-179  (R1)
  mapping_losses       model/scene_rep.py:493-527 + model/utils.py:170-256 (L1)
  total_loss           mp_slam/slam.py:145-190
  smoothness           mp_slam/slam.py:193-217     (TV1)
  mlp_forward          model/decoder.py:116-146    (D1; torch Linear, bias=False)

Third-party arithmetic that is NOT in /root/reference: the encodings come from
``tinycudann`` (NVlabs/tiny-cuda-nn ``bindings/torch``, un-pinned: requirements.txt:22).
``hash_encode`` / ``dense_grid_encode`` / ``oneblob_encode`` restate that library's published
algorithm (include/tiny-cuda-nn/encodings/grid.h, oneblob.h, common_device.h) as used at
the call sites model/encodings.py:33-51,65-76 and model/scene_rep.py:60-93.

PARITY STATUS: D1, R1, S1(deterministic part), L1 are pinned by golden vectors generated from
the reference's own importable torch code (tests/golden/make_golden.py).  E1/E2/E3 are
"parity unpinned": tinycudann cannot be installed or run here and the reference holds no
vectors for it.

Conventions restated from tiny-cuda-nn:
  * level l: scale = exp2(l*log2(pls))*base - 1 ; res = ceil(scale)+1 ;
    params_l = min(round_up(res^3, 8), 2^log2_T) (hash) ; tables concatenated, F floats/entry.
  * pos = fma(scale, x, 0.5) ; g = floor(pos) (as uint32) ; f = pos - g.
  * corner c (bit0=x, bit1=y, bit2=z): w = prod(bit ? f : 1-f), grid = g + bit.
  * index: stride walk ``for dim: if stride <= params_l: index += g[dim]*stride; stride *= res``,
    hashed (g0*1 ^ g1*2654435761 ^ g2*805459861, uint32) iff params_l < stride ; ``% params_l``.
  * output [B, L*F] level-major ; all integer arithmetic wraps at 32 bits.
  * OneBlob: out[d*n+k] = L(k+1) - L(k), L(k) = cdf(k/n-x)+cdf(k/n-x-1)+cdf(k/n-x+1),
    L(n) := L(0)+1, cdf(t) = clamp(15/16*u*(1-2/3u^2+1/5u^4)+1/2, 0, 1), u = t*n.
    The reference constructs OneBlob with ``dtype=torch.float`` (model/encodings.py:67-74, the
    ``dtype`` argument is on line 73), i.e. tinycudann Precision.Fp32: the 48 outputs are fp32 and
    ``pos_fp16=False`` is the reference behaviour.  ``pos_fp16=True`` (rounding the outputs to fp16
    and back) models tinycudann's *default* half-precision output and is kept only as the oracle of
    the product's explicit opt-in fast path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_M32 = 0xFFFFFFFF
_P1 = 2654435761
_P2 = 805459861


# ----------------------------------------------------------------------------- grid geometry
@dataclass
class GridMeta:
    n_levels: int
    n_feat: int
    scales: List[float]
    res: List[int]
    sizes: List[int]      # entries per level
    offsets: List[int]    # entry offset per level (len L+1)
    hashed: List[bool]

    @property
    def n_params(self) -> int:
        return self.offsets[-1] * self.n_feat


def grid_meta(n_levels: int, n_feat: int, log2_hashmap_size: int, base_resolution: int,
              per_level_scale: float, grid_type: str = "hash") -> GridMeta:
    """tiny-cuda-nn GridEncodingTemplated constructor (offset table) + grid_scale/grid_resolution.
    All float math in fp32 like the C++ (exp2f, ceilf)."""
    log2_pls = np.float32(np.log2(np.float32(per_level_scale)))
    scales, res, sizes, offsets, hashed = [], [], [], [0], []
    for l in range(n_levels):
        s = np.float32(np.exp2(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0)
        s = np.float32(s)
        r = int(np.ceil(s)) + 1
        dense = r ** 3
        maxp = (2 ** 32 - 1) // 2
        p = maxp if float(r) ** 3 > float(maxp) else dense
        p = (p + 7) // 8 * 8
        if grid_type == "hash":
            p = min(p, 1 << log2_hashmap_size)
        scales.append(float(s)); res.append(r); sizes.append(p)
        offsets.append(offsets[-1] + p)
        # hashed iff the stride walk ends with stride > params (grid_index)
        stride, d = 1, 0
        while d < 3 and stride <= p:
            stride *= r; d += 1
        hashed.append(grid_type == "hash" and p < stride)
    return GridMeta(n_levels, n_feat, scales, res, sizes, offsets, hashed)


def hashgrid_meta_from_config(log2_hashmap_size: int, desired_resolution: int, n_levels: int = 16,
                              level_dim: int = 2, base_resolution: int = 16) -> GridMeta:
    """model/encodings.py:33-51: per_level_scale = exp2(log2(R/n_levels)/(n_levels-1))."""
    pls = float(np.exp2(np.log2(desired_resolution / n_levels) / (n_levels - 1))) if n_levels > 1 else 1.0
    return grid_meta(n_levels, level_dim, log2_hashmap_size, base_resolution, pls, "hash")


def _corner_indices(g: torch.Tensor, res: int, size: int, hashed: bool) -> torch.Tensor:
    """g: [B,3] int64 holding uint32 values. Returns [B] int64 entry index (grid_index)."""
    g0, g1, g2 = g[:, 0], g[:, 1], g[:, 2]
    if hashed:
        h = (g0 & _M32) ^ ((g1 * _P1) & _M32) ^ ((g2 * _P2) & _M32)
        return (h & _M32) % size
    idx = torch.zeros_like(g0)
    stride = 1
    for gd in (g0, g1, g2):
        if stride <= size:
            idx = (idx + gd * stride) & _M32
            stride *= res
    return idx % size


def grid_encode(x: torch.Tensor, table: torch.Tensor, meta: GridMeta) -> torch.Tensor:
    """E1/E3 forward.  x [B,3] fp32, table flat fp32 [meta.n_params].  Returns [B, L*F].
    Differentiable w.r.t. table (gather) and x (through the interpolation weights)."""
    B = x.shape[0]
    Fd = meta.n_feat
    tab = table.view(-1, Fd)
    outs = []
    for l in range(meta.n_levels):
        scale = torch.tensor(meta.scales[l], dtype=torch.float32)
        # scale*x + 0.5 as ONE rounding, like tiny-cuda-nn's `fmaf(scale, x, 0.5f)` (grid.h, pos_fract) and the kernels' fmaf:
        # torch.addcmul on float32 CPU tensors rounds the product first, so the sum is formed in float64 (the product of two
        # float32 values is exact there; the one rounding of the sum differs from a true fma only where the float64 sum itself
        # rounds -- 29 bits below the float32 result: a double-rounding case no test point has met) and rounded once.
        # Differentiable in x like the addcmul it replaces (d pos / d x = scale).
        pos = (x.double() * scale.double() + 0.5).to(x.dtype) if x.dtype == torch.float32 else torch.addcmul(
            torch.tensor(0.5, dtype=x.dtype), x, scale.to(x.dtype))
        flo = torch.floor(pos)
        g = flo.detach().to(torch.int64) & _M32   # (uint32)(int)floor
        f = pos - flo.detach()
        acc = torch.zeros(B, Fd, dtype=torch.float32)
        for c in range(8):
            w = torch.ones(B, dtype=torch.float32)
            gl = g.clone()
            for d in range(3):
                if (c >> d) & 1:
                    w = w * f[:, d]
                    gl[:, d] = (g[:, d] + 1) & _M32
                else:
                    w = w * (1.0 - f[:, d])
            idx = _corner_indices(gl, meta.res[l], meta.sizes[l], meta.hashed[l]) + meta.offsets[l]
            acc = acc + w[:, None] * tab[idx]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


def dense_meta(base_resolution: int, n_feat: int) -> GridMeta:
    """model/scene_rep.py:60-93: Grid/Dense, n_levels=1, per_level_scale=1."""
    return grid_meta(1, n_feat, 0, base_resolution, 1.0, "dense")


# ----------------------------------------------------------------------------- OneBlob (E2)
def _quartic_cdf(t: torch.Tensor, n: int) -> torch.Tensor:
    u = t * float(n)
    u2 = u * u
    u4 = u2 * u2
    return torch.clamp((15.0 / 16.0) * u * (1.0 - (2.0 / 3.0) * u2 + (1.0 / 5.0) * u4) + 0.5, 0.0, 1.0)


def oneblob_encode(x: torch.Tensor, n_bins: int = 16, pos_fp16: bool = False) -> torch.Tensor:
    """x [B,3] -> [B, 3*n_bins], dim-major."""
    B, D = x.shape
    k = torch.arange(n_bins + 1, dtype=torch.float32) / float(n_bins)       # boundaries 0..1
    left = k[None, None, :n_bins] - x[:, :, None]                            # [B,D,n]
    Lk = _quartic_cdf(left, n_bins) + _quartic_cdf(left - 1.0, n_bins) + _quartic_cdf(left + 1.0, n_bins)
    right = torch.cat([Lk[:, :, 1:], Lk[:, :, :1] + 1.0], dim=-1)
    out = (right - Lk).reshape(B, D * n_bins)
    if pos_fp16:
        # straight-through rounding to half precision (NOT the reference: opt-in fast path only)
        out = out + (out.detach().to(torch.float16).to(torch.float32) - out.detach())
    return out


# ----------------------------------------------------------------------------- MLP (D1)
MLP_PROBE = None       # tests: a dict that mlp_forward fills with its layer inputs, pre-activations and hidden activations
                       # (the latter with retain_grad) -- what relu_tie_bounds() below needs


def mlp_forward(embed, embed_pos, ex_tsdf, ex_rgb, W1, W2, W3, W4):
    """model/decoder.py:132-146 with torch Linear weights W[out,in], no bias.
    W1 [32,81], W2 [16,32], W3 [32,66], W4 [3,32]."""
    x1 = torch.cat([embed, embed_pos, ex_tsdf], -1)
    p1 = F.linear(x1, W1)
    h1 = torch.relu(p1)
    h = F.linear(h1, W2)
    sdf, geo = h[..., :1], h[..., 1:]
    x3 = torch.cat([embed_pos, geo, ex_rgb], -1)
    p3 = F.linear(x3, W3)
    h3 = torch.relu(p3)
    rgb = F.linear(h3, W4)
    if MLP_PROBE is not None:
        for t in (h1, h3):
            if t.requires_grad:
                t.retain_grad()
        MLP_PROBE.update(X1=x1, P1=p1, H1=h1, X3=x3, P3=p3, H3=h3, W1=W1, W3=W3)
    return torch.cat([rgb, sdf], -1)


def relu_tie_bounds(probe, n_terms=(81, 66)):
    """Per ELEMENT of dW1 [32,81] and dW3 [32,66]: how much of the gradient hangs on hidden pre-activations that any fp32
    evaluation may round to the other side of zero.  `probe` = MLP_PROBE after a float64 forward + backward.
    A dot product of n fp32 terms, summed in whatever order, lies within gamma_n * sum |terms| of the exact value
    (gamma_n = n u / (1 - n u), u = 2^-24), so the sign of pre-activation (p, r) is the same in EVERY fp32 evaluation unless
    |exact| <= gamma_n * sum_i |W[r,i] X[p,i]|.  Only for those (sample, unit) pairs -- the possible ties -- may the sample's term
    dH[p,r] * X[p,:] enter or leave row r of the weight gradient.  The bound returned for element (r, i) is the sum of
    |dH[p,r] X[p,i]| over the possible ties of row r: zero for a row without one."""
    out = []
    for X, Pre, H, W, n, gk in ((probe["X1"], probe["P1"], probe["H1"], probe["W1"], n_terms[0], "G1"),
                                (probe["X3"], probe["P3"], probe["H3"], probe["W3"], n_terms[1], "G3")):
        X, Pre, W = X.detach().double().reshape(-1, X.shape[-1]), Pre.detach().double().reshape(-1, Pre.shape[-1]), W.detach().double()
        G = probe.get(gk, H.grad)                    # (callers that differentiate with autograd.grad store G1 / G3 themselves)
        if G is None:
            raise RuntimeError("relu_tie_bounds: run the backward first (no gradient at the hidden activations)")
        G = G.detach().double().reshape(-1, H.shape[-1])
        u = 2.0 ** -24
        gamma = n * u / (1.0 - n * u)
        tie = Pre.abs() <= 2.0 * gamma * (X.abs() @ W.abs().T)          # either of two evaluations may be off by gamma: 2 gamma apart
        out.append(((tie.double() * G.abs()).T @ X.abs(), int(tie.sum())))
    return out


# ----------------------------------------------------------------------------- field (Q1/Q2)
@dataclass
class FieldParams:
    hash_meta: GridMeta
    hash_table: torch.Tensor       # flat
    gbv: torch.Tensor              # flat [R^3*4]
    gbw: torch.Tensor              # flat [R^3]
    gbv_res: int
    W1: torch.Tensor
    W2: torch.Tensor
    W3: torch.Tensor
    W4: torch.Tensor
    c_trunc: float
    trunc: float
    map_clamp: float = 1.0
    n_bins: int = 16
    pos_fp16: bool = False        # reference: fp32 OneBlob (model/encodings.py:73)


def query_color_sdf(fp: FieldParams, x01: torch.Tensor, clamp: bool = False) -> torch.Tensor:
    """model/scene_rep.py:314-349 on already-normalised points [B,3] -> raw [B,4]."""
    emb = grid_encode(x01, fp.hash_table, fp.hash_meta)
    pos = oneblob_encode(x01, fp.n_bins, fp.pos_fp16)
    ex = grid_encode(x01, fp.gbv, dense_meta(fp.gbv_res, 4))
    t = ex[..., 0] * fp.c_trunc
    t = t / fp.trunc
    if clamp:
        t = torch.clamp(t, -fp.map_clamp, fp.map_clamp)
        cin = torch.clamp(t, -1, 1)
    else:
        t = torch.clamp(t, -1, 1)
        cin = t
    raw = mlp_forward(emb, pos, cin.unsqueeze(-1), ex[..., 1:], fp.W1, fp.W2, fp.W3, fp.W4)
    rgb = raw[..., :3] + ex[..., 1:]
    sdf = raw[..., 3] + t
    return torch.cat([rgb, sdf.unsqueeze(-1)], -1)


def query_sdf_res(fp: FieldParams, x01: torch.Tensor, embed: bool = False) -> torch.Tensor:
    """model/scene_rep.py:212-248 (always clamps to +-1; embed=True returns raw hash features)."""
    emb = grid_encode(x01, fp.hash_table, fp.hash_meta)
    if embed:
        return emb
    pos = oneblob_encode(x01, fp.n_bins, fp.pos_fp16)
    ex = grid_encode(x01, fp.gbv, dense_meta(fp.gbv_res, 4))
    t = torch.clamp(ex[..., 0] * fp.c_trunc / fp.trunc, -1, 1)
    h = F.linear(torch.relu(F.linear(torch.cat([emb, pos, t.unsqueeze(-1)], -1), fp.W1)), fp.W2)
    return h[..., 0] + t


def query_w_res(fp: FieldParams, x01: torch.Tensor) -> torch.Tensor:
    """model/scene_rep.py:269-282."""
    return grid_encode(x01, fp.gbw, dense_meta(fp.gbv_res, 1))[..., 0]


def query_sdf_ex(fp: FieldParams, x01: torch.Tensor) -> torch.Tensor:
    """model/scene_rep.py:250-265."""
    return grid_encode(x01, fp.gbv, dense_meta(fp.gbv_res, 4))[..., 0]


def query_color_ex(fp: FieldParams, x01: torch.Tensor) -> torch.Tensor:
    """model/scene_rep.py:300-310."""
    return grid_encode(x01, fp.gbv, dense_meta(fp.gbv_res, 4))[..., 1:]


def query_color_residual(fp: FieldParams, x01: torch.Tensor) -> torch.Tensor:
    """model/scene_rep.py:285-298: note the decoder gets the *raw* GBV tsdf (no rescale/clamp)."""
    emb = grid_encode(x01, fp.hash_table, fp.hash_meta)
    pos = oneblob_encode(x01, fp.n_bins, fp.pos_fp16)
    ex = grid_encode(x01, fp.gbv, dense_meta(fp.gbv_res, 4))
    raw = mlp_forward(emb, pos, ex[..., :1], ex[..., 1:], fp.W1, fp.W2, fp.W3, fp.W4)
    return raw[..., :3] + ex[..., 1:]


# ----------------------------------------------------------------------------- sampler (S1)
def sample_z_vals(target_d: torch.Tensor, near: float, far: float, range_d: float, n_range_d: int,
                  n_samples_d: int, perturb: float, rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """model/scene_rep.py:415-441.  target_d [n,1].  rand: the U[0,1) draw (shape [n,S]) used when
    perturb>0 (the reference calls torch.rand; pass it in to compare deterministically)."""
    n = target_d.shape[0]
    z_samples = torch.linspace(-range_d, range_d, steps=n_range_d).to(target_d)
    z_samples = z_samples[None, :].repeat(n, 1) + target_d
    z_samples[target_d.squeeze(-1) <= 0] = torch.linspace(near, far, steps=n_range_d).to(target_d)
    if n_samples_d > 0:
        z_vals = torch.linspace(near, far, n_samples_d)[None, :].repeat(n, 1).to(target_d)
        z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
    else:
        z_vals = z_samples
    if perturb > 0.0:
        mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        if rand is None:
            rand = torch.rand(z_vals.shape)
        z_vals = lower + (upper - lower) * rand
    return z_vals


# ----------------------------------------------------------------------------- render (R1)
def sdf2weights(sdf: torch.Tensor, z_vals: torch.Tensor, trunc: float, sc_factor: float = 1.0) -> torch.Tensor:
    """model/scene_rep.py:107-127."""
    weights = torch.sigmoid(sdf / trunc) * torch.sigmoid(-sdf / trunc)
    signs = sdf[:, 1:] * sdf[:, :-1]
    mask = torch.where(signs < 0.0, torch.ones_like(signs), torch.zeros_like(signs))
    inds = torch.argmax(mask, dim=1)[..., None]
    z_min = torch.gather(z_vals, 1, inds)
    mask = torch.where(z_vals < z_min + sc_factor * trunc, torch.ones_like(z_vals), torch.zeros_like(z_vals))
    weights = weights * mask
    return weights / (torch.sum(weights, dim=-1, keepdim=True) + 1e-8)


def raw2outputs(raw: torch.Tensor, z_vals: torch.Tensor, trunc: float, sc_factor: float = 1.0):
    """model/scene_rep.py:156-179.  raw [n,S,4] -> rgb [n,3], depth [n]."""
    w = sdf2weights(raw[..., 3], z_vals, trunc, sc_factor)
    return torch.sum(w[..., None] * raw[..., :3], -2), torch.sum(w * z_vals, -1)


# ----------------------------------------------------------------------------- losses (L1)
def get_masks(z_vals, target_d, truncation):
    """model/utils.py:170-198."""
    front = torch.where(z_vals < (target_d - truncation), torch.ones_like(z_vals), torch.zeros_like(z_vals))
    back = torch.where(z_vals > (target_d + truncation), torch.ones_like(z_vals), torch.zeros_like(z_vals))
    dmask = torch.where(target_d > 0.0, torch.ones_like(target_d), torch.zeros_like(target_d))
    sdf_mask = (1.0 - front) * (1.0 - back) * dmask
    n_fs = torch.count_nonzero(front)
    n_sdf = torch.count_nonzero(sdf_mask)
    n = n_sdf + n_fs
    return front, sdf_mask, 1.0 - n_fs / n, 1.0 - n_sdf / n


def get_sdf_loss(z_vals, target_d, predicted_sdf, truncation, middle_mask=None):
    """model/utils.py:219-256 (loss_type='l2')."""
    front, sdf_mask, fs_w, sdf_w = get_masks(z_vals, target_d, truncation)
    if middle_mask is not None:
        front = front * middle_mask[..., None]
        sdf_mask = sdf_mask * middle_mask[..., None]
    fs_loss = F.mse_loss(predicted_sdf * front, torch.ones_like(predicted_sdf) * front) * fs_w
    sdf_loss = F.mse_loss((z_vals + predicted_sdf * truncation) * sdf_mask, target_d * sdf_mask) * sdf_w
    return fs_loss, sdf_loss


def mapping_losses(rgb_map, depth_map, raw, z_vals, target_rgb, target_d, *, depth_trunc: float,
                   rgb_missing: float, trunc: float, sc_factor: float = 1.0) -> Dict[str, torch.Tensor]:
    """model/scene_rep.py:493-527."""
    td = target_d.squeeze(-1)
    valid = (td > 0.0) * (td < depth_trunc)
    rgb_w = valid.clone().unsqueeze(-1)          # bool, like the reference: the assignment below casts to bool
    rgb_w[rgb_w == 0] = rgb_missing
    rgb_loss = F.mse_loss(rgb_map * rgb_w, target_rgb * rgb_w)
    depth_loss = F.mse_loss(depth_map[valid], td[valid])
    fs_loss, sdf_loss = get_sdf_loss(z_vals, target_d, raw[..., 3], trunc * sc_factor, middle_mask=valid)
    return {"rgb_res_loss": rgb_loss, "depth_res_loss": depth_loss, "sdf_res_loss": sdf_loss,
            "fs_res_loss": fs_loss}


def total_loss(ret: Dict[str, torch.Tensor], w: Dict[str, float], smooth: Optional[torch.Tensor] = None):
    """mp_slam/slam.py:162-178."""
    loss = w["rgb_weight"] * ret["rgb_res_loss"] + w["depth_weight"] * ret["depth_res_loss"] \
        + w["sdf_weight"] * ret["sdf_res_loss"] + w["fs_weight"] * ret["fs_res_loss"]
    if smooth is not None and w.get("smooth_weight", 0) > 0:
        loss = loss + w["smooth_weight"] * smooth
    return loss


# ----------------------------------------------------------------------------- TV smoothness (TV1)
def smoothness_points(bbox: torch.Tensor, sample_points: int, voxel_size: float, margin: float,
                      rand_offset: torch.Tensor, rand_jitter: torch.Tensor) -> torch.Tensor:
    """mp_slam/slam.py:197-207.  Returns normalised lattice points [P,P,P,3], P = sample_points-1.
    rand_offset [3], rand_jitter [1,1,1,3] are the two torch.rand draws."""
    P = sample_points - 1
    grid_size = P * voxel_size
    offset_max = bbox[:, 1] - bbox[:, 0] - grid_size - 2 * margin
    offset = rand_offset * offset_max + margin
    ar = torch.arange(0, P, dtype=torch.long)
    coords = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), dim=-1).float()
    pts = (coords + rand_jitter) * voxel_size + bbox[:, 0] + offset
    return (pts - bbox[:, 0]) / (bbox[:, 1] - bbox[:, 0])


def smoothness_from_features(feat: torch.Tensor, sample_points: int) -> torch.Tensor:
    """mp_slam/slam.py:211-215.  feat [P,P,P,C]."""
    tx = torch.pow(feat[1:, ...] - feat[:-1, ...], 2).sum()
    ty = torch.pow(feat[:, 1:, ...] - feat[:, :-1, ...], 2).sum()
    tz = torch.pow(feat[:, :, 1:, ...] - feat[:, :, :-1, ...], 2).sum()
    return (tx + ty + tz) / (sample_points ** 3)


# ----------------------------------------------------------------------------- full forward
def render_rays(fp: FieldParams, bbox: torch.Tensor, rays_o, rays_d, z_vals, clamp=False, sc_factor=1.0):
    """model/scene_rep.py:443-454 given z_vals: pts -> normalise (:388) -> Q1 -> R1."""
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    flat = pts.reshape(-1, 3)
    # a float64 bound promotes this to float64 (scene_rep.py:388); tinycudann then casts its input to fp32
    x01 = ((flat - bbox[:, 0]) / (bbox[:, 1] - bbox[:, 0])).to(torch.float32)
    raw = query_color_sdf(fp, x01, clamp).reshape(*pts.shape[:-1], 4)
    rgb, depth = raw2outputs(raw, z_vals, fp.trunc, sc_factor)
    return {"rgb_res_map": rgb, "depth_res_map": depth, "z_vals": z_vals, "raw": raw}
