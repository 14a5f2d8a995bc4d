"""GPU parity: librfx field / render kernels (through the C ABI and the Python mirror) vs the torch
CPU oracle.  fp32 tolerances: forward rel 1e-4; gradients per element rel 1e-4 + 4x the oracle's own
fp32 summation noise (_grad_close); the size the bench runs at is covered by tests/test_timed_path_gpu.py."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import field_oracle as FO  # noqa: E402


def _close(got, ref, rtol, atol, what):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    lim = atol + rtol * ref.abs()
    if not bool((err <= lim).all()):
        i = int(torch.argmax(err - lim))
        raise AssertionError(f"{what}: max err {err.max():.3e} (ref scale {ref.abs().max():.3e}); worst "
                             f"got {got.reshape(-1)[i]:.6e} ref {ref.reshape(-1)[i]:.6e}")


def _grad_close(got, ref32, ref64, what, groups=None, rtol=1e-4, k=4.0, tie_bound=None):
    """gradient check per ELEMENT: |got - ref64| <= rtol |ref64| + k * noise, where ref64 is the oracle evaluated with
    float64 parameters (same fp32 inputs, hence the same cells and weights) and noise is the oracle's OWN fp32 error,
    max |ref32 - ref64| over the element's group (a hash level / a weight matrix): the floor is the size of fp32
    summation noise where the element lives, not a fraction of the tensor's largest entry.

    tie_bound (decoder weight gradients dW1 / dW3): a hidden pre-activation within rounding of zero takes the other branch of
    the ReLU when its sum is formed in another order -- the kernels form it as W.X^T in the forward and as X.W^T in the
    weight-gradient pass, torch's GEMMs in a third way, the reference's cuBLAS in a fourth -- and that sample's whole term
    enters or leaves one row of the gradient: an error of ONE term, typically 1e-4 of the matrix's largest entry, far above
    the summation noise (tools/dw_noise_stats.py).  Round 5: instead of excusing a number of elements, the ORACLE says which
    (sample, hidden unit) pairs can tie at all (field_oracle.relu_tie_bounds: |exact pre-activation| within the worst-case
    fp32 dot-product error) and `tie_bound[r, i]` is the summed magnitude of exactly those samples' terms in element (r, i):
    it is added to that element's limit, and is zero for every row without a possible tie."""
    got, r32, r64 = got.detach().cpu().double().reshape(-1), ref32.detach().cpu().double().reshape(-1), ref64.detach().cpu().double().reshape(-1)
    groups = groups or [(0, got.numel())]
    for a, b in groups:
        noise = float((r32[a:b] - r64[a:b]).abs().max())
        err = (got[a:b] - r64[a:b]).abs()
        lim = rtol * r64[a:b].abs() + k * noise + 1e-30
        if tie_bound is not None:
            lim = lim + tie_bound.detach().cpu().double().reshape(-1)[a:b]
        if not bool((err <= lim).all()):
            i = int(torch.argmax(err - lim))
            raise AssertionError(f"{what} [{a}:{b}]: err {float(err[i]):.3e} > {float(lim[i]):.3e} (ref {float(r64[a + i]):.6e}, got "
                                 f"{float(got[a + i]):.6e}, oracle fp32 noise of the group {noise:.3e})")


class _probe:
    """with _probe() as pr: <float64 oracle forward + backward>; pr.bounds() -> {"dW1": [32,81], "dW3": [32,66]} (see _grad_close)"""
    def __enter__(self):
        FO.MLP_PROBE = self.d = {}
        return self

    def __exit__(self, *a):
        FO.MLP_PROBE = None

    def bounds(self):
        (b1, n1), (b3, n3) = FO.relu_tie_bounds(self.d)
        self.n_ties = (n1, n3)
        return {"dW1": b1, "dW3": b3}


def _level_groups(meta):
    return [(meta.offsets[l] * meta.n_feat, meta.offsets[l + 1] * meta.n_feat) for l in range(meta.n_levels)]


def _f64_params(fp):
    """the oracle's parameters in float64 (inputs stay fp32: same cells, same interpolation weights)"""
    import copy
    q = copy.copy(fp)
    for k in ("hash_table", "W1", "W2", "W3", "W4"):
        setattr(q, k, getattr(fp, k).detach().double().requires_grad_(True))
    q.gbv, q.gbw = fp.gbv.double(), fp.gbw.double()
    return q


def _model(name="office0", hash_scale=0.5, seed=0, gbv_fill=True):
    """JointEncoding on the GPU with deterministic, non-trivial parameters + the matching oracle params."""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.scene_rep import JointEncoding
    cfg = synthetic_config(name)
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    torch.manual_seed(seed)
    m = JointEncoding(cfg, bb, num_kf=8).cuda()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        m.embed_res_fn.params.copy_(((torch.rand(m.embed_res_fn.params.shape, generator=g) * 2 - 1) * hash_scale).cuda())
        if gbv_fill:
            R = cfg["globalV"]["base_resolution"]
            gv = torch.rand((R ** 3, 4), generator=g)
            gv[:, 0] = gv[:, 0] * 2.4 - 1.2          # tsdf in c_trunc units, exercises both clamps
            m.GBV.params.copy_(gv.reshape(-1).cuda())
            m.GBW.params.copy_(torch.rand(R ** 3, generator=g).cuda())
    return cfg, m


def _oracle_params(cfg, m):
    w1, w2, w3, w4 = (w.detach().cpu().clone() for w in m.decoder_res.fused_weights())
    meta = FO.hashgrid_meta_from_config(cfg["grid"]["hash_size"], m.resolution_sdf)
    return FO.FieldParams(hash_meta=meta, hash_table=m.embed_res_fn.params.detach().cpu().clone(),
                          gbv=m.GBV.params.detach().cpu().clone(), gbw=m.GBW.params.detach().cpu().clone(),
                          gbv_res=cfg["globalV"]["base_resolution"], W1=w1, W2=w2, W3=w3, W4=w4,
                          c_trunc=cfg["training"]["c_trunc"], trunc=cfg["training"]["trunc"],
                          map_clamp=cfg["mapping"]["clamp"], n_bins=16, pos_fp16=False)     # reference: fp32 OneBlob


def _points(n, seed=0, lo=-0.15, hi=1.15):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n, 3), generator=g) * (hi - lo) + lo
    x[:7] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [1.0, 0.0, 0.3], [0.999999, 0.5, 0.0],
                          [-0.3, 1.4, 0.2], [0.25, 0.75, 1.0]])
    return x


def test_grid_meta_matches_oracle_and_survey_tables():
    from remixfusion_amd.model.encodings import make_grid_desc
    for T, R, n_params in ((16, 325, None), (16, 400, None), (19, 450, None), (21, 1750, None)):
        meta = FO.hashgrid_meta_from_config(T, R)
        pls = float(np.exp2(np.log2(R / 16) / 15))
        d, n = make_grid_desc(16, 2, T, 16, pls, True)
        assert n == meta.offsets[-1]
        for l in range(16):
            assert d.res[l] == meta.res[l] and d.size[l] == meta.sizes[l] and d.offset[l] == meta.offsets[l]
            assert bool(d.hashed[l]) == meta.hashed[l] and abs(d.scale[l] - meta.scales[l]) < 1e-6
    # office0 R=325 -> 1.64e6 params; scene0000 R=450,T=19 -> 1.02e7
    assert abs(FO.hashgrid_meta_from_config(16, 325).n_params - 1.64e6) < 2e4
    assert abs(FO.hashgrid_meta_from_config(19, 450).n_params - 1.02e7) < 2e5


def test_hash_and_dense_grid_forward():
    cfg, m = _model()
    fp = _oracle_params(cfg, m)
    x = _points(4099)
    ref = FO.grid_encode(x, fp.hash_table, fp.hash_meta)
    got = m.embed_res_fn(x.cuda())
    _close(got, ref, 1e-5, 1e-6, "hash grid")
    ref4 = FO.grid_encode(x, fp.gbv, FO.dense_meta(fp.gbv_res, 4))
    _close(m.GBV(x.cuda()), ref4, 1e-5, 1e-6, "GBV")
    ref1 = FO.grid_encode(x, fp.gbw, FO.dense_meta(fp.gbv_res, 1))
    _close(m.GBW(x.cuda()), ref1, 1e-5, 1e-6, "GBW")
    _close(m.query_w_res(x.cuda().view(-1, 1, 3)).reshape(-1), ref1[:, 0], 1e-5, 1e-6, "query_w_res")
    _close(m.query_sdf_ex(x.cuda()), ref4[:, 0], 1e-5, 1e-6, "query_sdf_ex")
    _close(m.query_color_ex(x.cuda()), ref4[:, 1:], 1e-5, 1e-6, "query_color_ex")
    assert m.embed_res_fn(x[:0].cuda()).shape == (0, 32)       # empty input


@pytest.mark.parametrize("name", ["scene0000"])
def test_hash_grid_forward_large_table(name):
    cfg, m = _model(name, gbv_fill=False)
    fp = _oracle_params(cfg, m)
    x = _points(2000, seed=3)
    _close(m.embed_res_fn(x.cuda()), FO.grid_encode(x, fp.hash_table, fp.hash_meta), 1e-5, 1e-6, "hash grid T=19")


def test_oneblob_forward():
    from remixfusion_amd.model.encodings import OneBlob
    x = _points(3000, seed=1, lo=-0.05, hi=1.05)
    for fp16, tol in ((False, 2e-6), (True, 5e-4)):     # fp16: one half-precision ulp near 1 is 4.9e-4
        got = OneBlob(16, fp16=fp16)(x.cuda())
        ref = FO.oneblob_encode(x, 16, pos_fp16=fp16)
        _close(got, ref, 0, tol, f"oneblob fp16={fp16}")
        assert abs(float(got.sum(1).mean()) - 3.0) < 1e-2   # each dim's bins integrate to 1
    # the fp16 path evaluates only the four unsaturated boundaries per coordinate (and falls back to all sixteen
    # outside [-0.5, 1.5)): identical to rounding the full evaluation, except where a boundary just past the kernel
    # edge rounds one ulp short of saturation there and leaks 2^-24 (one fp16 subnormal) into the next bin
    g = torch.Generator().manual_seed(8)
    xw = torch.rand((20000, 3), generator=g) * 2.6 - 0.8
    xw[:9, 0] = torch.tensor([0.0, 1.0, -1e-9, 0.0625, 0.5, 0.99999994, -0.5, 1.5, 1.4999999])
    full = OneBlob(16, fp16=False)(xw.cuda()).half().float()
    fast = OneBlob(16, fp16=True)(xw.cuda())
    diff = (fast - full).abs()
    assert float(diff.max()) <= 2.0 ** -24 and float((diff > 0).float().mean()) < 1e-4
    assert int((fast != 0).sum(1).max()) <= 12


@pytest.mark.parametrize("clamp,pos_fp16", [(False, False), (True, False), (True, True)])
def test_field_forward_matches_oracle(clamp, pos_fp16):
    """pos_fp16=False: the default and the reference's precision (fp32 OneBlob, model/encodings.py:73; every operand of
    the MLP in fp32).  pos_fp16=True: the explicit opt-in (OneBlob rounded to fp16, on the fp16 matrix pipe with
    hi/lo-split weights), compared with an oracle that applies the same rounding."""
    cfg, m = _model()
    cfg["mapping"]["clamp"] = 1.5
    fp = _oracle_params(cfg, m)
    fp.map_clamp = 1.5
    assert m.embedpos_fn.fp16 is False and fp.pos_fp16 is False and m._field_desc(False).pos_fp16 == 0     # the defaults
    fp.pos_fp16 = pos_fp16
    m.embedpos_fn.fp16 = pos_fp16
    for n in (1, 63, 64, 65, 1000, 4133):
        x = _points(max(n, 7), seed=n)[:n]
        m.clamp = clamp
        got = m.query_color_sdf(x.cuda())
        ref = FO.query_color_sdf(fp, x, clamp)
        _close(got, ref, 1e-4, 2e-5, f"raw4 n={n} clamp={clamp}")


def test_field_forward_points_far_outside_the_bound():
    """OneBlob's sparse fp32 form (three bins per coordinate) holds for x in [-0.5, 1.5); a wave that contains a point
    beyond that takes the dense evaluation.  Mixed and all-far batches, vs the fp32 oracle."""
    cfg, m = _model()
    fp = _oracle_params(cfg, m)
    for lo, hi, seed in ((-1.3, 2.3, 21), (1.55, 2.4, 22), (-0.49, 1.49, 23)):
        x = _points(777, seed=seed, lo=lo, hi=hi)[7:]
        x[5] = torch.tensor([1.5, -0.5, 0.3])
        x[6] = torch.tensor([1.4999999, -0.50000006, 2.0])
        _close(m.query_color_sdf(x.cuda()), FO.query_color_sdf(fp, x, False), 1e-4, 2e-5, f"raw4 far points [{lo},{hi}]")


def test_point_queries_match_oracle():
    cfg, m = _model()
    fp = _oracle_params(cfg, m)
    x = _points(1500, seed=9)
    _close(m.query_sdf_res(x.cuda().view(-1, 1, 3)).reshape(-1), FO.query_sdf_res(fp, x), 1e-4, 2e-5, "query_sdf_res")
    _close(m.query_color_residual(x.cuda()), FO.query_color_residual(fp, x), 1e-4, 2e-5, "query_color_residual")
    emb = m.query_sdf_res(x.cuda().view(10, 150, 3), embed=True)
    assert emb.shape == (10, 150, 32)
    _close(emb.reshape(-1, 32), FO.query_sdf_res(fp, x, embed=True), 1e-5, 1e-6, "embed=True")


# n >= 4096: LDS-privatised scatter; 7, 33, 65, 129: ragged 32-point batches / 64-point tiles of the staged rows
@pytest.mark.parametrize("clamp,n", [(False, 3001), (True, 3001), (True, 9001), (False, 7), (True, 33), (False, 65), (True, 129)])
def test_field_backward_matches_autograd_of_oracle(clamp, n):
    cfg, m = _model(hash_scale=0.5)
    cfg["mapping"]["clamp"] = 1.5
    fp = _oracle_params(cfg, m)
    fp.map_clamp = 1.5
    x = _points(n, seed=5, lo=0.02, hi=0.98)
    g = torch.Generator().manual_seed(11)
    draw = torch.randn((n, 4), generator=g)
    # oracle grads
    xo = x.clone().requires_grad_(True)
    for t in (fp.hash_table, fp.W1, fp.W2, fp.W3, fp.W4):
        t.requires_grad_(True)
    FO.query_color_sdf(fp, xo, clamp).backward(draw)
    # HIP grads
    m.clamp = clamp
    for p in m.parameters():
        p.grad = None
    xg = x.cuda().requires_grad_(True)
    m.query_color_sdf(xg).backward(draw.cuda())
    # the same gradients with float64 parameters: the yardstick, and (by difference) the oracle's own fp32 noise
    fq = _f64_params(fp)
    xq = x.clone().requires_grad_(True)
    FO.query_color_sdf(fq, xq, clamp).backward(draw.double())
    w1, w2, w3, w4 = m.decoder_res.fused_weights()
    for got, r32, r64, nm in ((w1.grad, fp.W1.grad, fq.W1.grad, "dW1"), (w2.grad, fp.W2.grad, fq.W2.grad, "dW2"),
                              (w3.grad, fp.W3.grad, fq.W3.grad, "dW3"), (w4.grad, fp.W4.grad, fq.W4.grad, "dW4")):
        _grad_close(got, r32, r64, nm)
    _grad_close(m.embed_res_fn.params.grad, fp.hash_table.grad, fq.hash_table.grad, "d_hash", _level_groups(fp.hash_meta))
    if n >= 1000:
        assert float((m.embed_res_fn.params.grad != 0).float().mean()) > 0.01
    _grad_close(xg.grad, xo.grad, xq.grad, "dx01", k=8.0)
    assert m.GBV.params.grad is None


def test_field_backward_with_few_rows_that_carry_a_gradient():
    """40 001 points of which 600 have a non-zero d_raw: the backward selects those rows on the device (n_sel), so the table
    scatter's sweep -- planned on the host for ~0.65 n rows, 40 staged rows per thread -- finds ONE staged row: most of its
    (segment, part) blocks get no rows at all and must leave their segment untouched.  Against autograd through the oracle."""
    n = 40001
    cfg, m = _model(hash_scale=0.5)
    fp = _oracle_params(cfg, m)
    x = _points(n, seed=21, lo=0.02, hi=0.98)
    g = torch.Generator().manual_seed(22)
    draw = torch.zeros((n, 4))
    rows = torch.randperm(n, generator=g)[:600]
    draw[rows] = torch.randn((600, 4), generator=g)
    for t in (fp.hash_table, fp.W1, fp.W2, fp.W3, fp.W4):
        t.requires_grad_(True)
    FO.query_color_sdf(fp, x.clone(), False).backward(draw)
    m.clamp = False
    for p in m.parameters():
        p.grad = None
    m.query_color_sdf(x.cuda()).backward(draw.cuda())
    fq = _f64_params(fp)
    FO.query_color_sdf(fq, x.clone(), False).backward(draw.double())
    w1, w2, w3, w4 = m.decoder_res.fused_weights()
    for got, r32, r64, nm in ((w1.grad, fp.W1.grad, fq.W1.grad, "dW1"), (w2.grad, fp.W2.grad, fq.W2.grad, "dW2"),
                              (w3.grad, fp.W3.grad, fq.W3.grad, "dW3"), (w4.grad, fp.W4.grad, fq.W4.grad, "dW4")):
        _grad_close(got, r32, r64, nm)
    _grad_close(m.embed_res_fn.params.grad, fp.hash_table.grad, fq.hash_table.grad, "d_hash", _level_groups(fp.hash_meta))
    assert float((m.embed_res_fn.params.grad != 0).float().sum()) > 600


@pytest.mark.parametrize("name,n", [("office0", 2000), ("office0", 12000), ("scene0000", 12000), ("cafeteria", 9000)])
def test_grid_encode_backward_standalone(name, n):
    """n < 4096: direct atomics; n >= 4096: LDS-privatised scatter (T = 2^16: <= 4 segments / level, 2^19: 32, 2^21: 128)."""
    cfg, m = _model(name, gbv_fill=False)
    fp = _oracle_params(cfg, m)
    x = _points(n, seed=2, lo=-0.05, hi=1.05)           # a few points outside the unit cube: dense levels wrap
    g = torch.Generator().manual_seed(4)
    dy = torch.randn((n, 32), generator=g)
    xo = x.clone().requires_grad_(True)
    fp.hash_table.requires_grad_(True)
    FO.grid_encode(xo, fp.hash_table, fp.hash_meta).backward(dy)
    xg = x.cuda().requires_grad_(True)
    m.embed_res_fn.params.grad = None
    m.embed_res_fn(xg).backward(dy.cuda())
    t64 = fp.hash_table.detach().double().requires_grad_(True)
    xq = x.clone().requires_grad_(True)
    FO.grid_encode(xq, t64, fp.hash_meta).backward(dy.double())
    _grad_close(m.embed_res_fn.params.grad, fp.hash_table.grad, t64.grad, "dtable", _level_groups(fp.hash_meta))
    _grad_close(xg.grad, xo.grad, xq.grad, "dx", k=8.0)


@pytest.mark.parametrize("name,n", [("office0", 6000), ("scene0000", 5000)])
def test_grid_encode_backward_far_outside_the_cube(name, n):
    """points up to twelve cube lengths outside, and some as far out as it takes for the cell's x + 1 to reach the segment bits
    of the fine hashed levels, so that the two corners of an x-pair can fall into different table segments -- the sweep's per-corner fallback (the wave-uniform fast path
    tests one corner per pair; round 6) and the binned sort's one-corner records; dense levels wrap modulo their size.  Few rows
    per thread as well (n / 1 024 = 5-6: fewer than some levels' parts)."""
    cfg, m = _model(name, gbv_fill=False)
    fp = _oracle_params(cfg, m)
    x = _points(n, seed=12, lo=-1.0, hi=12.0)
    x[: n // 2] = _points(n // 2, seed=13, lo=0.0, hi=1.0)      # half of them inside: the waves mix both kinds
    # ... and, for every level fine enough, forty points whose cell is x = 8 191 exactly: x + 1 = 8 192 is the first segment bit
    desc = m.embed_res_fn.desc
    k = n // 2
    for l in range(int(desc.n_levels)):
        sc = float(desc.scale[l])
        xs = (8191.3 - 0.5) / sc
        if int(desc.hashed[l]) and xs < 80.0:          # (25-70 cube lengths out at office0 sizes: only a fallback ever sees them)
            x[k:k + 40, 0] = xs
            k += 40
    assert k > n // 2 + 80, "no level fine enough to straddle: the test would not reach the fallback"
    x = x[torch.randperm(n, generator=torch.Generator().manual_seed(14))].contiguous()
    g = torch.Generator().manual_seed(15)
    dy = torch.randn((n, 32), generator=g)
    fp.hash_table.requires_grad_(True)
    FO.grid_encode(x.clone(), fp.hash_table, fp.hash_meta).backward(dy)
    t64 = fp.hash_table.detach().double().requires_grad_(True)
    FO.grid_encode(x.clone(), t64, fp.hash_meta).backward(dy.double())
    xg = x.cuda().requires_grad_(True)
    m.embed_res_fn.params.grad = None
    m.embed_res_fn(xg).backward(dy.cuda())
    _grad_close(m.embed_res_fn.params.grad, fp.hash_table.grad, t64.grad, "dtable", _level_groups(fp.hash_meta))


@pytest.mark.parametrize("name", ["scene0000", "cafeteria"])
def test_binned_scatter_is_the_same_one_level_or_all_levels_at_a_time(name):
    """T >= 2^19: the binned levels go through their four kernels in groups as large as the workspace allows --
    rfx_grid_encode_backward_workspace_bytes (the minimum): one level per group; ..._workspace_bytes_for(grid): all at once.
    Both against the oracle per element, and against each other to the order of the final float atomics."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    n = 9000
    cfg, m = _model(name, gbv_fill=False)
    fp = _oracle_params(cfg, m)
    enc = m.embed_res_fn
    x = _points(n, seed=5, lo=0.0, hi=1.0)
    g = torch.Generator().manual_seed(6)
    dy = torch.randn((n, 32), generator=g)
    fp.hash_table.requires_grad_(True)
    FO.grid_encode(x.clone(), fp.hash_table, fp.hash_meta).backward(dy)
    t64 = fp.hash_table.detach().double().requires_grad_(True)
    FO.grid_encode(x.clone(), t64, fp.hash_meta).backward(dy.double())
    small = int(lib.rfx_grid_encode_backward_workspace_bytes(n, 16))
    big = int(lib.rfx_grid_encode_backward_workspace_bytes_for(C.byref(enc.desc), n))
    assert big > 4 * small                              # several binned levels
    xg, dyg = x.cuda().contiguous(), dy.cuda().contiguous()
    got = []
    for nb in (small, big, (small + big) // 2 // 16 * 16):
        ws = torch.empty(nb // 4, device="cuda")
        dt = torch.zeros_like(enc.params)
        L.check(lib.rfx_grid_encode_backward(enc.desc, L.ptr(enc.params), L.ptr(xg), n, L.ptr(dyg), L.ptr(dt), None, L.ptr(ws), nb,
                                             L.stream_ptr(xg.device)), "rfx_grid_encode_backward")
        torch.cuda.synchronize()
        _grad_close(dt.view_as(enc.params), fp.hash_table.grad, t64.grad, f"dtable ({nb} B of workspace)", _level_groups(fp.hash_meta))
        got.append(dt)
    for other in got[1:]:
        assert float((got[0] - other).abs().max()) <= 1e-5 * float(got[0].abs().max())


def _rays(n, cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.1, -0.6, 0.2]) + 0.05 * torch.randn((n, 3), generator=g)
    d = torch.randn((n, 3), generator=g) * 0.4
    d[:, 0] = 1.0
    td = torch.rand((n, 1), generator=g) * 2.5 + 0.3
    td[::7] = 0.0                                  # rays without depth -> near..far sampling
    return o, d, td


@pytest.mark.parametrize("name", ["office0", "scene0000"])     # S = 59 and S = 117
def test_sampler_points_and_compositing(name):
    cfg, m = _model(name, gbv_fill=False)
    tr, cam = cfg["training"], cfg["cam"]
    S = tr["n_range_d"] + tr["n_samples_d"]
    n = 257
    o, d, td = _rays(n, cfg)
    torch.manual_seed(123)
    z = m.sample_z_vals(td.cuda(), n, torch.device("cuda"))
    torch.manual_seed(123)
    u = torch.rand((n, S), device="cuda").cpu()
    z_ref = FO.sample_z_vals(td, cam["near"], cam["far"], tr["range_d"], tr["n_range_d"], tr["n_samples_d"], tr["perturb"], u)
    _close(z, z_ref, 1e-6, 2e-6, "z_vals")
    # points
    from remixfusion_amd.model.scene_rep import _RayPointsFn
    x01 = _RayPointsFn.apply(o.cuda(), d.cuda(), z, m)
    bb = m.bounding_box
    pts = o[:, None, :] + d[:, None, :] * z.cpu()[..., None]
    ref = ((pts.reshape(-1, 3) - bb[:, 0]) / (bb[:, 1] - bb[:, 0])).float()
    _close(x01, ref, 0, 1e-6, "x01")
    # compositing forward/backward on crafted raw: includes no-crossing, all-negative, crossing at the end
    g = torch.Generator().manual_seed(5)
    raw = torch.rand((n, S, 4), generator=g)
    raw[..., 3] = torch.linspace(1.0, -1.0, S)[None, :] * (0.5 + torch.rand((n, 1), generator=g)) + 0.02 * torch.randn((n, S), generator=g)
    raw[0, :, 3] = 0.7
    raw[1, :, 3] = -0.4
    raw[2, :, 3] = 0.5
    raw[2, -1, 3] = -0.5
    raw_g = raw.cuda().requires_grad_(True)
    rgb, dep = m.raw2outputs(raw_g, z)
    raw_o = raw.clone().requires_grad_(True)
    rgb_ref, dep_ref = FO.raw2outputs(raw_o, z.cpu(), tr["trunc"], cfg["data"]["sc_factor"])
    _close(rgb, rgb_ref, 1e-4, 1e-5, "rgb_map")
    _close(dep, dep_ref, 1e-4, 1e-5, "depth_map")
    gr, gd = torch.randn((n, 3), generator=g), torch.randn((n,), generator=g)
    (rgb * gr.cuda()).sum().add((dep * gd.cuda()).sum()).backward()
    (rgb_ref * gr).sum().add((dep_ref * gd).sum()).backward()
    raw_q = raw.double().requires_grad_(True)           # float64 oracle: the yardstick, and by difference the fp32 noise
    rgb_q, dep_q = FO.raw2outputs(raw_q, z.cpu().double(), tr["trunc"], cfg["data"]["sc_factor"])
    (rgb_q * gr.double()).sum().add((dep_q * gd.double()).sum()).backward()
    _grad_close(raw_g.grad, raw_o.grad, raw_q.grad, "d_raw", k=8.0)
    w = m.sdf2weights(raw[..., 3].cuda(), z)
    _close(w, FO.sdf2weights(raw[..., 3], z.cpu(), tr["trunc"], cfg["data"]["sc_factor"]), 1e-4, 1e-6, "weights")


@pytest.mark.parametrize("name", ["office0", "scene0000"])
def test_fused_render_matches_oracle(name):
    cfg, m = _model(name)
    fp = _oracle_params(cfg, m)
    tr, cam = cfg["training"], cfg["cam"]
    S = tr["n_range_d"] + tr["n_samples_d"]
    n = 150
    o, d, td = _rays(n, cfg, seed=3)
    torch.manual_seed(77)
    rgb, dep = m.render_fused(o.cuda(), d.cuda(), td.cuda())
    torch.manual_seed(77)
    u = torch.rand((n, S), device="cuda").cpu()
    z = FO.sample_z_vals(td, cam["near"], cam["far"], tr["range_d"], tr["n_range_d"], tr["n_samples_d"], tr["perturb"], u)
    ref = FO.render_rays(fp, m.bounding_box, o, d, z, clamp=False, sc_factor=cfg["data"]["sc_factor"])
    _close(rgb, ref["rgb_res_map"], 1e-4, 2e-5, "fused rgb")
    _close(dep, ref["depth_res_map"], 1e-4, 2e-5, "fused depth")


def test_mapping_losses_and_total_gradient_match_oracle():
    """JointEncoding.mapping() end to end (train mode) vs the oracle, including parameter grads."""
    cfg, m = _model(hash_scale=0.05)
    fp = _oracle_params(cfg, m)
    tr, cam = cfg["training"], cfg["cam"]
    S = tr["n_range_d"] + tr["n_samples_d"]
    n = 300
    o, d, td = _rays(n, cfg, seed=8)
    g = torch.Generator().manual_seed(2)
    tgt = torch.rand((n, 3), generator=g)
    m.train()
    torch.manual_seed(5)
    ret = m.mapping(o.cuda(), d.cuda(), tgt.cuda(), td.cuda())
    torch.manual_seed(5)
    u = torch.rand((n, S), device="cuda").cpu()
    z = FO.sample_z_vals(td, cam["near"], cam["far"], tr["range_d"], tr["n_range_d"], tr["n_samples_d"], tr["perturb"], u)
    for t in (fp.hash_table, fp.W1, fp.W2, fp.W3, fp.W4):
        t.requires_grad_(True)
    rend = FO.render_rays(fp, m.bounding_box, o, d, z, clamp=False, sc_factor=cfg["data"]["sc_factor"])
    ref = FO.mapping_losses(rend["rgb_res_map"], rend["depth_res_map"], rend["raw"], z, tgt, td,
                            depth_trunc=cam["depth_trunc"], rgb_missing=tr["rgb_missing"], trunc=tr["trunc"],
                            sc_factor=cfg["data"]["sc_factor"])
    for k in ("rgb_res_loss", "depth_res_loss", "sdf_res_loss", "fs_res_loss"):
        _close(ret[k], ref[k], 1e-4, 1e-7, k)
    w = {k: tr[k] for k in ("rgb_weight", "depth_weight", "sdf_weight", "fs_weight")}
    FO.total_loss(ref, w).backward()
    for p in m.parameters():
        p.grad = None
    FO.total_loss(ret, w).backward()
    fq = _f64_params(fp)
    rend64 = FO.render_rays(fq, m.bounding_box.cpu().double(), o.double(), d.double(), z.double(), clamp=False, sc_factor=cfg["data"]["sc_factor"])
    ref64 = FO.mapping_losses(rend64["rgb_res_map"], rend64["depth_res_map"], rend64["raw"], z.double(), tgt.double(), td.double(),
                              depth_trunc=cam["depth_trunc"], rgb_missing=tr["rgb_missing"], trunc=tr["trunc"],
                              sc_factor=cfg["data"]["sc_factor"])
    FO.total_loss(ref64, w).backward()
    for got, a, q, nm in zip(m.decoder_res.fused_weights(), (fp.W1, fp.W2, fp.W3, fp.W4), (fq.W1, fq.W2, fq.W3, fq.W4),
                             ("dL/dW1", "dL/dW2", "dL/dW3", "dL/dW4")):
        _grad_close(got.grad, a.grad, q.grad, nm)
    _grad_close(m.embed_res_fn.params.grad, fp.hash_table.grad, fq.hash_table.grad, "dL/dhash", _level_groups(fp.hash_meta))
    m.eval()
    out = m.mapping(o.cuda(), d.cuda(), tgt.cuda(), td.cuda())
    assert set(out) == {"rgb_res_map", "depth_res_map", "z_vals", "raw"}


@pytest.mark.parametrize("name", ["office0", "scene0000"])
def test_product_mapping_matches_reference_golden(name):
    """the HIP path against vectors produced by the reference's own JointEncoding.mapping()
    (tests/golden/mapping_*.npz, perturb=0): z_vals, raw, rgb/depth maps and the four losses."""
    import os
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.scene_rep import JointEncoding
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"mapping_{name}.npz"))
    cfg = synthetic_config(name)
    cfg["training"]["perturb"] = 0
    cfg["globalV"]["base_resolution"] = int(g["gbv_res"])
    cfg["grid"]["hash_size"] = int(g["hash_T"])
    cfg["grid"]["voxel_sdf"] = int(g["hash_R"])           # > 10: taken as the resolution itself (scene_rep.py:29-30)
    m = JointEncoding(cfg, torch.from_numpy(np.array(cfg["mapping"]["bound"])), num_kf=4).cuda()
    with torch.no_grad():
        m.embed_res_fn.params.copy_(torch.from_numpy(g["table"]).cuda())
        m.GBV.params.copy_(torch.from_numpy(g["gbv"]).cuda())
        for p, k in zip(m.decoder_res.fused_weights(), ("W1", "W2", "W3", "W4")):
            p.copy_(torch.from_numpy(g[k]).cuda())
    o, d, td, tgt = (torch.from_numpy(g[k]).cuda() for k in ("o", "d", "td", "tgt"))
    for clamp, tag in ((False, "c0"), (True, "c1")):
        m.eval()
        rend = m.mapping(o, d, tgt, td, clamp=clamp)
        _close(rend["z_vals"], torch.from_numpy(g[f"{tag}_z_vals"]), 0, 2e-6, "z_vals")
        _close(rend["raw"], torch.from_numpy(g[f"{tag}_raw"]), 1e-4, 2e-5, "raw")
        _close(rend["rgb_res_map"], torch.from_numpy(g[f"{tag}_rgb_res_map"]), 1e-4, 2e-5, "rgb map")
        _close(rend["depth_res_map"], torch.from_numpy(g[f"{tag}_depth_res_map"]), 1e-4, 2e-5, "depth map")
        m.train()
        ret = m.mapping(o, d, tgt, td, clamp=clamp)
        for k in ("rgb_res_loss", "depth_res_loss", "sdf_res_loss", "fs_res_loss"):
            _close(ret[k], torch.from_numpy(g[f"{tag}_{k}"]), 1e-4, 1e-8, k)
        # fused renderer on the same rays (no jitter)
        rgb, dep = m.render_fused(o, d, td, jitter=False)
        _close(rgb, torch.from_numpy(g[f"{tag}_rgb_res_map"]), 1e-4, 2e-5, "fused rgb") if not clamp else None


def test_fused_mapping_node_equals_unfused_path_and_tv_node():
    """_MappingFn / _SmoothFn (one autograd node each) against the per-kernel + torch-op formulation."""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.datasets import get_dataset
    from remixfusion_amd.mp_slam.slam import SLAM
    cfg, m = _model(hash_scale=0.05)
    n = 300
    o, d, td = _rays(n, cfg, seed=4)
    g = torch.Generator().manual_seed(6)
    tgt = torch.rand((n, 3), generator=g)
    m.train()
    tr = cfg["training"]
    w = {k: tr[k] for k in ("rgb_weight", "depth_weight", "sdf_weight", "fs_weight")}
    res = []
    for fn in (m.mapping, m.mapping_unfused):
        for p in m.parameters():
            p.grad = None
        og = o.cuda().requires_grad_(True)
        dg = d.cuda().requires_grad_(True)
        torch.manual_seed(9)
        ret = fn(og, dg, tgt.cuda(), td.cuda(), clamp=True)
        FO.total_loss(ret, w).backward()
        res.append((ret, m.embed_res_fn.params.grad.clone(), m.decoder_res.fused_weights()[2].grad.clone(), og.grad.clone(), dg.grad.clone()))
    (ra, ha, wa, oa, da), (rb, hb, wb, ob, db) = res
    for k in ("rgb_res_loss", "depth_res_loss", "sdf_res_loss", "fs_res_loss", "rgb_res", "depth_res"):
        _close(ra[k], rb[k], 1e-5, 1e-7, k)
    _close(ha, hb, 2e-3, 1e-3 * float(hb.abs().max()), "hash grad")
    _close(wa, wb, 2e-3, 1e-3 * float(wb.abs().max()), "W3 grad")
    _close(oa, ob, 5e-3, 5e-3 * float(ob.abs().mean()), "rays_o grad")
    _close(da, db, 5e-3, 5e-3 * float(db.abs().mean()), "rays_d grad")
    # TV node
    ds = get_dataset(cfg, device="cuda", n_frames=2)
    slam = SLAM(cfg, ds, m, torch.device("cuda"))
    torch.manual_seed(3)
    m.embed_res_fn.params.grad = None
    tv = slam.smoothness(tr["smooth_pts"], tr["smooth_vox"], margin=tr["smooth_margin"])
    tv.backward()
    ga = m.embed_res_fn.params.grad.clone()
    torch.manual_seed(3)
    P = tr["smooth_pts"] - 1
    u6 = torch.rand(6, device="cuda")                      # the six uniforms slam.smoothness drew
    pts01 = slam.smoothness_points_torch(u6, tr["smooth_pts"], tr["smooth_vox"], tr["smooth_margin"])
    m.embed_res_fn.params.grad = None
    tv_ref = slam.smoothness_unfused(pts01, tr["smooth_pts"])
    tv_ref.backward()
    _close(tv, tv_ref, 1e-5, 1e-9, "TV value")
    _close(ga, m.embed_res_fn.params.grad, 2e-3, 1e-3 * float(ga.abs().max()), "TV grad")
    # and against the oracle's formulation of the same lattice
    fp = _oracle_params(cfg, m)
    feat = FO.grid_encode(pts01.reshape(-1, 3).float().cpu(), fp.hash_table, fp.hash_meta).reshape(P, P, P, 32)
    _close(tv, FO.smoothness_from_features(feat, tr["smooth_pts"]), 1e-4, 1e-9, "TV vs oracle")
    # weight folded into the node; weighted sum of the four losses folded into the mapping node
    torch.manual_seed(3)
    m.embed_res_fn.params.grad = None
    tvw = slam.smoothness(tr["smooth_pts"], tr["smooth_vox"], margin=tr["smooth_margin"], weight=0.37)
    tvw.backward()
    _close(tvw, 0.37 * tv.detach(), 1e-6, 1e-12, "weighted TV")
    _close(m.embed_res_fn.params.grad, 0.37 * ga, 2e-3, 1e-3 * float(ga.abs().max()) * 0.37, "weighted TV grad")
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(9)
    ret = m.mapping(o.cuda(), d.cuda(), tgt.cuda(), td.cuda(), clamp=True)
    _close(ret["loss_weighted"], FO.total_loss(ret, w).detach(), 1e-6, 1e-9, "loss_weighted")
    assert slam.get_loss_from_ret(ret) is ret["loss_weighted"]
    ret["loss_weighted"].backward()
    _close(m.embed_res_fn.params.grad, ha, 2e-3, 1e-3 * float(ha.abs().max()), "hash grad through loss_weighted")


@pytest.mark.parametrize("name", ["office0", "cafeteria"])     # float64 bound / integer bound (jitter truncates, fp32 math)
def test_tv_lattice_kernel_matches_tensor_ops(name):
    from remixfusion_amd.datasets import get_dataset
    from remixfusion_amd.mp_slam.slam import SLAM
    from remixfusion_amd import _lib as L
    cfg, m = _model(name, gbv_fill=False)
    cfg["cam"].update({"H": 60, "W": 80})
    slam = SLAM(cfg, get_dataset(cfg, device="cuda", n_frames=2), m, torch.device("cuda"))
    tr = cfg["training"]
    P = tr["smooth_pts"] - 1
    u6 = torch.rand(6, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    ref = slam.smoothness_points_torch(u6, tr["smooth_pts"], tr["smooth_vox"], tr["smooth_margin"]).reshape(-1, 3)
    pts = torch.empty((P ** 3, 3), device="cuda")
    L.check(L.load().rfx_tv_lattice(L.ptr(u6), P, float(tr["smooth_vox"]), float(tr["smooth_margin"]), m._bbox6, m._bbox_f64, 1,
                                    L.ptr(pts), L.stream_ptr(pts.device)), "rfx_tv_lattice")
    assert (m._bbox_f64 == 1) == (name == "office0")
    assert float((pts - ref.float()).abs().max()) < 2e-7
    assert float(pts.min()) >= 0.0 and float(pts.max()) <= 1.0


def test_mapping_pipeline_learns_the_scene():
    """end-to-end sanity on a small synthetic stream: the mapping schedule drives the rendered depth /
    colour error down (quality smoke test; thresholds are loose on purpose)."""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 100, "sample": 512})
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
    pipe = MappingPipeline(cfg, n_frames=40, seed=1)
    frames = pipe.prefetch(list(range(31)))

    def err(i):
        b = frames[i]
        pipe.model.train()
        with torch.no_grad():
            rgb, dep = pipe.slam.render_single(i, b["depth"][None], b["rgb"][None], b["c2w"], b["direction"])
        return float((dep - b["depth"]).abs().mean()), float((rgb - b["rgb"]).abs().mean())

    pipe.track_frame(0, frames[0])
    pipe.mapper.init_mapvolume()
    d0, c0 = err(0)                                    # untrained field on an empty global volume
    pipe.mapper.first_frame_mapping({k: v for k, v in frames[0].items() if k != "rgb255"}, cfg["mapping"]["first_iters"])
    d1, c1 = err(0)
    for i in range(1, 31):
        pipe.step(i, frames[i])
    d2, c2 = err(25)                                   # a later frame, seen only through the keyframe schedule
    assert d1 < 0.5 * d0 and c1 < 0.7 * c0, (d0, d1, c0, c1)
    assert d2 < 0.10 and c2 < 0.12, (d2, c2)
    assert int(pipe.slam.mapping_idx[0]) == 25


def test_prestaged_weight_image_equals_in_kernel_staging():
    """rfx_field_stage_weights + the 16-byte copy prologue vs every block deriving the operand layout itself;
    the image is refreshed when the weights change in place."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    cfg, m = _model()
    lib = L.load()
    x = _points(3000, seed=3).cuda().contiguous()
    outs = []
    for staged in (True, False):
        d = m._field_desc(False)
        assert d.staged
        if not staged:
            d.staged = None
        raw = torch.empty((x.shape[0], 4), device="cuda")
        L.check(lib.rfx_field_forward(C.byref(d), L.ptr(x), x.shape[0], L.ptr(raw), L.stream_ptr(x.device)), "fwd")
        outs.append(raw)
    assert torch.equal(outs[0], outs[1])
    before = m.query_color_sdf(x).clone()
    with torch.no_grad():
        m.decoder_res.fused_weights()[2].mul_(1.5)          # in-place update, like an optimizer step
    after = m.query_color_sdf(x)
    assert float((after[:, :3] - before[:, :3]).detach().abs().max()) > 1e-4
    d = m._field_desc(False)
    d.staged = None
    raw = torch.empty((x.shape[0], 4), device="cuda")
    L.check(lib.rfx_field_forward(C.byref(d), L.ptr(x), x.shape[0], L.ptr(raw), L.stream_ptr(x.device)), "fwd")
    assert torch.equal(after, raw)


def test_half_passes_of_the_forward_tail_equal_full_passes():
    """Round 6: with more 64-point passes than the 2 048 wave slots the forward deals the passes behind the last full round out as
    32-point HALF passes (lane l and lane l + 32 serve one point: levels 0-7 / 8-15, OneBlob columns 5 + 4, one MFMA tile;
    csrc/rfx_field_mlp.h::mlp_forward_123_half).  Here: 131 072 + 4 827 points (151 half passes, the last one ragged) in ONE call
    against the same points in two calls that fit the slots (full passes only): raw to fp32 rounding of another summation order,
    the stashed hash features bit for bit (each level is looked up by one lane either way) -- and the oracle on a sample."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    cfg, m = _model(hash_scale=0.5)
    n0, n = 2048 * 64, 2048 * 64 + 4827
    x = _points(n, seed=9, lo=0.02, hi=0.98).cuda().contiguous()
    x[n0 + 5] = torch.tensor([1.7, -0.4, 0.5])                      # a point far outside the bound inside a half pass (OneBlob's wrap column)
    desc = m._field_desc(False)
    assert not bool(desc.pos_fp16)
    st = L.stream_ptr(x.device)

    def forward(xx, stash):
        k = xx.shape[0]
        raw = torch.full((k, 4), float("nan"), device="cuda")
        if not stash:
            L.check(lib.rfx_field_forward(C.byref(desc), L.ptr(xx), k, L.ptr(raw), st), "forward")
            return raw, None
        nb = int(lib.rfx_field_backward_workspace_bytes(k))
        ws = torch.full((nb // 4 + 16,), float("nan"), device="cuda")
        wsp = (ws.data_ptr() + 15) // 16 * 16
        L.check(lib.rfx_field_forward_stash(C.byref(desc), L.ptr(xx), k, L.ptr(raw), wsp, nb, st), "forward_stash")
        torch.cuda.synchronize()
        o = (wsp - ws.data_ptr()) // 4
        tiles = (k + 63) // 64
        emb = ws[o:o + tiles * 8 * 256].view(tiles, 8, 64, 4).permute(0, 2, 1, 3).reshape(tiles * 64, 32)[:k].clone()     # piece-major tiles -> rows
        return raw, emb

    for stash in (False, True):
        raw_all, emb_all = forward(x, stash)                          # tail in half passes
        raw_a, emb_a = forward(x[:n0].contiguous(), stash)            # 2 048 passes: every wave one full pass
        raw_b, emb_b = forward(x[n0:].contiguous(), stash)            # 76 passes
        torch.cuda.synchronize()
        assert bool(torch.isfinite(raw_all).all())
        assert torch.equal(raw_all[:n0], raw_a)                       # the full rounds are untouched
        ref = raw_b
        err = (raw_all[n0:] - ref).abs()
        assert float((err / (ref.abs() + 1e-3)).max()) < 2e-5, float((err / (ref.abs() + 1e-3)).max())
        if stash:
            assert torch.equal(emb_all[:n0], emb_a) and torch.equal(emb_all[n0:], emb_b)
    fp = _oracle_params(cfg, m)
    sel = torch.cat([torch.arange(n0, n0 + 300), torch.arange(n - 40, n)])
    want = FO.query_color_sdf(fp, x[sel].cpu())
    _close(raw_all[sel].cpu(), want, 1e-4, 2e-5, "half-pass raw vs oracle")


@pytest.mark.parametrize("fp16", [False, True])
def test_stashed_forward_and_chains_are_bit_identical_to_the_recomputing_ones(fp16):
    """rfx_field_forward_stash leaves the hash features in the workspace and the three _stashed chains read them instead
    of looking the table up again: same raw, same dW, same dx01 (bit for bit), same d_hash (atomic order aside)."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    cfg, m = _model(hash_scale=0.5)
    m.embedpos_fn.fp16 = fp16
    n = 9001                                       # ragged last tile of 64
    x = _points(n, seed=6, lo=0.02, hi=0.98).cuda().contiguous()
    draw = torch.randn((n, 4), generator=torch.Generator().manual_seed(12)).cuda().contiguous()
    desc = m._field_desc(False)
    assert bool(desc.pos_fp16) == fp16
    st = L.stream_ptr(x.device)
    nbytes = int(lib.rfx_field_backward_workspace_bytes(n))
    table = m.embed_res_fn.params

    def run(chain, stashed, want_w, want_hash, want_dx):
        ws = torch.full((nbytes // 4 + 16,), float("nan"), device="cuda")
        wsp = (ws.data_ptr() + 15) // 16 * 16
        raw = torch.empty((n, 4), device="cuda")
        if stashed:
            L.check(lib.rfx_field_forward_stash(C.byref(desc), L.ptr(x), n, L.ptr(raw), wsp, nbytes, st), "forward_stash")
        else:
            L.check(lib.rfx_field_forward(C.byref(desc), L.ptr(x), n, L.ptr(raw), st), "forward")
        dws = [torch.zeros_like(w) for w in m.decoder_res.fused_weights()]
        d_hash, dx = torch.zeros_like(table), torch.zeros((n, 3), device="cuda")
        L.check(chain(C.byref(desc), L.ptr(x), n, L.ptr(draw), wsp, nbytes, st), "chain")
        if want_w:
            L.check(lib.rfx_field_backward_weights(n, L.ptr(draw), *[L.ptr(g) for g in dws], wsp, nbytes, st), "weights")
        if want_hash or want_dx:
            L.check(lib.rfx_field_backward_scatter(C.byref(desc), L.ptr(x), n, L.ptr(d_hash) if want_hash else None,
                                                   L.ptr(dx) if want_dx else None, wsp, nbytes, st), "scatter")
        if want_dx:
            L.check(lib.rfx_field_backward_dx(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(dx), wsp, nbytes, st), "dx")
        torch.cuda.synchronize()
        return raw, dws, d_hash, dx

    for plain, stash, w, h, d in ((lib.rfx_field_backward_chain, lib.rfx_field_backward_chain_stashed, True, True, True),
                                  (lib.rfx_field_backward_chain_weights, lib.rfx_field_backward_chain_weights_stashed, True, True, False),
                                  (lib.rfx_field_backward_chain_inputs, lib.rfx_field_backward_chain_inputs_stashed, False, False, True)):
        r0, w0, h0, x0 = run(plain, False, w, h, d)
        r1, w1, h1, x1 = run(stash, True, w, h, d)
        assert torch.equal(r0, r1) and bool(torch.isfinite(r1).all())
        for a, b in zip(w0, w1):
            assert torch.equal(a, b) and bool(torch.isfinite(b).all())
        if w:
            assert float(w1[0].abs().max()) > 0
        assert bool(torch.isfinite(h1).all()) and float((h0 - h1).abs().max()) <= 1e-5 * max(float(h0.abs().max()), 1e-30)
        assert torch.equal(x0, x1) and bool(torch.isfinite(x1).all())
        if d:
            assert float(x1.abs().max()) > 0
    # a workspace that is too small or misaligned is refused, like everywhere else
    raw = torch.empty((n, 4), device="cuda")
    ws = torch.empty(nbytes // 4 + 16, device="cuda")
    assert lib.rfx_field_forward_stash(C.byref(desc), L.ptr(x), n, L.ptr(raw), ws.data_ptr(), nbytes - 16, st) == -4
    assert lib.rfx_field_forward_stash(C.byref(desc), L.ptr(x), n, L.ptr(raw), None, nbytes, st) == -4


@pytest.mark.parametrize("stashed", [False, True])
def test_backward_on_selected_points_equals_the_sum_over_unselected_halves(stashed):
    """From 16 384 points on, the chain stage puts the points with a non-zero loss gradient first and every later stage
    works on those only (a third of a mapping batch has d_raw == 0 exactly).  Check against the same launch cut into two
    halves that are too small for the selection: gradients are sums over points, so dW and d_hash must agree to fp32
    summation order and dx01 -- one value per point -- bit for bit, zeros included."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    cfg, m = _model(hash_scale=0.5)
    n = 20001
    x = _points(n, seed=8, lo=0.02, hi=0.98).cuda().contiguous()
    g = torch.Generator().manual_seed(13)
    draw = torch.randn((n, 4), generator=g)
    draw[torch.rand(n, generator=g) < 0.4] = 0.0                 # rows without a gradient, scattered through the batch
    draw[5000:5300] = 0.0                                         # ... and whole tiles of them
    draw = draw.cuda().contiguous()
    desc = m._field_desc(True)
    st = L.stream_ptr(x.device)
    table = m.embed_res_fn.params

    def run(xx, dd):
        k = xx.shape[0]
        nbytes = int(lib.rfx_field_backward_workspace_bytes(k))
        ws = torch.full((nbytes // 4 + 16,), float("nan"), device="cuda")
        wsp = (ws.data_ptr() + 15) // 16 * 16
        raw = torch.empty((k, 4), device="cuda")
        if stashed:
            L.check(lib.rfx_field_forward_stash(C.byref(desc), L.ptr(xx), k, L.ptr(raw), wsp, nbytes, st), "forward_stash")
        chain = lib.rfx_field_backward_chain_stashed if stashed else lib.rfx_field_backward_chain
        dws = [torch.zeros_like(w) for w in m.decoder_res.fused_weights()]
        d_hash, dx = torch.zeros_like(table), torch.full((k, 3), float("nan"), device="cuda")
        L.check(chain(C.byref(desc), L.ptr(xx), k, L.ptr(dd), wsp, nbytes, st), "chain")
        L.check(lib.rfx_field_backward_weights(k, L.ptr(dd), *[L.ptr(t) for t in dws], wsp, nbytes, st), "weights")
        L.check(lib.rfx_field_backward_scatter(C.byref(desc), L.ptr(xx), k, L.ptr(d_hash), L.ptr(dx), wsp, nbytes, st), "scatter")
        L.check(lib.rfx_field_backward_dx(C.byref(desc), L.ptr(xx), k, L.ptr(dd), L.ptr(dx), wsp, nbytes, st), "dx")
        torch.cuda.synchronize()
        return dws, d_hash, dx

    h = n // 2
    assert h < 16384 <= n
    fw, fh, fx = run(x, draw)
    aw, ah, ax = run(x[:h].contiguous(), draw[:h].contiguous())
    bw, bh, bx = run(x[h:].contiguous(), draw[h:].contiguous())
    assert torch.equal(fx, torch.cat([ax, bx])) and bool(torch.isfinite(fx).all())
    zero_rows = (draw == 0).all(dim=1)
    assert float(zero_rows.float().mean()) > 0.3 and bool((fx[zero_rows] == 0).all()) and float(fx[~zero_rows].abs().max()) > 0
    for f_, a_, b_ in zip(fw, aw, bw):
        ref = a_ + b_
        assert bool(torch.isfinite(f_).all()) and float((f_ - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    ref = ah + bh
    assert bool(torch.isfinite(fh).all()) and float((fh - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) and float(ref.abs().max()) > 0


def test_backward_with_no_gradient_anywhere_gives_exact_zeros():
    """every row of d_raw zero (a batch of rays that all ended in free space): nothing is selected, and all gradients,
    dx01 included, come out as exact zeros rather than stale workspace contents."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    cfg, m = _model(hash_scale=0.5)
    n = 17000
    x = _points(n, seed=9, lo=0.02, hi=0.98).cuda().contiguous()
    draw = torch.zeros((n, 4), device="cuda")
    desc = m._field_desc(False)
    st = L.stream_ptr(x.device)
    nbytes = int(lib.rfx_field_backward_workspace_bytes(n))
    ws = torch.full((nbytes // 4 + 16,), float("nan"), device="cuda")
    wsp = (ws.data_ptr() + 15) // 16 * 16
    dws = [torch.zeros_like(w) for w in m.decoder_res.fused_weights()]
    d_hash, dx = torch.zeros_like(m.embed_res_fn.params), torch.full((n, 3), float("nan"), device="cuda")
    L.check(lib.rfx_field_backward(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(d_hash), *[L.ptr(t) for t in dws], L.ptr(dx),
                                   wsp, nbytes, st), "backward")
    torch.cuda.synchronize()
    assert all(bool((t == 0).all()) for t in dws) and bool((d_hash == 0).all()) and bool((dx == 0).all())


@pytest.mark.parametrize("clamp", [False, True])
def test_backward_chain_variants_agree_with_the_full_chain(clamp):
    """rfx_field_backward_chain_weights (map phase: rows + d_emb) and rfx_field_backward_chain_inputs (pose phase: dX1
    only) stage exactly what the stages that may follow them read: same dW / d_hash, resp. same dx01, as the full chain."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    cfg, m = _model(hash_scale=0.5)
    n = 9001
    x = _points(n, seed=5, lo=0.02, hi=0.98).cuda().contiguous()
    draw = torch.randn((n, 4), generator=torch.Generator().manual_seed(11)).cuda().contiguous()
    desc = m._field_desc(clamp)
    st = L.stream_ptr(x.device)
    nbytes = int(lib.rfx_field_backward_workspace_bytes(n))
    table = m.embed_res_fn.params

    def run(chain, want_w, want_hash, want_dx):
        ws = torch.full((nbytes // 4 + 16,), float("nan"), device="cuda")       # a stage reading what was not staged shows
        wsp = (ws.data_ptr() + 15) // 16 * 16
        dws = [torch.zeros_like(w) for w in m.decoder_res.fused_weights()]
        d_hash, dx = torch.zeros_like(table), torch.zeros((n, 3), device="cuda")
        L.check(chain(C.byref(desc), L.ptr(x), n, L.ptr(draw), wsp, nbytes, st), "chain")
        if want_w:
            L.check(lib.rfx_field_backward_weights(n, L.ptr(draw), *[L.ptr(g) for g in dws], wsp, nbytes, st), "weights")
        if want_hash or want_dx:
            L.check(lib.rfx_field_backward_scatter(C.byref(desc), L.ptr(x), n, L.ptr(d_hash) if want_hash else None,
                                                   L.ptr(dx) if want_dx else None, wsp, nbytes, st), "scatter")
        if want_dx:
            L.check(lib.rfx_field_backward_dx(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(dx), wsp, nbytes, st), "dx")
        torch.cuda.synchronize()
        return dws, d_hash, dx

    fw, fh, fx = run(lib.rfx_field_backward_chain, True, True, True)
    ww, wh, _ = run(lib.rfx_field_backward_chain_weights, True, True, False)
    _, _, ix = run(lib.rfx_field_backward_chain_inputs, False, False, True)
    for a, b in zip(fw, ww):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    assert bool(torch.isfinite(wh).all()) and float((fh - wh).abs().max()) <= 1e-5 * float(fh.abs().max())    # atomic order
    assert float(fh.abs().max()) > 0
    assert torch.equal(fx, ix) and bool(torch.isfinite(ix).all()) and float(ix.abs().max()) > 0
