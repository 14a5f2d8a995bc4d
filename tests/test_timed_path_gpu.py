"""GPU parity of the path bench.py TIMES, at the size it times it, against the CPU oracle.

bench.py's mapper iterations go through ONE library call each (``rfx_ba_forward_backward``, csrc/rfx_ba.hip, issued by
mp_slam/direct.py): ray gather -> S1 sampler -> Q1 field (forward + stash) -> R1 -> L1 -> TV1 -> backward with the row
selection (>= 16 384 points), the stashed chains, the recomputing weight gradients and the LDS-sweep / binned scatter.
Here that call runs on a real iteration's batch (2 048 keyframe rays + the current frame's share, S samples each, the TV
lattice on) and every gradient it returns is compared PER ELEMENT with autograd through oracle/field_oracle.py
(reference: mp_slam/mapper.py:392-423, 458-499; model/scene_rep.py:460-529) fed the very ray batch the call drew
(``rfx_ba_workspace_layout``).  Tolerance: ``_grad_close`` = rel 1e-4 + 4x the oracle's own fp32 summation noise.
The full-frame fused renderer (grid-stride over 307 200 rays) is checked the same way on 2 000 random rays of the frame.
"""
import ctypes as C
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import draws_oracle as DO  # noqa: E402
from oracle import field_oracle as FO  # noqa: E402
from test_field_gpu import _close, _f64_params, _grad_close, _level_groups, _oracle_params, _probe  # noqa: E402


def _pipeline(name, n_frames, seed=1):
    """the bench's pipeline at the bench's camera and batch sizes; only the moving volume (not under test here) is coarser"""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config(name)
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 30})
    cfg["synthetic"]["tracker"] = False
    pipe = MappingPipeline(cfg, n_frames=n_frames + 8, seed=seed)
    frames = pipe.prefetch(list(range(n_frames)))
    pipe.start(frames[0])
    for i in range(1, n_frames):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    return cfg, pipe, frames


def _ws_fields(lib, B, n, S, P, n_feat, n_levels):
    """views of the iteration's intermediates inside the one-call workspace (include/rfx.h: rfx_ba_workspace_layout)"""
    off = (C.c_size_t * 15)()
    assert lib.rfx_ba_workspace_layout(n, S, P, n_feat, n_levels, off, 15) == 15
    base = (B.p.ws - B.t.ws.data_ptr()) // 4
    nt = P ** 3
    shapes = [("o", (n, 3)), ("d", (n, 3)), ("tgt", (n, 3)), ("td", (n,)), ("d_cam", (n, 3)), ("pidx", (n,)), ("z", (n, S)),
              ("x01", (n * S, 3)), ("raw", (n * S, 4)), ("rgb_map", (n, 3)), ("depth_map", (n,)), ("pts", (nt, 3)),
              ("feat", (nt, n_feat)), ("d_raw", (n * S, 4)), ("dx", (n * S, 3))]
    out = {}
    for (nm, shp), o in zip(shapes, off):
        numel = int(np.prod(shp))
        v = B.t.ws[base + o // 4: base + o // 4 + numel]
        out[nm] = (v.view(torch.int32) if nm == "pidx" else v).view(shp).clone()
    return out


def _oracle_iteration(fp, cfg, bbox, o, d, z, tgt, td, clamp, tv_pts=None):
    """render -> four losses -> weighted total (+ TV term): the reference's iteration body"""
    tr, cam = cfg["training"], cfg["cam"]
    dt = fp.W1.dtype                                   # float64 parameters: the losses are formed in float64 too
    o, d, z, tgt, td = (t.to(dt) for t in (o, d, z, tgt, td))
    rend = FO.render_rays(fp, bbox, o, d, z, clamp=clamp, sc_factor=cfg["data"]["sc_factor"])
    ls = FO.mapping_losses(rend["rgb_res_map"], rend["depth_res_map"], rend["raw"], z, tgt, td[:, None],
                           depth_trunc=cam["depth_trunc"], rgb_missing=tr["rgb_missing"], trunc=tr["trunc"],
                           sc_factor=cfg["data"]["sc_factor"])
    w = {k: tr[k] for k in ("rgb_weight", "depth_weight", "sdf_weight", "fs_weight", "smooth_weight")}
    tv = None
    if tv_pts is not None:
        P = int(tr["smooth_pts"]) - 1
        feat = FO.grid_encode(tv_pts, fp.hash_table, fp.hash_meta).reshape(P, P, P, -1)
        tv = FO.smoothness_from_features(feat, int(tr["smooth_pts"]))
    return rend, ls, FO.total_loss(ls, w, smooth=tv)


@pytest.mark.parametrize("name,frames", [("office0", 41), ("scene0000", 21)])
def test_one_call_ba_iteration_matches_oracle_at_bench_size(name, frames):
    """office0: T = 2^16 (LDS sweep), S = 59, 31^3 lattice, 8 keyframes -> 2 304 rays = 1.36e5 points (the bench's batch);
    scene0000: T = 2^19 (binned levels), S = 117, 63^3 lattice."""
    from remixfusion_amd import _lib as L
    lib = L.load()
    cfg, pipe, fr = _pipeline(name, frames)
    mp, model, slam = pipe.mapper, pipe.model, pipe.slam
    direct = mp._direct_iterations()
    assert direct is not None
    direct.stagewise_every = 0
    tr, m = cfg["training"], cfg["mapping"]
    S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
    enc = model.embed_res_fn
    nF, nL = enc.n_output_dims, int(enc.desc.n_levels)
    last = frames - 1
    b = fr[last]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    n_kf = len(mp.keyframe.frame_ids)
    n = int(m["sample"]) + max(int(m["sample"]) // n_kf, int(m["min_pixels_cur"]))
    assert n * S >= 16384                              # the row selection is active
    if name == "office0":
        assert n_kf == 8 and n == 2304 and n * S == 135936
    bbox = model.bounding_box.cpu()
    dev = slam.est_c2w_data.device if slam.est_c2w_data.is_cuda else torch.device("cuda:0")
    params = [enc.params] + list(model.decoder_res.fused_weights())

    # ------------------------------------------------------------------ map phase (clamp off, TV on, hash + decoder grads)
    poses = slam.est_c2w_data[0:last + 1:m["keyframe_every"]].clone().float().contiguous()
    poses_all = torch.cat([poses, slam.est_c2w_data[last:last + 1].float()], 0)[:n_kf + 1].contiguous()
    random.seed(3); torch.manual_seed(3)
    for p_ in params:
        p_.grad = None
    fp = _oracle_params(cfg, model)                   # before the call: same parameters (no optimizer step in between)
    lc = direct.map_gradients(cur, poses_all)
    torch.cuda.synchronize()
    got = [p_.grad.detach().clone() for p_ in params]
    B = direct._buffers(n, 0, dev)
    f = {k: v.cpu() for k, v in _ws_fields(lib, B, n, S, P, nF, nL).items()}
    # the batch itself: rays from the poses the call was given (mapper.py:407-409), sampler (scene_rep.py:421-441)
    pidx = f["pidx"].long()
    pa = poses_all.cpu()
    _close(f["o"], pa[pidx, :3, 3], 0, 1e-7, "rays_o")
    _close(f["d"], torch.sum(f["d_cam"][:, None, :] * pa[pidx, :3, :3], -1), 1e-6, 1e-6, "rays_d")
    cam = cfg["cam"]
    # the uniforms the call drew for itself (rfx_ba_desc.seed_u): restated by oracle/draws_oracle.py
    assert not direct.torch_draws and direct.last_seed_u
    u_z = torch.from_numpy(DO.uniform_draws(direct.last_seed_u, 0, n * S)).view(n, S)
    z_ref = FO.sample_z_vals(f["td"][:, None], cam["near"], cam["far"], tr["range_d"], tr["n_range_d"], tr["n_samples_d"],
                             tr["perturb"], u_z)
    _close(f["z"], z_ref, 1e-6, 2e-6, "z_vals")
    u6 = torch.from_numpy(DO.uniform_draws(direct.last_seed_u, 1, 6))
    lat = slam.smoothness_points_torch(u6.to(dev), tr["smooth_pts"], tr["smooth_vox"], tr["smooth_margin"]).cpu()   # slam.py:198-207
    _close(f["pts"], lat.reshape(-1, 3).float(), 0, 2e-7, "TV lattice")
    for t in (fp.hash_table, fp.W1, fp.W2, fp.W3, fp.W4):
        t.requires_grad_(True)
    rend, ls, total = _oracle_iteration(fp, cfg, bbox, f["o"], f["d"], f["z"], f["tgt"], f["td"], False, f["pts"])
    _close(f["raw"].view(n, S, 4), rend["raw"], 1e-4, 2e-5, "raw")
    _close(f["rgb_map"], rend["rgb_res_map"], 1e-4, 2e-5, "rgb map")
    _close(f["depth_map"], rend["depth_res_map"], 1e-4, 2e-5, "depth map")
    for i, k in enumerate(("rgb_res_loss", "depth_res_loss", "sdf_res_loss", "fs_res_loss")):
        _close(lc[i], ls[k], 1e-4, 1e-8, k)
    total.backward()
    fq = _f64_params(fp)
    with _probe() as pr:
        _, _, total64 = _oracle_iteration(fq, cfg, bbox, f["o"], f["d"], f["z"], f["tgt"], f["td"], False, f["pts"])
        total64.backward()
        ties = pr.bounds()                                # per element of dW1 / dW3: the terms of the samples whose ReLU can tie
    print(f"{name}: possible ReLU ties among {n * S} samples x 32 units: layer 1 {pr.n_ties[0]}, layer 3 {pr.n_ties[1]}")
    _grad_close(got[0], fp.hash_table.grad, fq.hash_table.grad, "d_hash (map iteration)", _level_groups(fp.hash_meta))
    for g, a, q, nm in zip(got[1:], (fp.W1, fp.W2, fp.W3, fp.W4), (fq.W1, fq.W2, fq.W3, fq.W4), ("dW1", "dW2", "dW3", "dW4")):
        _grad_close(g, a.grad, q.grad, nm + " (map iteration)", tie_bound=ties.get(nm))
    assert float((got[0] != 0).float().mean()) > 0.001
    frac_zero = float((f["d_raw"] == 0).all(dim=1).float().mean())
    assert 0.15 < frac_zero < 0.7, frac_zero           # a real batch: a good share of the rows carry no gradient

    # ------------------------------------------------------------------ pose phase (clamp on, pose gradients only)
    for p_ in params:
        p_.grad = None
    idx = torch.arange(0, n_kf + 1, device=dev).contiguous()
    random.seed(4); torch.manual_seed(4)
    lc = direct.pose_gradients(cur, idx, map_grads=False)
    torch.cuda.synchronize()
    K = n_kf + 1
    R = direct._buffers(n, K, dev)
    f = {k: v.cpu() for k, v in _ws_fields(lib, R, n, S, P, nF, nL).items()}
    got_dp = R.t.dposes[:K].detach().cpu().clone()
    pose_in = R.t.poses[:K].detach().cpu().clone()
    pidx = f["pidx"].long()

    def pose_loss(fpp, dtype):
        pz = pose_in.clone().to(dtype).requires_grad_(True)
        o = pz[pidx, :3, 3]
        d = torch.sum(f["d_cam"].to(dtype)[:, None, :] * pz[pidx, :3, :3], -1)
        rend_, ls_, tot = _oracle_iteration(fpp, cfg, bbox, o, d, f["z"], f["tgt"], f["td"], True)
        tot.backward()
        return pz.grad, rend_, ls_

    fp2 = _oracle_params(cfg, model)
    g32, rend, ls = pose_loss(fp2, torch.float32)
    _close(f["raw"].view(n, S, 4), rend["raw"].detach(), 1e-4, 2e-5, "raw (pose phase)")
    for i, k in enumerate(("rgb_res_loss", "depth_res_loss", "sdf_res_loss", "fs_res_loss")):
        _close(lc[i], ls[k].detach(), 1e-4, 1e-8, k + " (pose phase)")
    fq2 = _f64_params(fp2)
    for k in ("hash_table", "W1", "W2", "W3", "W4"):
        setattr(fq2, k, getattr(fq2, k).detach())
    g64, _, _ = pose_loss(fq2, torch.float64)
    assert float(g64[:, :3, :].abs().max()) > 0
    # groups: rotation block and translation column of every camera (their scales differ by orders of magnitude)
    rot = torch.zeros((K, 4, 4), dtype=torch.bool); rot[:, :3, :3] = True
    tra = torch.zeros((K, 4, 4), dtype=torch.bool); tra[:, :3, 3] = True
    for mask, nm in ((rot, "d poses (rotation)"), (tra, "d poses (translation)")):
        _grad_close(got_dp[mask], g32[mask], g64[mask], nm, k=8.0)
    assert float(got_dp[:, 3, :].abs().max()) == 0.0


def test_map_phase_defines_every_entry_of_a_large_table_without_a_zero_fill():
    """Round 6, T = 2^21 (cafeteria sizes: 166 MB, nine hashed levels of 2^21 entries): in the map phase the one-call iteration
    skips the zero-fill of the trailing hashed levels and lets the scatter's reduce WRITE them (one block per segment, zeros
    included).  The gradient buffer is poisoned with NaN before the call: every entry must come out defined, and equal -- per
    level, to within what two runs of the float-atomic levels differ by -- to the stage-by-stage issue, which zero-fills the
    whole buffer and adds (same seeds: the same ray batch and lattice)."""
    cfg, pipe, fr = _pipeline("cafeteria", 11)
    mp, model, slam = pipe.mapper, pipe.model, pipe.slam
    direct = mp._direct_iterations()
    tr, m = cfg["training"], cfg["mapping"]
    enc = model.embed_res_fn
    assert int(cfg["grid"]["hash_size"]) == 21 and sum(int(enc.desc.hashed[l]) for l in range(16)) >= 8
    last = 10
    b = fr[last]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    n_kf = len(mp.keyframe.frame_ids)
    poses = slam.est_c2w_data[0:last + 1:m["keyframe_every"]].clone().float().contiguous()
    poses_all = torch.cat([poses, slam.est_c2w_data[last:last + 1].float()], 0)[:n_kf + 1].contiguous()
    dev = poses_all.device
    params = [enc.params] + list(model.decoder_res.fused_weights())
    n = direct._n_rays()

    def run(stagewise, seed):
        for p_ in params:
            p_.grad = None
        direct.stagewise_every = 1 if stagewise else 0
        direct._count = 0
        if not stagewise:
            direct._buffers(n, 0, dev).t.dt.fill_(float("nan"))          # the one-call form must define ALL of it
        random.seed(seed); torch.manual_seed(seed)
        lc = direct.map_gradients(cur, poses_all).clone()
        torch.cuda.synchronize()
        return enc.params.grad.detach().clone(), [w.grad.detach().clone() for w in params[1:]], lc

    one_a, dw_a, lc_a = run(False, 21)
    one_b, _, _ = run(False, 21)
    stg, dw_s, lc_s = run(True, 21)
    direct.stagewise_every = 0
    assert bool(torch.isfinite(one_a).all()) and bool(torch.isfinite(stg).all())
    assert torch.allclose(lc_a[:4], lc_s[:4], rtol=1e-5, atol=0)
    for a_, s_ in zip(dw_a, dw_s):
        assert torch.allclose(a_, s_, rtol=2e-3, atol=2e-5 * float(s_.pow(2).mean().sqrt()))
    touched = 0
    for l in range(16):
        lo, hi = int(enc.desc.offset[l]) * 2, (int(enc.desc.offset[l]) + int(enc.desc.size[l])) * 2
        ref, got, again = stg[lo:hi], one_a[lo:hi], one_b[lo:hi]
        scale = float(ref.abs().max())
        assert scale > 0, l
        noise = float((again - got).abs().max())                           # run-to-run difference of the same call
        err = float((got - ref).abs().max())
        assert err <= 4 * noise + 4e-6 * scale, (l, err, noise, scale)
        assert torch.equal(got == 0, ref == 0) or float(((got == 0) != (ref == 0)).float().mean()) < 1e-4, l      # the same entries untouched
        touched += int((got != 0).sum())
    assert touched > 1e6


@pytest.mark.parametrize("stashed", [False, True])
def test_backward_with_selection_matches_oracle_above_the_threshold(stashed):
    """n >= 16 384 with ~40 % of the d_raw rows exactly zero: the stable partition + stash + LDS sweep against the oracle
    directly (not against the same kernels on halves), per element."""
    from remixfusion_amd import _lib as L
    from test_field_gpu import _model, _points
    lib = L.load()
    cfg, m = _model(hash_scale=0.5)
    fp = _oracle_params(cfg, m)
    n = 20001
    x = _points(n, seed=8, lo=0.02, hi=0.98)
    g = torch.Generator().manual_seed(13)
    draw = torch.randn((n, 4), generator=g)
    draw[torch.rand(n, generator=g) < 0.4] = 0.0
    draw[5000:5300] = 0.0
    xo = x.clone().requires_grad_(True)
    for t in (fp.hash_table, fp.W1, fp.W2, fp.W3, fp.W4):
        t.requires_grad_(True)
    FO.query_color_sdf(fp, xo, True).backward(draw)
    fq = _f64_params(fp)
    xq = x.clone().requires_grad_(True)
    with _probe() as pr:
        FO.query_color_sdf(fq, xq, True).backward(draw.double())
        ties = pr.bounds()
    desc = m._field_desc(True)
    st = L.stream_ptr(torch.device("cuda"))
    xx, dd = x.cuda().contiguous(), draw.cuda().contiguous()
    nbytes = int(lib.rfx_field_backward_workspace_bytes(n))
    ws = torch.full((nbytes // 4 + 16,), float("nan"), device="cuda")
    wsp = (ws.data_ptr() + 15) // 16 * 16
    raw = torch.empty((n, 4), device="cuda")
    if stashed:
        L.check(lib.rfx_field_forward_stash(C.byref(desc), L.ptr(xx), n, L.ptr(raw), wsp, nbytes, st), "forward_stash")
    chain = lib.rfx_field_backward_chain_stashed if stashed else lib.rfx_field_backward_chain
    dws = [torch.zeros_like(w) for w in m.decoder_res.fused_weights()]
    d_hash, dx = torch.zeros_like(m.embed_res_fn.params), torch.full((n, 3), float("nan"), device="cuda")
    L.check(chain(C.byref(desc), L.ptr(xx), n, L.ptr(dd), wsp, nbytes, st), "chain")
    L.check(lib.rfx_field_backward_weights(n, L.ptr(dd), *[L.ptr(t) for t in dws], wsp, nbytes, st), "weights")
    L.check(lib.rfx_field_backward_scatter(C.byref(desc), L.ptr(xx), n, L.ptr(d_hash), L.ptr(dx), wsp, nbytes, st), "scatter")
    L.check(lib.rfx_field_backward_dx(C.byref(desc), L.ptr(xx), n, L.ptr(dd), L.ptr(dx), wsp, nbytes, st), "dx")
    torch.cuda.synchronize()
    for got, a, q, nm in zip(dws, (fp.W1, fp.W2, fp.W3, fp.W4), (fq.W1, fq.W2, fq.W3, fq.W4), ("dW1", "dW2", "dW3", "dW4")):
        _grad_close(got, a.grad, q.grad, nm, tie_bound=ties.get(nm))
    _grad_close(d_hash, fp.hash_table.grad, fq.hash_table.grad, "d_hash", _level_groups(fp.hash_meta))
    _grad_close(dx, xo.grad, xq.grad, "dx01", k=8.0)
    zero_rows = (draw == 0).all(dim=1)
    assert bool((dx.cpu()[zero_rows] == 0).all())


def test_full_frame_fused_render_matches_oracle_on_random_rays():
    """rfx_render_rays over a whole 640x480 frame (307 200 rays: the grid-stride loop and the XCD ray order are active),
    2 000 random rays of it against the oracle at rel 1e-4."""
    from test_field_gpu import _model
    from remixfusion_amd.datasets import get_dataset
    cfg, m = _model("office0")
    fp = _oracle_params(cfg, m)
    tr, cam = cfg["training"], cfg["cam"]
    S = tr["n_range_d"] + tr["n_samples_d"]
    ds = get_dataset(cfg, device="cuda", n_frames=4)
    b = ds[2]
    H, W = cam["H"], cam["W"]
    assert H * W == 307200
    c2w = b["c2w"].cuda()
    rays_d = torch.sum(b["direction"].reshape(-1, 3).cuda().unsqueeze(1) * c2w[None, :3, :3], -1).reshape(-1, 3).contiguous()
    rays_o = c2w[:3, -1].repeat(rays_d.shape[0], 1).contiguous()
    td = b["depth"].reshape(-1, 1).cuda().contiguous()
    m.train()
    torch.manual_seed(21)
    rgb, dep = m.render_fused(rays_o, rays_d, td)
    torch.manual_seed(21)
    u = torch.rand((H * W, S), device="cuda")            # the draw render_fused made
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(5))[:2000]
    sel[:4] = torch.tensor([0, W - 1, H * W - W, H * W - 1])
    o, d, t, uu = rays_o.cpu()[sel], rays_d.cpu()[sel], td.cpu()[sel], u.cpu()[sel]
    z = FO.sample_z_vals(t, cam["near"], cam["far"], tr["range_d"], tr["n_range_d"], tr["n_samples_d"], tr["perturb"], uu)
    ref = FO.render_rays(fp, m.bounding_box.cpu(), o, d, z, clamp=False, sc_factor=cfg["data"]["sc_factor"])
    _close(rgb.cpu()[sel], ref["rgb_res_map"], 1e-4, 2e-5, "full-frame fused rgb")
    _close(dep.cpu()[sel], ref["depth_res_map"], 1e-4, 2e-5, "full-frame fused depth")
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(dep).all())


def test_stage_events_time_the_one_call_iteration_without_changing_it():
    """ABI 10: rfx_ba_desc.stage_events.  The one-call iteration records the caller's events at its stage boundaries (what
    bench.py's roofline is made of): with and without them the same gradients (same seeds: the decoder gradients bit for bit, the
    table gradient to within the float-atomic flush's own run-to-run difference), the stages that ran have positive times that
    add up to no more than the whole call, and the stages of the OTHER phase record nothing."""
    from remixfusion_amd import _lib
    cfg, pipe, fr = _pipeline("office0", 11)
    mp, model, slam = pipe.mapper, pipe.model, pipe.slam
    direct = mp._direct_iterations()
    m = cfg["mapping"]
    enc = model.embed_res_fn
    last = 10
    b = fr[last]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    n_kf = len(mp.keyframe.frame_ids)
    poses = slam.est_c2w_data[0:last + 1:m["keyframe_every"]].clone().float().contiguous()
    poses_all = torch.cat([poses, slam.est_c2w_data[last:last + 1].float()], 0)[:n_kf + 1].contiguous()
    lib = _lib.load()
    params = [enc.params] + list(model.decoder_res.fused_weights())
    direct.stagewise_every = 0

    def new_set():
        arr = (C.c_void_p * _lib.BA_STAGE_EVENTS)()
        for i in range(_lib.BA_STAGE_EVENTS):
            ev = C.c_void_p()
            _lib.check(lib.rfx_event_create(C.byref(ev)), "rfx_event_create")
            arr[i] = ev.value
        return arr

    handed = []

    def provider(phase, n):
        arr = new_set()
        handed.append((phase, n, arr))
        return C.addressof(arr)

    def run_map(with_events, seed):
        for p_ in params:
            p_.grad = None
        direct.stage_events = provider if with_events else None
        random.seed(seed); torch.manual_seed(seed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lc = direct.map_gradients(cur, poses_all).clone()
        e1.record()
        torch.cuda.synchronize()
        return enc.params.grad.detach().clone(), [w.grad.detach().clone() for w in params[1:]], lc, e0.elapsed_time(e1)

    try:
        plain, dw_p, lc_p, _ = run_map(False, 5)
        again, _, _, _ = run_map(False, 5)
        timed, dw_t, lc_t, whole_ms = run_map(True, 5)
        assert torch.equal(lc_p, lc_t)
        for a_, t_ in zip(dw_p, dw_t):
            assert torch.equal(a_, t_)
        noise = float((again - plain).abs().max())
        assert float((timed - plain).abs().max()) <= 4 * noise + 4e-6 * float(plain.abs().max())
        assert len(handed) == 1 and handed[0][0] == "map" and handed[0][1] == direct._n_rays()
        arr = handed[0][2]
        ev = _lib.BA_EV
        order = ("start", "prologue", "forward", "loss", "chain", "weights", "scatter")
        total = 0.0
        for a_, b_ in zip(order[:-1], order[1:]):
            ms = C.c_float()
            assert lib.rfx_event_elapsed_ms(arr[ev[a_]], arr[ev[b_]], C.byref(ms)) == 0, (a_, b_)
            assert 0.0 < ms.value < 5.0, (b_, ms.value)
            total += ms.value
        assert total <= whole_ms * 1.05 + 0.02, (total, whole_ms)
        ms = C.c_float()
        for name in ("dx_table", "dx", "pose"):          # never recorded in the map phase: no elapsed time to be had
            assert lib.rfx_event_elapsed_ms(arr[ev["scatter"]], arr[ev[name]], C.byref(ms)) != 0, name
        # the pose phase records its own stages
        idx = torch.arange(0, poses_all.shape[0], device=poses_all.device)        # as Mapper.global_pose forms it
        direct.stage_events = provider
        random.seed(6); torch.manual_seed(6)
        direct.pose_gradients(cur, idx, map_grads=False)
        torch.cuda.synchronize()
        assert len(handed) == 2 and handed[1][0] == "pose"
        arr = handed[1][2]
        for a_, b_ in zip(("start", "prologue", "forward", "loss", "chain", "dx_table", "dx"), ("prologue", "forward", "loss", "chain", "dx_table", "dx", "pose")):
            assert lib.rfx_event_elapsed_ms(arr[ev[a_]], arr[ev[b_]], C.byref(ms)) == 0 and ms.value > 0.0, (a_, b_)
        assert lib.rfx_event_elapsed_ms(arr[ev["chain"]], arr[ev["weights"]], C.byref(ms)) != 0
    finally:
        direct.stage_events = None
        for _, _, arr in handed:
            for i in range(len(arr)):
                lib.rfx_event_destroy(arr[i])
