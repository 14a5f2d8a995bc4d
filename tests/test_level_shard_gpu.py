"""The hash table partitioned by LEVEL over several GPUs (include/rfx.h ABI 6: rfx_field_stash_put, rfx_field_forward_stashed,
rfx_field_backward_demb_rows, rfx_grid_encode_backward_merged, rfx_ba_shard_*), on ONE GPU and in ONE process: the ranks of a
world are played in turn, each with a workspace, gradient buffers and exchange buffers of its own, and the collectives between
the phases are tensor copies.  What a world of any size computes must be the single-GPU iteration
(rfx_ba_forward_backward, itself held to the oracle at this size by tests/test_timed_path_gpu.py; reference:
mp_slam/mapper.py:392-423 / :470-505) up to the order of floating-point sums.  tests/test_dist_gpu.py and
tests/test_sharded_configs_gpu.py run the same code over real process groups."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pipeline(name="office0", n_frames=21, small=True):
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config(name)
    if small:
        cfg["cam"].update({"H": 240, "W": 320, "fx": 288.0, "fy": 288.0, "cx": 159.5, "cy": 119.5})
        cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 10})
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.02})
    cfg["pipeline"] = {"mv_stream": False}
    pipe = MappingPipeline(cfg, n_frames=n_frames + 4, seed=7)
    frames = pipe.prefetch(list(range(n_frames)))
    pipe.start(frames[0])
    for i in range(1, n_frames):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    return cfg, pipe, frames


def _sub_desc(desc, l0, l1):
    from remixfusion_amd import _lib as L
    g = L.GridDesc()
    g.n_levels, g.n_feat = l1 - l0, desc.n_feat
    for i in range(l1 - l0):
        for f in ("scale", "res", "size", "offset", "hashed"):
            getattr(g, f)[i] = getattr(desc, f)[l0 + i]
    return g


def test_features_looked_up_level_by_level_give_the_fused_forward_bit_for_bit():
    """rfx_grid_encode_forward on sub-grids of the table -> blocks [points, 2k] -> rfx_field_stash_put ->
    rfx_field_forward_stashed == rfx_field_forward; and the stashed chain + rfx_field_backward_demb_rows + merged scatter of the
    sub-grids == rfx_field_backward's hash gradient (float atomics: to rounding)."""
    from remixfusion_amd import _lib as L
    from remixfusion_amd.dist import level_partition
    lib = L.load()
    cfg, pipe, _ = _pipeline()
    model = pipe.model
    enc = model.embed_res_fn
    dev = enc.params.device
    st = L.stream_ptr(dev)
    g = torch.Generator(device="cuda").manual_seed(1)
    for n in (1000, 20011):                       # below / above the row selection's threshold (16 384)
        x = torch.rand((n, 3), device=dev, generator=g) * 0.9 + 0.05
        desc = model._field_desc(False)
        raw_ref = torch.empty((n, 4), device=dev)
        L.check(lib.rfx_field_forward(C.byref(desc), L.ptr(x), n, L.ptr(raw_ref), st), "forward")
        draw = torch.randn((n, 4), device=dev, generator=g)
        draw[torch.rand(n, device=dev, generator=g) < 0.4] = 0.0            # rows the selection drops
        wsb = int(lib.rfx_field_backward_workspace_bytes(n))
        ws = torch.empty(wsb // 4 + 16, device=dev)
        dt_ref = torch.zeros_like(enc.params)
        dws_ref = [torch.zeros_like(w) for w in model.decoder_res.fused_weights()]
        L.check(lib.rfx_field_backward(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(dt_ref), *[L.ptr(w) for w in dws_ref], None,
                                       L.ptr(ws), wsb, st), "backward")
        for world in (1, 3, 16):
            cuts = level_partition(enc.desc, world)
            blocks = []
            for q in range(world):
                sub = _sub_desc(enc.desc, cuts[q], cuts[q + 1])
                f = torch.empty((n, 2 * (cuts[q + 1] - cuts[q])), device=dev)
                L.check(lib.rfx_grid_encode_forward(C.byref(sub), L.ptr(enc.params), L.ptr(x), n, L.ptr(f), st), "encode")
                blocks.append(f)
            rows = L.LevelRows()
            for q in range(world):
                for l in range(cuts[q], cuts[q + 1]):
                    rows.rows[l], rows.ld[l], rows.col[l] = blocks[q].data_ptr(), blocks[q].shape[1], 2 * (l - cuts[q])
            ws2 = torch.full_like(ws, float("nan"))
            L.check(lib.rfx_field_stash_put(C.byref(rows), n, L.ptr(ws2), wsb, st), "stash_put")
            raw = torch.empty((n, 4), device=dev)
            L.check(lib.rfx_field_forward_stashed(C.byref(desc), L.ptr(x), n, L.ptr(raw), L.ptr(ws2), wsb, st), "forward_stashed")
            assert torch.equal(raw, raw_ref), (n, world)
            # backward: chain on the stash, gradient rows out, scattered sub-grid by sub-grid
            L.check(lib.rfx_field_backward_chain_weights_stashed(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(ws2), wsb, st), "chain")
            dws = [torch.zeros_like(w) for w in dws_ref]
            L.check(lib.rfx_field_backward_weights(n, L.ptr(draw), *[L.ptr(w) for w in dws], L.ptr(ws2), wsb, st), "weights")
            outs = [torch.full_like(b, float("nan")) for b in blocks]
            orow = L.LevelRows()
            for q in range(world):
                for l in range(cuts[q], cuts[q + 1]):
                    orow.rows[l], orow.ld[l], orow.col[l] = outs[q].data_ptr(), outs[q].shape[1], 2 * (l - cuts[q])
            L.check(lib.rfx_field_backward_demb_rows(n, C.byref(orow), None, 0, None, L.ptr(ws2), wsb, st), "demb_rows")
            dt = torch.zeros_like(enc.params)
            for q in range(world):
                assert bool(torch.isfinite(outs[q]).all())
                zero_rows = (draw == 0).all(1)
                assert float(outs[q][zero_rows].abs().max()) == 0.0          # no gradient: exact zeros
                sub = _sub_desc(enc.desc, cuts[q], cuts[q + 1])
                nb = int(lib.rfx_grid_encode_backward_workspace_bytes(n, sub.n_levels))
                sws = torch.empty(nb // 4, device=dev)
                half = n // 2                                              # as two point sets through the merged entry point
                L.check(lib.rfx_grid_encode_backward_merged(C.byref(sub), L.ptr(enc.params), L.ptr(x[:half]), half, L.ptr(outs[q][:half]),
                                                            L.ptr(x[half:]), n - half, L.ptr(outs[q][half:]), L.ptr(dt), L.ptr(sws), nb, st),
                        "merged")
            torch.cuda.synchronize()
            for a, b in zip(dws, dws_ref):
                assert torch.equal(a, b)                                     # same points, same order: the deterministic sums
            for l in range(16):
                lo, hi = int(enc.desc.offset[l]) * 2, (int(enc.desc.offset[l]) + int(enc.desc.size[l])) * 2
                ref, got = dt_ref[lo:hi], dt[lo:hi]
                scale = float(ref.abs().max())
                assert scale > 0
                # float atomics: the same terms added in another order
                assert float((got - ref).abs().max()) <= 2e-5 * scale + 1e-12, (n, world, l)
                assert float(((got - ref).abs() / (ref.abs() + 1e-3 * scale)).max()) < 2e-2, (n, world, l)


def _alloc_rank(lib, L, direct, B, d, q, world, cuts, n, S, K, dev, map_grads, pose):
    from remixfusion_amd.dist import ray_partition
    enc = direct.model.embed_res_fn
    r = {}
    r["ws"] = torch.empty(B.ws_bytes // 4 + 64, device=dev)
    r["wsp"] = (r["ws"].data_ptr() + 255) // 256 * 256
    r["dt"] = torch.full_like(enc.params, float("nan"))
    r["dw"] = torch.full_like(B.t.dw_flat, float("nan"))
    r["dposes"] = torch.full((K, 4, 4), float("nan"), device=dev)
    r["lc"] = torch.zeros(8, device=dev)
    rs = ray_partition(n, world)
    m = rs[q + 1] - rs[q]
    k = cuts[q + 1] - cuts[q]
    r["m"], r["k"], r["rs"] = m, k, rs
    f32 = dict(dtype=torch.float32, device=dev)
    r["feat_send"] = torch.full((n * S, 2 * k), float("nan"), **f32)
    r["feat_recv"] = torch.full((max(m, 1) * S * 32,), float("nan"), **f32)
    r["demb_send"] = torch.full((max(m, 1) * S * 32,), float("nan"), **f32)
    r["demb_recv"] = torch.full((n * S, 2 * k), float("nan"), **f32)
    r["dx_send"] = torch.full((n * S, 3), float("nan"), **f32)
    r["dx_recv"] = torch.full((world, max(m, 1) * S, 3), float("nan"), **f32)
    r["sums8"] = torch.full((8,), float("nan"), dtype=torch.float64, device=dev)
    sh = L.BaShard()
    sh.rank, sh.world = q, world
    for i in range(world + 1):
        sh.level_start[i], sh.ray_start[i] = cuts[i], rs[i]
    for nm in ("feat_send", "feat_recv", "demb_send", "demb_recv", "dx_send", "dx_recv"):
        setattr(sh, nm, r[nm].data_ptr())
    sh.loss_sums8 = r["sums8"].data_ptr()
    r["shard"] = sh
    dq = type(d).from_buffer_copy(d)
    dq.d_hash, dq.d_w = (r["dt"].data_ptr(), r["dw"].data_ptr()) if map_grads else (None, None)
    dq.d_poses16 = r["dposes"].data_ptr() if pose else None
    dq.losses8 = r["lc"].data_ptr()
    dq.rba = dq.rba_acts = dq.rba_grads = dq.rba_ws = None
    r["desc"] = dq
    return r


def _play_world(lib, L, direct, B, d, world, n, S, K, dev, map_grads, pose, st):
    """the ranks of a world in turn; returns (d_hash assembled from the own slices, dW summed, dposes summed, losses)"""
    from remixfusion_amd.dist import level_partition
    enc = direct.model.embed_res_fn
    cuts = level_partition(enc.desc, world)
    R = [_alloc_rank(lib, L, direct, B, d, q, world, cuts, n, S, K, dev, map_grads, pose) for q in range(world)]
    call = lambda fn, r: L.check(fn(C.byref(r["desc"]), C.byref(r["shard"]), r["wsp"], B.ws_bytes, st), fn.__name__)
    for r in R:
        call(lib.rfx_ba_shard_lookup, r)
    for q, r in enumerate(R):               # all-to-all: the own rays' rows of every rank's features
        a, b = r["rs"][q] * S, r["rs"][q + 1] * S
        r["feat_recv"][:r["m"] * S * 32].copy_(torch.cat([o["feat_send"][a:b].reshape(-1) for o in R]))
    for r in R:
        call(lib.rfx_ba_shard_render, r)
    for q, r in enumerate(R):               # all-to-all back: block q of every rank's gradient rows, in rank (= ray) order
        parts = []
        for o in R:
            off = o["m"] * S * 2 * cuts[q]
            parts.append(o["demb_send"][off:off + o["m"] * S * 2 * r["k"]].view(o["m"] * S, 2 * r["k"]))
        r["demb_recv"].copy_(torch.cat(parts, 0))
    sums = torch.stack([r["sums8"] for r in R]).sum(0)
    assert bool(torch.isfinite(sums).all())
    for r in R:
        call(lib.rfx_ba_shard_scatter, r)
    dposes = None
    if pose:
        for q, r in enumerate(R):
            a, b = r["rs"][q] * S, r["rs"][q + 1] * S
            for j, o in enumerate(R):
                r["dx_recv"][j, :r["m"] * S].copy_(o["dx_send"][a:b])
        for r in R:
            L.check(lib.rfx_ba_shard_pose(C.byref(r["desc"]), C.byref(r["shard"]), r["wsp"], B.ws_bytes, st), "pose")
        dposes = torch.stack([r["dposes"] for r in R]).sum(0)
    lc = torch.zeros(8, device=dev)
    L.check(lib.rfx_mapping_loss_finalize(sums.data_ptr(), n, S, lc.data_ptr(), lc.data_ptr() + 16, st), "finalize")
    dt = dw = None
    if map_grads:
        dt = torch.full_like(enc.params, float("nan"))
        for q, r in enumerate(R):
            lo = int(enc.desc.offset[cuts[q]]) * 2
            hi = (int(enc.desc.offset[cuts[q + 1] - 1]) + int(enc.desc.size[cuts[q + 1] - 1])) * 2
            dt[lo:hi] = r["dt"][lo:hi]
            outside = torch.cat([r["dt"][:lo], r["dt"][hi:]])
            assert bool(torch.isnan(outside).all())                # a rank writes the own levels' part of the gradient only
        dw = torch.stack([r["dw"] for r in R]).sum(0)
    torch.cuda.synchronize()
    return dt, dw, dposes, lc


@pytest.mark.parametrize("name,frames,small", [("office0", 21, True), ("scene0000", 11, False)])
def test_a_world_of_any_size_computes_the_single_gpu_iteration(name, frames, small):
    """map phase (hash + decoder gradients, TV term) and pose phase (pose gradients, with and without map gradients) through
    rfx_ba_shard_lookup/_render/_scatter/_pose for worlds of 1, 2, 3 (uneven shares), 5 and 16 ranks against ONE
    rfx_ba_forward_backward call with the same seeds.  scene0000: T = 2^19 (binned levels, sub-grids of <= 8 levels)."""
    from remixfusion_amd import _lib as L
    lib = L.load()
    cfg, pipe, fr = _pipeline(name, frames, small)
    mp, model, slam = pipe.mapper, pipe.model, pipe.slam
    direct = mp._direct_iterations()
    m, tr = cfg["mapping"], cfg["training"]
    S = int(tr["n_range_d"]) + int(tr["n_samples_d"])
    last = frames - 1
    b = fr[last]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    n = direct._n_rays()
    dev = cur.device
    st = L.stream_ptr(dev)
    n_kf = len(mp.keyframe.frame_ids)
    poses = slam.est_c2w_data[0:last + 1:m["keyframe_every"]].clone().float().contiguous()
    poses_all = torch.cat([poses, slam.est_c2w_data[last:last + 1].float()], 0)[:n_kf + 1].contiguous()
    K = poses_all.shape[0]
    enc = model.embed_res_fn
    for phase, clamp, map_grads, pose in (("map", False, True, False), ("pose", True, False, True), ("pose+map", True, True, True)):
        B = direct._buffers(n, K, dev)
        random.seed(11)
        d = direct._fill(B, cur, poses_all.data_ptr(), K, clamp, B.p.dposes if pose else None, map_grads, None)
        d = type(d).from_buffer_copy(d)              # a snapshot: the seeds of THIS iteration
        # ---- the single-GPU iteration, twice (the float atomics' own run-to-run noise is the yardstick for the hash gradient)
        refs = []
        for _ in range(2):
            dt = torch.full_like(enc.params, float("nan"))
            dw = torch.full_like(B.t.dw_flat, float("nan"))
            dp = torch.full((K, 4, 4), float("nan"), device=dev)
            lc = torch.zeros(8, device=dev)
            d1 = type(d).from_buffer_copy(d)
            d1.d_hash, d1.d_w = (dt.data_ptr(), dw.data_ptr()) if map_grads else (None, None)
            d1.d_poses16 = dp.data_ptr() if pose else None
            d1.losses8 = lc.data_ptr()
            L.check(lib.rfx_ba_forward_backward(C.byref(d1), B.p.ws, B.ws_bytes, st), "single")
            torch.cuda.synchronize()
            refs.append((dt, dw, dp, lc))
        dt0, dw0, dp0, lc0 = refs[0]
        for world in (1, 2, 3, 5, 16):
            dt, dw, dp, lc = _play_world(lib, L, direct, B, d, world, n, S, K, dev, map_grads, pose, st)
            tag = (name, phase, world)
            assert torch.allclose(lc[:4], lc0[:4], rtol=2e-6, atol=0), (tag, lc, lc0)
            assert torch.allclose(lc[4:], lc0[4:], rtol=1e-6, atol=0), (tag, lc, lc0)
            if map_grads:
                assert bool(torch.isfinite(dt).all()) and bool(torch.isfinite(dw).all())
                # decoder gradients: the same per-point terms, grouped by rank; an element's error is a rounding of the terms'
                # magnitude, bounded here by the gradient's root mean square (elements near zero cancel)
                rms = float(dw0.pow(2).mean().sqrt())
                assert torch.allclose(dw, dw0, rtol=2e-3, atol=2e-5 * rms), tag
                for l in range(16):
                    lo, hi = int(enc.desc.offset[l]) * 2, (int(enc.desc.offset[l]) + int(enc.desc.size[l])) * 2
                    noise = float((refs[1][0][lo:hi] - dt0[lo:hi]).abs().max())      # single-GPU run against itself
                    err = float((dt[lo:hi] - dt0[lo:hi]).abs().max())
                    scale = float(dt0[lo:hi].abs().max())
                    assert err <= 4 * noise + 4e-6 * scale, (tag, l, err, noise, scale)
            if pose:
                assert bool(torch.isfinite(dp).all())
                scale = float(dp0.abs().max())
                assert float((dp - dp0).abs().max()) <= 2e-4 * scale, (tag, float((dp - dp0).abs().max()), scale)
                assert torch.allclose(dp[:, :3], dp0[:, :3], rtol=5e-3, atol=2e-4 * scale), tag


def test_lookup_in_two_launches_is_the_one_launch_bit_for_bit():
    """ABI 9: rfx_ba_shard_lookup_rays followed by rfx_ba_shard_lookup_tv (the caller starts the feature all-to-all between the
    two) leaves exactly what rfx_ba_shard_lookup leaves: the rows to send, the whole workspace (ray batch, points, lattice and
    its features), the zero-filled own range of the gradient -- and nothing outside that range."""
    from remixfusion_amd import _lib as L
    from remixfusion_amd.dist import level_partition
    lib = L.load()
    cfg, pipe, fr = _pipeline("office0", 21, True)
    mp, model, slam = pipe.mapper, pipe.model, pipe.slam
    direct = mp._direct_iterations()
    m, tr = cfg["mapping"], cfg["training"]
    S = int(tr["n_range_d"]) + int(tr["n_samples_d"])
    last = 20
    b = fr[last]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    n = direct._n_rays()
    dev = cur.device
    st = L.stream_ptr(dev)
    n_kf = len(mp.keyframe.frame_ids)
    poses = slam.est_c2w_data[0:last + 1:m["keyframe_every"]].clone().float().contiguous()
    poses_all = torch.cat([poses, slam.est_c2w_data[last:last + 1].float()], 0)[:n_kf + 1].contiguous()
    K = poses_all.shape[0]
    enc = model.embed_res_fn
    for map_grads, pose in ((True, False), (False, True)):
        B = direct._buffers(n, K, dev)
        random.seed(5)
        d = direct._fill(B, cur, poses_all.data_ptr(), K, pose, B.p.dposes if pose else None, map_grads, None)
        d = type(d).from_buffer_copy(d)
        world, q = 3, 1
        cuts = level_partition(enc.desc, world)
        one, two = (_alloc_rank(lib, L, direct, B, d, q, world, cuts, n, S, K, dev, map_grads, pose) for _ in range(2))
        for r in (one, two):
            r["ws"].zero_()
        call = lambda fn, r: L.check(fn(C.byref(r["desc"]), C.byref(r["shard"]), r["wsp"], B.ws_bytes, st), fn.__name__)
        call(lib.rfx_ba_shard_lookup, one)
        call(lib.rfx_ba_shard_lookup_rays, two)
        feat_after_rays = two["feat_send"].clone()           # complete before the lattice launch: what the all-to-all sends
        call(lib.rfx_ba_shard_lookup_tv, two)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(one["feat_send"]).all())
        assert torch.equal(one["feat_send"], two["feat_send"]) and torch.equal(feat_after_rays, one["feat_send"])
        assert torch.equal(one["ws"], two["ws"])
        if map_grads:
            lo = int(enc.desc.offset[cuts[q]]) * 2
            hi = (int(enc.desc.offset[cuts[q + 1] - 1]) + int(enc.desc.size[cuts[q + 1] - 1])) * 2
            for r in (one, two):
                assert float(r["dt"][lo:hi].abs().max()) == 0.0
                assert bool(torch.isnan(torch.cat([r["dt"][:lo], r["dt"][hi:]])).all())


def test_sliced_adam_steps_only_the_own_levels():
    """optim.Adam.slices: elements outside [lo, hi) keep parameter and state; inside they take torch.optim.Adam's step"""
    from remixfusion_amd.optim import Adam
    g = torch.Generator(device="cuda").manual_seed(0)
    p = torch.nn.Parameter(torch.randn(10000, device="cuda", generator=g))
    q = torch.nn.Parameter(p.detach().clone())
    a, b = Adam([p], lr=1e-2, betas=(0.9, 0.99), eps=1e-15), torch.optim.Adam([q], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    a.slices[p] = (1000, 4000)
    p0 = p.detach().clone()
    for _ in range(3):
        grad = torch.randn(10000, device="cuda", generator=g)
        p.grad, q.grad = grad.clone(), grad.clone()
        a.step(); b.step()
    assert torch.equal(p[:1000], p0[:1000]) and torch.equal(p[4000:], p0[4000:])
    assert torch.allclose(p[1000:4000], q[1000:4000], rtol=1e-5, atol=1e-7)
    assert float(a.state[p]["exp_avg"][:1000].abs().max()) == 0.0 and float(a.state[p]["exp_avg"][1000:4000].abs().min()) > 0.0
