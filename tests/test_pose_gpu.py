"""GPU tests of the fused RBA pose MLP: librfx kernels vs oracle/rba_oracle.py (the reference's
model/rba.py:71-100 with kornia 0.6.12's published angle-axis formulas, written independently of the product) and vs the
same map as ATen ops (`RBA.forward_torch`); the graph-free BA iterations vs autograd through the oracles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rba(num_cams=40, seed=0, scale=1e-2):
    from remixfusion_amd.model.rba import RBA, make_c2w
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    aa = torch.randn((num_cams, 3), generator=g) * 0.8
    aa[3] = 0.0                                           # identity rotation: first-order branch
    aa[4] = torch.tensor([1e-4, -2e-4, 3e-4])            # theta^2 < eps
    t = torch.randn((num_cams, 3), generator=g) * 2.0
    init = make_c2w(aa, t)
    m = RBA(num_cams, init_c2w=init, scale=scale, device="cuda").cuda()
    return m


@pytest.mark.parametrize("scale", [1e-2, 1.0])
def test_fused_rba_matches_torch_ops(scale):
    m = _rba(scale=scale)
    ids = torch.tensor([0, 1, 2, 3, 4, 7, 7, 39, 20, 5], device="cuda").unsqueeze(-1)
    ref = m.forward_torch(ids)
    got = m(ids)
    assert got.shape == ref.shape == (10, 4, 4)
    assert float((got - ref).detach().abs().max()) < 2e-6
    assert float((got[0] - m.init_c2w[0]).detach().abs().max()) < 1e-5      # camera 0 is pinned to its initial pose
    g = torch.Generator().manual_seed(9)
    dp = torch.randn((10, 4, 4), generator=g).cuda()
    params = list(m.parameters())
    gr = torch.autograd.grad(ref, params, dp, allow_unused=True)
    gg = torch.autograd.grad(got, params, dp, allow_unused=True)
    for p, a, b in zip(params, gg, gr):
        assert a is not None and a.shape == p.shape
        tol = 2e-4 * float(b.abs().max()) + 1e-9
        assert float((a - b).abs().max()) <= tol, (tuple(p.shape), float((a - b).abs().max()), float(b.abs().max()))
    assert float(gr[2].abs().max()) > 0
    # ---- against the independent oracle: forward 2e-6, gradients per element (rel 1e-4 + 4x the oracle's fp32 noise)
    from oracle import rba_oracle as RO
    from test_field_gpu import _grad_close
    lin = m._linears()
    idc = ids.reshape(-1).cpu()

    def oracle(dtype):
        prm = [t.detach().cpu().to(dtype).requires_grad_(True) for l in lin for t in (l.weight, l.bias)]
        c2w = RO.rba_forward(prm, m.init_r.cpu().to(dtype), m.init_t.cpu().to(dtype), idc, m.num_cams, scale)
        return c2w, torch.autograd.grad(c2w, prm, dp.cpu().to(dtype))

    o32, g32 = oracle(torch.float32)
    o64, g64 = oracle(torch.float64)
    assert float((got.detach().cpu() - o32).abs().max()) < 2e-6
    got_by_lin = [g for l in lin for g in (dict(zip(params, gg))[l.weight], dict(zip(params, gg))[l.bias])]
    for a, r32, r64 in zip(got_by_lin, g32, g64):
        _grad_close(a, r32, r64, f"RBA grad {tuple(a.shape)}", k=8.0)


def test_frame_pose_kernel_matches_torch_inverse():
    """rfx_frame_pose: est <- c2w, rel <- c2w @ inverse(kf) (the tracker-side bookkeeping of one frame)."""
    from remixfusion_amd import _lib as L
    from remixfusion_amd.model.rba import angle_axis_to_rotation_matrix
    lib = L.load()
    g = torch.Generator().manual_seed(2)
    st = L.stream_ptr(torch.device("cuda"))
    for trial in range(6):
        def pose():
            P = torch.eye(4)
            P[:3, :3] = angle_axis_to_rotation_matrix(torch.randn((1, 3), generator=g) * 0.7)[0]
            P[:3, 3] = torch.randn(3, generator=g) * 2.0
            if trial == 5:
                P[:3, :3] = P[:3, :3] * 1.01 + 0.001 * torch.randn((3, 3), generator=g)     # not exactly rigid
            return P
        c2w, kf = pose().cuda(), pose().cuda()
        est, rel = torch.zeros((4, 4), device="cuda"), torch.zeros((4, 4), device="cuda")
        assert lib.rfx_frame_pose(L.ptr(c2w), L.ptr(kf), L.ptr(est), L.ptr(rel), st) == 0
        ref = (c2w.double() @ torch.linalg.inv(kf.double())).float()
        assert torch.equal(est, c2w)
        assert float((rel - ref).abs().max()) < 2e-5, trial
        est2 = torch.zeros((4, 4), device="cuda")
        assert lib.rfx_frame_pose(L.ptr(c2w), None, L.ptr(est2), None, st) == 0 and torch.equal(est2, c2w)
    assert lib.rfx_frame_pose(None, None, L.ptr(est), None, st) == -1
    assert lib.rfx_frame_pose(L.ptr(c2w), None, L.ptr(est), L.ptr(rel), st) == -1


def test_set_init_pose_kernel_matches_the_tensor_formulation():
    """RBA.update_init_pose on the device is one librfx launch; same init_c2w / init_t / init_r as the tensor-op
    formulation (model/rba.py::rotation_matrix_to_angle_axis), incl. tiny rotations and rotations next to pi."""
    from remixfusion_amd.model.rba import RBA, angle_axis_to_rotation_matrix, rotation_matrix_to_angle_axis
    g = torch.Generator().manual_seed(4)
    aa = torch.randn((12, 3), generator=g)
    aa = aa / aa.norm(dim=-1, keepdim=True)
    ang = torch.tensor([0.0, 1e-7, 1e-5, 3e-4, 0.1, 1.0, 2.0, 3.0, 3.14159, 3.1415926, 3.141, 2.5])
    R = angle_axis_to_rotation_matrix(aa * ang[:, None])
    rba = RBA(12, device="cuda")
    for i in range(12):
        c2w = torch.eye(4)
        c2w[:3, :3] = R[i]
        c2w[:3, 3] = torch.tensor([0.1 * i, -0.2, 0.3])
        rba.update_init_pose(i, c2w.cuda())
        ref = rotation_matrix_to_angle_axis(R[i:i + 1]).reshape(-1)
        got = rba.init_r[i].cpu()
        # the angle-axis of a rotation next to pi is defined up to the sign of the axis: compare the rotations
        Rg = angle_axis_to_rotation_matrix(got[None])[0]
        assert float((Rg - R[i]).abs().max()) < 2e-3 if ang[i] > 3.14 else float((got - ref).abs().max()) < 1e-5, (i, got, ref)
        assert torch.equal(rba.init_c2w[i].cpu(), c2w) and torch.equal(rba.init_t[i].cpu(), c2w[:3, 3])


def test_fused_rba_in_the_pose_loop():
    """a few Adam steps on a pose-only objective move the poses the same way through both paths."""
    import copy
    m1 = _rba(num_cams=12, seed=3)
    m2 = copy.deepcopy(m1)
    ids = torch.arange(12, device="cuda").unsqueeze(-1)
    target = m1.forward_torch(ids).detach().clone()
    target[:, :3, 3] += 0.05
    for m, fn in ((m1, m1.forward), (m2, m2.forward_torch)):
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        for _ in range(20):
            opt.zero_grad()
            loss = ((fn(ids) - target) ** 2).sum()
            loss.backward()
            opt.step()
    a, b = m1(ids), m2.forward_torch(ids)
    assert float((a - b).detach().abs().max()) < 1e-4
    assert float((a[1:, :3, 3] - target[1:, :3, 3]).detach().abs().mean()) < 0.05 - 1e-3      # moved towards the target


def test_fused_ray_batch_equals_sample_plus_world_rays():
    """rfx_gather_rays / rfx_pose_grad vs the unfused host glue (sample_global_rays + random_subset + cat +
    poses[ids] + einsum), same python `random` state -> same rays, same pose gradient."""
    import random
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 5, "sample": 512, "iters": 1, "BA_iters": 1})
    pipe = MappingPipeline(cfg, n_frames=30, seed=1)
    frames = pipe.prefetch(list(range(17)))
    pipe.start(frames[0])
    for i in range(1, 17):
        pipe.step(i, frames[i])
    mp = pipe.mapper
    assert len(mp.keyframe) == 4 and mp.keyframe.device_sampling
    b = frames[16]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    K = 5
    g = torch.Generator().manual_seed(2)
    poses = torch.eye(4).repeat(K, 1, 1)
    poses[:, :3, :] += 0.3 * torch.randn((K, 3, 4), generator=g)
    p1 = poses.cuda().requires_grad_(True)
    p2 = poses.cuda().requires_grad_(True)
    random.seed(77)
    o1, d1, s1, t1 = mp._ray_batch(cur, p1)
    random.seed(77)
    rays, ids_all = mp._sample_rays(cur)
    o2, d2, s2, t2 = mp._world_rays(rays, ids_all, p2)
    n = 512 + max(512 // 4, cfg["mapping"]["min_pixels_cur"])
    assert o1.shape == (n, 3) and t1.shape == (n, 1)
    assert torch.equal(o1, o2) and torch.equal(s1, s2) and torch.equal(t1, t2)
    assert float((d1 - d2).detach().abs().max()) < 1e-6
    go, gd = torch.randn((n, 3), generator=g).cuda(), torch.randn((n, 3), generator=g).cuda()
    (g1,) = torch.autograd.grad([o1, d1], [p1], [go, gd])
    (g2,) = torch.autograd.grad([o2, d2], [p2], [go, gd])
    assert float((g1 - g2).abs().max()) < 1e-4 * float(g2.abs().max())
    assert float(g1[:, 3].abs().max()) == 0.0


def test_checkpoint_round_trip(tmp_path):
    """Mapper.save_ckpt / SLAM.load_ckpt (reference mapper.py:257-265, slam.py:128-135): same keys, same state."""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 5, "sample": 512, "iters": 1, "BA_iters": 1})
    pipe = MappingPipeline(cfg, n_frames=20, seed=1)
    frames = pipe.prefetch(list(range(7)))
    pipe.start(frames[0])
    for i in range(1, 7):
        pipe.step(i, frames[i])
    path = str(tmp_path / "checkpoint.pt")
    pipe.mapper.save_ckpt(path)
    keys = set(pipe.model.state_dict().keys())
    assert {"embed_res_fn.params", "GBV.params", "GBW.params"} <= keys
    assert any(k.startswith("decoder_res.") for k in keys) and any(k.startswith("rba.layers.") for k in keys)
    other = MappingPipeline(cfg, n_frames=20, seed=5)
    other.slam.load_ckpt(path)
    for k, v in pipe.model.state_dict().items():
        assert torch.equal(v, other.model.state_dict()[k]), k
    assert torch.equal(other.slam.est_c2w_data.cpu(), pipe.slam.est_c2w_data.cpu())
    b = frames[5]
    pipe.model.train(); other.model.train()
    with torch.no_grad():
        torch.manual_seed(0)                      # the sampler jitters the depths (training.perturb)
        r1 = pipe.slam.render_single(5, b["depth"][None], b["rgb"][None], b["c2w"], b["direction"], gap=4)
        torch.manual_seed(0)
        r2 = other.slam.render_single(5, b["depth"][None], b["rgb"][None], b["c2w"], b["direction"], gap=4)
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])


def test_direct_iterations_equal_autograd_iterations():
    """mp_slam/direct.py (kernels issued without an autograd graph) vs the autograd formulation of one
    global_mapping / global_pose iteration: same state and random draws -> same gradients.  (Parameters after an
    Adam step are not compared: with eps = 1e-15 the update is +-lr for any non-zero gradient, so round-off in
    near-zero entries flips whole steps.)"""
    import random
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    from remixfusion_amd.mp_slam.direct import DirectIterations
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 5, "sample": 512, "iters": 2, "BA_iters": 2})
    cfg["training"].update({"smooth_pts": 16})
    pipe = MappingPipeline(cfg, n_frames=30, seed=1)
    frames = pipe.prefetch(list(range(12)))
    pipe.start(frames[0])
    for i in range(1, 11):
        pipe.step(i, frames[i])
    mp, model, slam = pipe.mapper, pipe.model, pipe.slam
    assert mp._direct_iterations() is not None            # the pipeline above ran on the direct path
    direct = DirectIterations(mp)
    direct.torch_draws = True                             # both formulations take their uniforms from torch's generator
    b = frames[10]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    K = len(mp.keyframe) + 1
    all_index = torch.arange(0, K, device="cuda").unsqueeze(-1)
    params = [model.embed_res_fn.params] + list(model.decoder_res.fused_weights())
    rba_params = list(model.rba.parameters())

    def grads_of(ps):
        return [p.grad.detach().clone() for p in ps]

    def reset():
        for p in params + rba_params:
            p.grad = None
        random.seed(11); torch.manual_seed(11)

    # ---- map phase
    poses = slam.est_c2w_data[0:11:5].clone().cuda().float()
    poses = torch.cat([poses, poses[-1:]], 0)[:K]
    reset()
    rays_o, rays_d, ts, td = mp._ray_batch(cur, poses)
    slam.get_loss_from_ret(model.mapping(rays_o, rays_d, ts, td), smooth=True).backward()
    ref = grads_of(params)
    reset()
    direct.map_gradients(cur, poses)
    got = grads_of(params)
    # both formulations against autograd through the oracle, fed the ray batch the call drew (same seeds: the two
    # formulations draw the same batch), per element
    from oracle import field_oracle as FO
    from oracle import rba_oracle as RO
    from test_field_gpu import _f64_params, _grad_close, _level_groups, _oracle_params, _probe
    from test_timed_path_gpu import _oracle_iteration, _ws_fields
    from remixfusion_amd import _lib as L
    lib = L.load()
    tr = cfg["training"]
    S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
    enc = model.embed_res_fn
    n = direct._n_rays()
    bbox = model.bounding_box.cpu()

    def ws(B):
        return {k: v.cpu() for k, v in _ws_fields(lib, B, n, S, P, enc.n_output_dims, int(enc.desc.n_levels)).items()}

    f = ws(direct._buffers(n, 0, poses.device))
    fp = _oracle_params(cfg, model)
    for t in (fp.hash_table, fp.W1, fp.W2, fp.W3, fp.W4):
        t.requires_grad_(True)
    _oracle_iteration(fp, cfg, bbox, f["o"], f["d"], f["z"], f["tgt"], f["td"], False, f["pts"])[2].backward()
    fq = _f64_params(fp)
    with _probe() as pr:
        _oracle_iteration(fq, cfg, bbox, f["o"], f["d"], f["z"], f["tgt"], f["td"], False, f["pts"])[2].backward()
        ties = pr.bounds()
    o32 = [fp.hash_table.grad, fp.W1.grad, fp.W2.grad, fp.W3.grad, fp.W4.grad]
    o64 = [fq.hash_table.grad, fq.W1.grad, fq.W2.grad, fq.W3.grad, fq.W4.grad]
    for which, grads in (("direct", got), ("autograd", ref)):
        for g, a, q, nm in zip(grads, o32, o64, ("d_hash", "dW1", "dW2", "dW3", "dW4")):
            _grad_close(g, a, q, f"{nm} ({which}, map phase)", _level_groups(fp.hash_meta) if nm == "d_hash" else None,
                        tie_bound=ties.get(nm))
    assert float((ref[0] != 0).float().mean()) > 0.001
    # ---- pose phase
    reset()
    poses_all = model.rba(all_index)
    rays_o, rays_d, ts, td = mp._ray_batch(cur, poses_all)
    slam.get_loss_from_ret(model.mapping(rays_o, rays_d, ts, td, clamp=True), smooth=True).backward()
    ref_r, ref_m = grads_of(rba_params), grads_of(params)
    reset()
    direct.pose_gradients(cur, all_index.reshape(-1).contiguous())
    got_r, got_m = grads_of(rba_params), grads_of(params)
    # the pose phase against the oracle chain: pose MLP (oracle/rba_oracle.py) -> rays -> field -> losses, autograd
    K = all_index.shape[0]
    f = ws(direct._buffers(n, K, all_index.device))
    lin = model.rba._linears()
    idc = all_index.reshape(-1).cpu()
    pidx = f["pidx"].long()

    def pose_chain(fpp, dtype):
        prm = [t.detach().cpu().to(dtype).requires_grad_(True) for l in lin for t in (l.weight, l.bias)]
        c2w = RO.rba_forward(prm, model.rba.init_r.cpu().to(dtype), model.rba.init_t.cpu().to(dtype), idc, model.rba.num_cams,
                             float(model.rba.scale))
        o = c2w[pidx, :3, 3]
        d = torch.sum(f["d_cam"].to(dtype)[:, None, :] * c2w[pidx, :3, :3], -1)
        tot = _oracle_iteration(fpp, cfg, bbox, o, d, f["z"], f["tgt"], f["td"], True, f["pts"])[2]
        if FO.MLP_PROBE is not None:                 # the float64 run: also the gradients at the hidden activations (ReLU tie bounds)
            gs = torch.autograd.grad(tot, prm + [fpp.hash_table, fpp.W1, fpp.W2, fpp.W3, fpp.W4, FO.MLP_PROBE["H1"], FO.MLP_PROBE["H3"]])
            FO.MLP_PROBE.update(G1=gs[-2], G3=gs[-1])
            return gs[:-2]
        return torch.autograd.grad(tot, prm + [fpp.hash_table, fpp.W1, fpp.W2, fpp.W3, fpp.W4])

    fp2 = _oracle_params(cfg, model)
    for t in (fp2.hash_table, fp2.W1, fp2.W2, fp2.W3, fp2.W4):
        t.requires_grad_(True)
    p32 = pose_chain(fp2, torch.float32)
    with _probe() as pr:
        p64 = pose_chain(_f64_params(fp2), torch.float64)
        ties = pr.bounds()
    by_lin = lambda grads: [dict(zip(rba_params, grads))[t] for l in lin for t in (l.weight, l.bias)]    # noqa: E731
    for which, grads in (("direct", got_r), ("autograd", ref_r)):
        for g, a, q in zip(by_lin(grads), p32[:8], p64[:8]):
            _grad_close(g, a, q, f"pose MLP grad {tuple(g.shape)} ({which})", k=8.0)
    assert float(ref_r[0].abs().max()) > 0
    for which, grads in (("direct", got_m), ("autograd", ref_m)):
        for g, a, q, nm in zip(grads, p32[8:], p64[8:], ("d_hash", "dW1", "dW2", "dW3", "dW4")):
            _grad_close(g, a, q, f"{nm} ({which}, pose phase)", _level_groups(fp2.hash_meta) if nm == "d_hash" else None,
                        tie_bound=ties.get(nm))
    # ---- the one-call driver (rfx_ba_forward_backward) vs the same iteration issued stage by stage
    direct.stagewise_every = 1
    reset()
    direct.pose_gradients(cur, all_index.reshape(-1).contiguous())
    st_r, st_m = grads_of(rba_params), grads_of(params)
    direct.stagewise_every = 0
    for g, r in zip(got_m[1:], st_m[1:]):
        assert torch.equal(g, r)                              # dW: deterministic reductions, identical kernels
    assert float((got_m[0] - st_m[0]).abs().max()) <= 1e-4 * float(st_m[0].abs().max())      # hash grads: atomic order
    for g, r in zip(got_r, st_r):
        assert float((g - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-12
    # ---- pose phase without the map gradients nobody consumes (the default of Mapper.global_pose): same pose-MLP
    #      gradients, map parameters untouched; both ways of issuing it
    for every in (0, 1):
        direct.stagewise_every = every
        reset()
        direct.pose_gradients(cur, all_index.reshape(-1).contiguous(), map_grads=False)
        assert all(p.grad is None for p in params)
        for g, r in zip(grads_of(rba_params), got_r):
            assert float((g - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-12
    direct.stagewise_every = 0


@pytest.mark.parametrize("unused", [False, True])
def test_pose_phase_leaves_the_map_alone(unused):
    """Mapper.global_pose steps only the pose MLP (reference mp_slam/mapper.py:494-499): with or without the map
    gradients that its backward also produces (mapping.unused_gradients), the map parameters and their optimizer state
    come out of the phase bit-identical, the pose MLP moves, and no gradient is left behind."""
    import random
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    random.seed(5); torch.manual_seed(5)
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 5, "sample": 512, "iters": 2, "BA_iters": 2, "unused_gradients": unused})
    cfg["training"].update({"smooth_pts": 16})
    pipe = MappingPipeline(cfg, n_frames=30, seed=1)
    frames = pipe.prefetch(list(range(12)))
    pipe.start(frames[0])
    for i in range(1, 11):
        pipe.step(i, frames[i])
    mp, model = pipe.mapper, pipe.model
    assert mp._direct_iterations().unused_gradients == unused
    map_params = [model.embed_res_fn.params] + list(model.decoder_res.fused_weights())
    rba_params = list(model.rba.parameters())

    def snap():
        st = mp.map_optimizer.state
        return ([p.detach().clone() for p in map_params] + [st[p]["exp_avg"].clone() for p in map_params]
                + [st[p]["exp_avg_sq"].clone() for p in map_params] + [st[p]["step"].clone() for p in map_params])

    before, rba_before = snap(), [p.detach().clone() for p in rba_params]
    b = mp.dataset[10]
    batch = {k: (v[None, ...] if isinstance(v, torch.Tensor) else torch.tensor([v])) for k, v in b.items()}
    mp.global_pose(batch, 10)
    torch.cuda.synchronize()
    for x, y in zip(before, snap()):
        assert torch.equal(x, y)
    assert any(not torch.equal(x, p.detach()) for x, p in zip(rba_before, rba_params))
    assert all(p.grad is None for p in map_params + rba_params)


def test_error_statuses_of_the_round_one_entry_points():
    """every entry point returns a status instead of faulting on bad arguments (include/rfx.h: error behaviour)."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    st = L.stream_ptr(torch.device("cuda"))
    f = torch.zeros(64, device="cuda")
    i64 = torch.zeros(8, dtype=torch.int64, device="cuda")
    ERR_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE = -1, -3, -4
    # ray batch: more samples than rays in the population
    assert lib.rfx_gather_rays(L.ptr(f), 4, 1, i64.data_ptr(), 5, L.ptr(f), 4, 8, 0, 1, 2, L.ptr(f), 1, L.ptr(f), L.ptr(f), L.ptr(f),
                               L.ptr(f), L.ptr(f), i64.data_ptr(), st) == ERR_ARG
    assert lib.rfx_gather_rays(None, 0, 0, None, 5, None, 0, 0, 0, 1, 2, L.ptr(f), 1, L.ptr(f), L.ptr(f), L.ptr(f), L.ptr(f), L.ptr(f),
                               i64.data_ptr(), st) == 0                                  # empty batch is fine
    # pose MLP: only the reference width is implemented
    prm = L.RbaParams(*([L.ptr(f)] * 8), 128)
    assert lib.rfx_rba_forward(C.byref(prm), L.ptr(f), L.ptr(f), i64.data_ptr(), 2, 4, 0.01, L.ptr(f), L.ptr(f), st) == ERR_UNSUPPORTED
    prm = L.RbaParams(*([L.ptr(f)] * 8), 256)
    assert lib.rfx_rba_forward(C.byref(prm), L.ptr(f), L.ptr(f), i64.data_ptr(), 0, 4, 0.01, L.ptr(f), L.ptr(f), st) == 0    # K = 0
    # marching cubes needs at least one cell
    assert lib.rfx_mc_count(L.ptr(f), None, 1, 4, 4, 0.0, i64.data_ptr(), i64.data_ptr(), st) == ERR_ARG
    # TV lattice
    assert lib.rfx_tv_lattice(L.ptr(f), 0, 0.1, 0.05, L.farr(L._D6, [0, 1, 0, 1, 0, 1]), 1, 1, L.ptr(f), st) == ERR_ARG
    # one-call BA iteration: undersized / misaligned workspace, missing outputs
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.scene_rep import JointEncoding
    import numpy as np
    cfg = synthetic_config("office0")
    m = JointEncoding(cfg, torch.from_numpy(np.array(cfg["mapping"]["bound"])), num_kf=4).cuda()
    d = L.BaDesc()
    d.field, d.sampler = m._field_desc(False), m._sampler_desc()
    d.n_kf_samples, d.n_cur, d.tv_P, d.K, d.hash_entries = 8, 8, 4, 1, 16
    big = torch.zeros(1 << 20, device="cuda")
    for name in ("d_hash", "d_w", "u6", "poses16", "loss_w_dev"):
        setattr(d, name, L.ptr(big))
    need = lib.rfx_ba_workspace_bytes(16, 59, 4, 32, 16)
    assert need > 0 and lib.rfx_ba_workspace_bytes(0, 59, 4, 32, 16) == 0
    base = (big.data_ptr() + 255) // 256 * 256
    assert lib.rfx_ba_forward_backward(C.byref(d), base, need - 1, st) == ERR_WORKSPACE
    assert lib.rfx_ba_forward_backward(C.byref(d), base + 4, need, st) == ERR_ARG
    d.d_hash = None                             # d_hash and d_w come as a pair
    assert lib.rfx_ba_forward_backward(C.byref(d), base, need, st) == ERR_ARG
    d.d_w = None                                # neither: only allowed with somewhere to put the pose gradients
    assert lib.rfx_ba_forward_backward(C.byref(d), base, need, st) == ERR_ARG
    torch.cuda.synchronize()


def test_uniform_draws_match_oracle_bit_exact():
    """rfx_uniform_draws (the uniforms a BA iteration draws for itself, include/rfx.h) against oracle/draws_oracle.py."""
    import numpy as np
    from oracle import draws_oracle as DO
    from remixfusion_amd import _lib as L
    lib = L.load()
    for seed, stream, n in ((1, 0, 6), (0x9E3779B97F4A7C15, 0, 135936), (0xFFFFFFFFFFFFFFFF, 1, 6), (0x0123456789ABCDEF, 1, 70001)):
        out = torch.empty(n, device="cuda")
        L.check(lib.rfx_uniform_draws(seed, stream, n, L.ptr(out), L.stream_ptr(out.device)), "rfx_uniform_draws")
        assert np.array_equal(out.cpu().numpy(), DO.uniform_draws(seed, stream, n)), (seed, stream, n)
    assert lib.rfx_uniform_draws(1, 0, 4, None, None) != 0          # no output buffer


@pytest.mark.parametrize("sample,min_cur", [(128, 128), (512, 1024)])
def test_one_call_iteration_equals_stagewise_on_both_sides_of_its_switches(sample, min_cur):
    """rfx_ba_forward_backward against the same iteration issued through the public per-stage entry points, on the paths the
    big tests do not reach: sample = 128 -> 256 rays x 59 < 16 384 points (no row selection: the losses are finished by a
    launch of their own); 512 keyframe + 1 024 current-frame rays on 3 cameras -> more than 800 rays on the heaviest camera (the
    pose phase's tail as separate stages instead of the one-launch chain).  Same seeds -> same draws (rfx_uniform_draws in the staged
    issue): losses, decoder gradients and d_raw-derived quantities equal, hash / pose-MLP gradients to atomic / grouping order."""
    import random
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    from remixfusion_amd.mp_slam.direct import DirectIterations
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 5, "sample": sample, "iters": 2, "BA_iters": 2, "min_pixels_cur": min_cur})
    cfg["training"].update({"smooth_pts": 16})
    pipe = MappingPipeline(cfg, n_frames=30, seed=1)
    frames = pipe.prefetch(list(range(12)))
    pipe.start(frames[0])
    for i in range(1, 11):
        pipe.step(i, frames[i])
    mp, model, slam = pipe.mapper, pipe.model, pipe.slam
    direct = DirectIterations(mp)
    S = int(cfg["training"]["n_range_d"]) + int(cfg["training"]["n_samples_d"])
    n = direct._n_rays()
    assert n == sample + min_cur and (n * S < 16384) == (sample == 128)
    b = frames[10]
    cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
    K = len(mp.keyframe) + 1
    assert K == 3
    all_index = torch.arange(0, K, device="cuda").unsqueeze(-1)
    params = [model.embed_res_fn.params] + list(model.decoder_res.fused_weights())
    rba_params = list(model.rba.parameters())

    def run(every, phase):
        for p in params + rba_params:
            p.grad = None
        random.seed(21); torch.manual_seed(21)
        direct.stagewise_every = every
        if phase == "map":
            poses = slam.est_c2w_data[0:11:5].clone().cuda().float()
            poses = torch.cat([poses, poses[-1:]], 0)[:K]
            lc = direct.map_gradients(cur, poses).clone()
        else:
            lc = direct.pose_gradients(cur, all_index.reshape(-1).contiguous(), map_grads=False).clone()
        torch.cuda.synchronize()
        return lc, [p.grad.clone() for p in params if p.grad is not None], [p.grad.clone() for p in rba_params if p.grad is not None]

    for phase in ("map", "pose"):
        lc1, m1, r1 = run(0, phase)          # one call
        lc2, m2, r2 = run(1, phase)          # stage by stage
        assert torch.isfinite(lc1).all() and float(lc1[:4].abs().sum()) > 0
        assert float((lc1[:4] - lc2[:4]).abs().max()) <= 1e-6 * float(lc2[:4].abs().max())        # losses (sums grouped differently)
        assert torch.equal(lc1[4:], lc2[4:])                                                       # coefficients: exact counts
        if phase == "map":
            for g, r in zip(m1[1:], m2[1:]):
                assert torch.equal(g, r)                                   # dW: deterministic reductions, identical kernels
            assert float((m1[0] - m2[0]).abs().max()) <= 1e-4 * float(m2[0].abs().max())
        else:
            assert not m1 and not m2
            for g, r in zip(r1, r2):
                assert float((g - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-12
            assert float(r1[0].abs().max()) > 0
    direct.stagewise_every = 0


def test_filter_depth_keyframe_rays_are_drawn_on_the_device_without_a_host_round_trip():
    """KeyFrameDatabase.sample_single_keyframe_rays(option='filter_depth') (reference model/keyframe.py:37-52) with device
    sampling: k distinct rays, all with a valid depth when more than k are valid, any rays when not; the first-frame quirk
    (indices drawn among the valid rays, applied to all rays); and no synchronising call (torch's sync debug mode)."""
    import random
    import warnings
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.keyframe import KeyFrameDatabase
    cfg = synthetic_config("scene0000")
    H, W, k = 48, 64, 500
    kf = KeyFrameDatabase(cfg, H, W, 4, k, torch.device("cuda"))
    assert kf.device_sampling
    g = torch.Generator(device="cuda").manual_seed(0)
    rays = torch.rand((1, H * W, 7), device="cuda", generator=g)
    rays[..., :6] += torch.arange(H * W, device="cuda")[None, :, None]          # every ray recognisable
    depth = torch.rand(H * W, device="cuda", generator=g) * 8.0                 # depth_trunc = 5: ~ 3/8 invalid
    depth[::7] = 0.0
    rays[0, :, 6] = depth
    valid = (depth > 0) & (depth <= cfg["cam"]["depth_trunc"])
    random.seed(1)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = kf.sample_single_keyframe_rays(rays, "filter_depth")
            out_first = kf.sample_single_keyframe_rays(rays, "filter_depth", first=True)
            few = rays.clone()
            few[0, 300:, 6] = 0.0                                               # < k valid rays left: uniform over the frame
            out_few = kf.sample_single_keyframe_rays(few, "filter_depth")
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert out.shape == (k, 7)
    src = out[:, 0].floor().long()                                             # which ray each row is
    assert torch.equal(out, rays[0, src]) and src.unique().numel() == k
    assert bool(valid[src].all())
    # uniform over the valid rays: the picks spread over the whole frame
    assert int(src.min()) < H * W // 10 and int(src.max()) > H * W * 9 // 10
    assert out_first.shape == (1, k, 7)
    src1 = out_first[0, :, 0].floor().long()
    assert src1.unique().numel() == k and int(src1.max()) < int(valid.sum())    # indices among the VALID count, applied to all rays
    assert not bool(valid[src1].all())
    srcf = out_few.reshape(-1, 7)[:, 0].floor().long()
    assert srcf.unique().numel() == k and int((few[0, srcf, 6] == 0).sum()) > 0 and int(srcf.max()) > 300
