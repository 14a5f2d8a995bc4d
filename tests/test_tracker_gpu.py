"""GPU parity + behaviour tests of the tracker kernels against the C oracle."""
import numpy as np
import pytest
import torch

from conftest import small_frame

pytestmark = pytest.mark.gpu


def _setup(H=120, W=160, frame=4):
    from oracle import tsdf as O
    K, c2w, rgb255, depth, _ = small_frame(H=H, W=W, frame=frame)
    dims, origin, voxel, trunc = (100, 100, 75), np.array([-4, -5, -3], np.float32), 0.08, 0.24
    n = int(np.prod(dims))
    t, w, c = np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    orc = O.load()
    for f in (2, 3, 4):
        Kf, pf, rf, df, _ = small_frame(H=H, W=W, frame=f)
        orc.mv_integrate(t, w, c, dims, origin, voxel, Kf, pf, O.pack_color(rf), df, trunc)
    return orc, K, c2w, depth, t, dims, origin, voxel, trunc


def test_vertex_normal_evaluate_match_oracle():
    from remixfusion_amd import _lib as L
    lib = L.load()
    orc, K, c2w, depth, tsdf, dims, origin, voxel, trunc = _setup()
    H, W = depth.shape
    rng = np.random.default_rng(0)
    for sample_range in (0.0, 0.5, 2.0):
        u = rng.uniform(1e-3, 1.0, (H, 2)).astype(np.float32)
        ref_v = orc.tr_vertex(depth, K, 2.5, trunc, sample_range, u)          # cut_dist 2.5 m removes far pixels
        d_dep = torch.from_numpy(depth).cuda().reshape(-1)
        d_v = torch.ones(H * W * 4, device="cuda")
        L.check(lib.rfx_track_vertex(L.ptr(d_dep), L.ptr(d_v), L.farr(L._F9, K.reshape(-1)), H, W, 2.5, trunc, sample_range,
                                     7, L.ptr(torch.from_numpy(u).cuda()), L.stream_ptr()), "vertex")
        assert np.array_equal(d_v.cpu().numpy().reshape(-1, 4).view(np.uint32), ref_v.view(np.uint32))
    # from here on: the configuration the reference runs (sample_range 0, cut_dist 8 m)
    u = np.full((H, 2), 0.5, np.float32)
    ref_v = orc.tr_vertex(depth, K, 8.0, trunc, 0.0, u)
    L.check(lib.rfx_track_vertex(L.ptr(d_dep), L.ptr(d_v), L.farr(L._F9, K.reshape(-1)), H, W, 8.0, trunc, 0.0, 7, None,
                                 L.stream_ptr()), "vertex")
    assert np.array_equal(d_v.cpu().numpy().reshape(-1, 4).view(np.uint32), ref_v.view(np.uint32))
    # normals (borders keep their initial ones, like the reference's np.ones buffer)
    ref_n = orc.tr_normal(ref_v, H, W)
    d_n = torch.ones(H * W * 3, device="cuda")
    L.check(lib.rfx_track_normal(L.ptr(d_v), L.ptr(d_n), H, W, L.stream_ptr()), "normal")
    got_n = d_n.cpu().numpy().reshape(H, W, 3)
    assert np.array_equal(got_n[1:-1, 1:-1].view(np.uint32), ref_n.reshape(H, W, 3)[1:-1, 1:-1].view(np.uint32))
    assert (got_n[0] == 1).all() and (got_n[:, 0] == 1).all()
    # evaluate: small perturbations around the true pose, three pyramid levels
    from remixfusion_amd.model.ROtracker import make_pst
    q6 = make_pst(1024, 3)
    ss = np.full(6, 0.02, np.float32)
    d_t = torch.from_numpy(tsdf).cuda()
    d_q = torch.from_numpy(q6).cuda()
    nrm_full = got_n.reshape(-1, 3)
    for level, li in ((32, 5), (16, 10), (8, 3)):
        ref_val, ref_cnt, ref_q30 = orc.tr_evaluate(tsdf, dims, origin, voxel, ref_v, nrm_full, c2w[:3, :3], c2w[:3, 3], q6, ss, K, H, W, level, li)
        val, cnt = torch.empty(1024, dtype=torch.int64, device="cuda"), torch.empty(1024, dtype=torch.int64, device="cuda")
        L.check(lib.rfx_track_evaluate(L.ptr(d_t), *dims, L.farr(L._F3, origin), voxel, L.ptr(d_v), L.ptr(d_n),
                                       L.farr(L._F9, c2w[:3, :3].reshape(-1)), L.farr(L._F3, c2w[:3, 3]), L.ptr(d_q), L.farr(L._F6, ss),
                                       1024, L.farr(L._F9, K.reshape(-1)), H, W, level, li, L.ptr(val), L.ptr(cnt), L.stream_ptr()), "eval")
        assert np.array_equal(cnt.cpu().numpy(), ref_cnt.astype(np.int64))      # hit counts are exact
        # sums (ABI 8): every term truncated to 2^-30 and added as an integer -- the oracle's fixed-point sums EXACTLY, whatever
        # the order; and they are the float running sum of the reference's arithmetic to that sum's own rounding
        assert np.array_equal(val.cpu().numpy(), ref_q30)
        assert np.allclose(ref_q30 * 2.0 ** -30, ref_val, rtol=2e-6, atol=1e-6)
        assert ref_cnt.max() > 5
        # the volume in x-slabs (one scene on several GPUs): each slab adds the pixels whose nearest voxel it holds
        plane = dims[1] * dims[2]
        sv, sc = np.zeros(1024, np.int64), np.zeros(1024, np.int64)
        for x0, x1 in ((0, 33), (33, 34), (34, 100)):
            v2, c2 = torch.empty(1024, dtype=torch.int64, device="cuda"), torch.empty(1024, dtype=torch.int64, device="cuda")
            L.check(lib.rfx_track_evaluate_slab(L.ptr(d_t[x0 * plane:x1 * plane]), *dims, x0, x1, L.farr(L._F3, origin), voxel, L.ptr(d_v),
                                                L.ptr(d_n), L.farr(L._F9, c2w[:3, :3].reshape(-1)), L.farr(L._F3, c2w[:3, 3]), L.ptr(d_q),
                                                L.farr(L._F6, ss), 1024, L.farr(L._F9, K.reshape(-1)), H, W, level, li, L.ptr(v2), L.ptr(c2),
                                                L.stream_ptr()), "eval slab")
            sv += v2.cpu().numpy(); sc += c2.cpu().numpy()
        assert np.array_equal(sc, ref_cnt.astype(np.int64)) and np.array_equal(sv, ref_q30)      # slabs add up to the whole volume EXACTLY
    # the null candidate at the true pose fits better than most perturbed ones
    mean = ref_val / (ref_cnt + 1e-6)
    assert mean[0] <= np.percentile(mean[1:], 30)


def test_rotracker_recovers_a_perturbed_pose():
    """the full random-optimisation loop pulls a perturbed initial pose back towards the truth."""
    import random
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.datasets import get_dataset
    from remixfusion_amd.model.ROtracker import ROTracker
    random.seed(0)
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 240, "W": 320, "fx": 288.0, "fy": 288.0, "cx": 159.5, "cy": 119.5})
    cfg["volume"].update({"voxel_size": 0.02, "trunc": 0.06})
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
    ds = get_dataset(cfg, device="cuda", n_frames=12)
    tr = ROTracker(cfg, ds)
    for i in range(1, 6):                                     # build some map with true poses
        b = ds[i]
        tr.post_processing(i, b["c2w"].numpy(), torch.floor(b["rgb"] * 255.0), b["depth"], None)
    b = ds[6]
    gt = b["c2w"].numpy()
    init = gt.copy()
    fwd = gt[:3, 2]                                          # viewing direction: the depth-observable translation
    init[:3, 3] += 0.03 * fwd
    ang = 0.02
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    init[:3, :3] = Rz @ init[:3, :3]
    est, _, _ = tr.do_tracking(init, None, b, "cuda")
    e0 = abs(float((init[:3, 3] - gt[:3, 3]) @ fwd))
    e1 = abs(float((est[:3, 3] - gt[:3, 3]) @ fwd))
    r0 = np.arccos(np.clip((np.trace(init[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1))
    r1 = np.arccos(np.clip((np.trace(est[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1))
    # (translation parallel to the flat walls of the synthetic room is not observable from depth)
    assert e1 < 0.5 * e0 and r1 < 0.6 * r0, (e0, e1, r0, r1)
    # the optimisation never accepts a worse fit: the null candidate's fitness is monotone
    tr.transform_candidate = tr.get_PST(tr.tiff_index[0])
    tr._cand_dev = tr._get_PST_dev(tr.tiff_index[0])
    tr.init_searchsize()
    tr.current_global_R, tr.current_global_T = init[:3, :3].copy(), init[:3, 3].copy()
    f_init = tr.evaluate_tsdf(6, 16, 10240, tr.K, 3)[0][0]
    tr.current_global_R, tr.current_global_T = est[:3, :3].copy(), est[:3, 3].copy()
    f_est = tr.evaluate_tsdf(6, 16, 10240, tr.K, 3)[0][0]
    assert f_est < f_init


def _tracker_with_a_map(device_search):
    import random
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.datasets import get_dataset
    from remixfusion_amd.model.ROtracker import ROTracker
    random.seed(0)
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 240, "W": 320, "fx": 288.0, "fy": 288.0, "cx": 159.5, "cy": 119.5})
    cfg["volume"].update({"voxel_size": 0.02, "trunc": 0.06})
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0, "clutter": 24})
    cfg["RO"]["device_search"] = device_search
    ds = get_dataset(cfg, device="cuda", n_frames=12)
    tr = ROTracker(cfg, ds)
    for i in range(1, 6):
        b = ds[i]
        tr.post_processing(i, b["c2w"].numpy(), torch.floor(b["rgb"] * 255.0), b["depth"], None)
    b = ds[6]
    gt = b["c2w"].numpy()
    init = gt.copy()
    init[:3, 3] += 0.03 * gt[:3, 2]
    ang = 0.02
    init[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32) @ init[:3, :3]
    return tr, b, gt, init


def test_device_search_iterations_follow_the_host_loop_step_by_step():
    """rfx_track_search_* (the 20 iterations of random_optimization on the device) against the host loop, ONE ITERATION AT A
    TIME on the same inputs: T3 from the device state gives the sums rfx_track_evaluate gives for the same pose and box (hit
    counts and fixed-point sums exactly); fed those very sums, the host's cal_transform + bookkeeping (model/ROtracker.py `_search_step`,
    reference :606-709, :745-826, :493-534) and rfx_track_search_update leave the same pose, search box, template index,
    pixel offset and flags (bit for bit; R to one float32 ulp) -- every iteration of a 20-iteration search, including failed ones (a second search starts 50 m away,
    where no vertex meets the volume and no candidate beats the null candidate)."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    tr, b, gt, init = _tracker_with_a_map(True)
    st_ptr = L.stream_ptr(tr.device)
    n_failed = n_succeeded = 0
    far = init.copy()
    far[:3, 3] += 50.0
    for start, box in ((init, None), (far, None)):
        tr.current_global_R = np.asarray(start[:3, :3], np.float32).copy()
        tr.current_global_T = np.asarray(start[:3, 3], np.float32).copy()
        tr.init_searchsize()
        tr.init_depth_vertex(b["depth"].squeeze(), tr.K)
        tr.init_normal()
        s = tr._search_desc(tr.K)
        s.beta = 0.9
        L.check(lib.rfx_track_search_begin(C.byref(s), L.farr(L._F9, tr.current_global_R.reshape(-1)), L.farr(L._F3, tr.current_global_T),
                                           L.farr(L._F6, tr.search_size), st_ptr), "begin")
        host = {"previous_success": False, "success": False, "count_particle": 0, "level_index": 5}
        for i in range(20):
            if not host["success"]:
                host["count_particle"] = 0
            cp = host["count_particle"]
            state = tr._search_state.cpu().numpy()
            flags = state.view(np.int32)
            assert flags[32] == cp and flags[33] == host["level_index"] and flags[39] == i
            # ---- the evaluation: from the device state == from host arguments
            L.check(lib.rfx_track_search_evaluate(C.byref(s), st_ptr), "evaluate")
            sums = tr._search_sums.cpu().numpy().copy()
            tr.transform_candidate = tr.get_PST(tr.tiff_index[cp])
            tr._cand_dev = tr._get_PST_dev(tr.tiff_index[cp])
            P = int(tr.PST_size[cp % 3] // 1024) * 1024
            _, sv, sc = tr.evaluate_tsdf(6, tr.depth_level[cp], tr.PST_size[cp % 3], tr.K, host["level_index"])
            dev_v = (sums[0].astype(np.float64) * 2.0 ** -30).astype(np.float32)
            assert np.array_equal(sums[1, :P].astype(np.float32), sc[:P]) and (sc[:P].max() > 50) == (start is init)
            assert np.array_equal(dev_v[:P], sv[:P])                  # order-independent sums: the two evaluations agree bit for bit
            assert not sums[:, P:].any()
            # ---- the update: host step on the device's sums
            n_all = tr.transform_candidate.shape[0]
            dv, dc = np.zeros(n_all, np.float32), np.zeros(n_all, np.float32)
            dv[:P], dc[:P] = dev_v[:P], sums[1, :P].astype(np.float32)
            tr._search_step(i, host, dv / (dc + 1e-6), 0.9)
            L.check(lib.rfx_track_search_update(C.byref(s), i, st_ptr), "update")
            state = tr._search_state.cpu().numpy()
            flags = state.view(np.int32)
            tag = (start is init, i, host)
            assert bool(flags[34]) == host["success"] and bool(flags[35]) == host["previous_success"], tag
            assert flags[32] == (host["count_particle"] if host["success"] else 0) and flags[33] == host["level_index"], tag
            assert flags[37] == 0
            n_failed += not host["success"]
            n_succeeded += bool(host["success"])
            assert np.abs(state[0:9].reshape(3, 3) - tr.current_global_R).max() <= 1.2e-7, tag     # float32 3x3 product: BLAS may fuse
            assert np.array_equal(state[9:12], tr.current_global_T), tag                           # everything else bit for bit
            assert np.array_equal(state[12:18], tr.search_size), tag
            assert np.array_equal(state[18:24], tr.previous_search_size), tag
            assert state[24] == np.float32(host["min_tsdf"]), tag
            assert not tr._search_sums.any()                                                  # zeroed for the next evaluation
            # continue from the device's numbers, so that every iteration is compared on identical inputs
            tr.current_global_R, tr.current_global_T = state[0:9].reshape(3, 3).copy(), state[9:12].copy()
            tr.search_size[:] = state[12:18]
            tr.previous_search_size[:] = state[18:24]
    assert n_succeeded >= 10 and n_failed >= 10, (n_succeeded, n_failed)


def test_device_search_finds_the_pose_the_host_loop_finds():
    """a whole frame: rfx_track_search_run (one device->host copy) against the host loop from the same perturbed pose -- both
    recover it, to the same accuracy.  (The evaluation sums no longer depend on the order of their additions (ABI 8), but the
    host composes R with numpy's float32 3x3 product, which may fuse where the device rounds every product: one ulp of R per
    successful iteration, which the search amplifies like any perturbation of its fitness values.)"""
    est = {}
    for dev in (True, False):
        tr, b, gt, init = _tracker_with_a_map(dev)
        est[dev], _, _ = tr.do_tracking(init, None, b, "cuda")
        if dev:
            assert tr.search_successes >= 5
    fwd = gt[:3, 2]
    e0 = abs(float((init[:3, 3] - gt[:3, 3]) @ fwd))
    for dev in (True, False):
        e1 = abs(float((est[dev][:3, 3] - gt[:3, 3]) @ fwd))
        r1 = np.arccos(np.clip((np.trace(est[dev][:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1))
        assert e1 < 0.5 * e0 and r1 < 0.012, (dev, e0, e1, r1)
    assert np.abs(est[True][:3, 3] - est[False][:3, 3]).max() < 2e-3
    assert np.abs(est[True][:3, :3] - est[False][:3, :3]).max() < 2e-3


def test_device_search_reports_an_invalid_template_and_refuses_bad_descriptors():
    """a selected candidate whose scaled quaternion vector part is longer than 1: the reference prints and exits
    (model/ROtracker.py:662-669); the host loop raises, the device search raises after its one read of the state (flag 37).
    Descriptors the kernels are not built for are refused before any launch."""
    import ctypes as C
    from remixfusion_amd import _lib as L
    lib = L.load()
    for dev in (True, False):
        tr, b, gt, init = _tracker_with_a_map(dev)
        for cls, t in tr.ALL_PST_dev.items():                   # vector parts of +-60: |q| > 1 at any search size >= 0.02
            t[..., 3:6] = torch.where(t[..., 3:6] >= 0, 60.0, -60.0)
            t[:, 0, :] = 0
            tr.ALL_PST[cls] = t.cpu().numpy()
        with pytest.raises(ValueError, match="invalid quaternion"):
            tr.do_tracking(init, None, b, "cuda")
    tr, b, gt, init = _tracker_with_a_map(True)
    tr.init_depth_vertex(b["depth"].squeeze(), tr.K)
    tr.init_normal()
    st = L.stream_ptr(tr.device)
    args = (L.farr(L._F9, np.eye(3).reshape(-1)), L.farr(L._F3, np.zeros(3)), L.farr(L._F6, np.full(6, 0.02)))
    s = tr._search_desc(tr.K)
    assert lib.rfx_track_search_run(C.byref(s), *args, 1, st) == 0
    for field, value in (("count_search", 0), ("count_search", L.RFX_TRACK_MAX_COUNT_SEARCH + 1), ("x1", 10 ** 6), ("voxel", 0.0), ("state", None)):
        s = tr._search_desc(tr.K)
        setattr(s, field, value)
        assert lib.rfx_track_search_run(C.byref(s), *args, 1, st) == -1, field          # RFX_ERR_ARG
    s = tr._search_desc(tr.K)
    s.template_rows[3] = 16 * 1024 + 1
    assert lib.rfx_track_search_evaluate(C.byref(s), st) == -1
    s = tr._search_desc(tr.K)
    s.n_eval[0] = s.template_rows[0] + 1
    assert lib.rfx_track_search_update(C.byref(s), 0, st) == -1
    torch.cuda.synchronize()


def test_tracker_pipeline_is_reproducible_and_takes_unprefetched_frames():
    """Two things that round 5 made testable.  (1) The tracker's evaluation sums no longer depend on the order of their
    additions (ABI 8), so two runs of one stream give the SAME trajectory, bit for bit (up to round 4 float atomics let them
    drift millimetres apart within a few frames).  (2) A frame handed to `step()` straight from the dataset -- produced on the
    current stream just before the call, not prefetched -- is read by the tracker's own stream only after that stream has
    been ordered behind its producer (pipeline.py: wait_stream + record_stream, as for the volume's stream): the same
    trajectory again."""
    import random
    import warnings
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    N = 9

    def run(prefetch):
        random.seed(0)
        cfg = synthetic_config("office0")
        cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
        cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
        cfg["mapping"].update({"first_iters": 6, "sample": 512})
        cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.02, "tracker": True, "clutter": 32})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pipe = MappingPipeline(cfg, n_frames=N + 4, seed=5)
        assert pipe.track_stream is not None
        if prefetch:
            frames = pipe.prefetch(list(range(N)))
            pipe.start(frames[0])
            for i in range(1, N):
                pipe.step(i, frames[i])
        else:
            pipe.start(pipe.dataset[0])
            for i in range(1, N):
                pipe.step(i, pipe.dataset[i])          # rendered on the current stream right here
        torch.cuda.synchronize()
        return pipe.slam.RO_c2w_data[:N].detach().cpu().clone()

    a, b, c = run(True), run(True), run(False)
    assert float((a[1:, :3, 3] - a[0, :3, 3]).norm(dim=1).max()) > 0.02       # the camera moved and was followed
    assert torch.equal(a, b)
    assert torch.equal(a, c)
