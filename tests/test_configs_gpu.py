"""The BASELINE.json configurations that the bench line does not run, on the GPU against the C oracle at their own
sizes: config 1 (room0, 320x240, TSDF only: a stream with a volume move), the moving-volume sizes of configs 4 and 5
(cafeteria 700x700x300 @ 2 cm under a 1280x720 frame; apartment 1600x1600x600 @ 1 cm = 1.5e9 voxels, whole and as the slab
one of eight GPUs owns), config 3 end to end with the tracker on, and the north-star target volume (1000^3 @ 1 cm, >= 30
frames/s on one GPU)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(cfg, i=0, n=None):
    from remixfusion_amd.datasets import get_dataset
    ds = get_dataset(cfg, device="cuda", n_frames=n or (i + 1))
    return ds, ds[i]


def _dev_equal(torch, got, ref_np, what):
    ref = torch.from_numpy(ref_np).to(got.device)
    same = torch.equal(got.view(torch.int32), ref.view(torch.int32))
    if not same:
        bad = int((got.view(torch.int32) != ref.view(torch.int32)).sum())
        raise AssertionError(f"{what}: {bad} of {ref.numel()} voxels differ")


def test_config1_room0_tsdf_only_stream_with_a_volume_move():
    """BASELINE config 1: Replica room0 bound, 320x240, moving volume 200x200x150 @ 4 cm, no neural field.  The pipeline's
    volume (V1 every frame; V7 + V2 when the camera leaves the t_treshold box) against the C oracle doing the same."""
    import torch
    from oracle import tsdf as O
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config("room0_tsdf")
    assert cfg["synthetic"]["tsdf_only"] and (cfg["cam"]["H"], cfg["cam"]["W"]) == (240, 320)
    pipe = MappingPipeline(cfg, n_frames=12)
    assert pipe.model is None and tuple(int(v) for v in pipe.mv.vol_dim) == (200, 200, 150)
    frames = pipe.prefetch(list(range(10)))
    orc = O.load(True)
    dims = tuple(int(v) for v in pipe.mv.vol_dim)
    n = int(np.prod(dims))
    vol = [np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)]
    bnds = np.array(pipe.mv.vol_bnds)
    moves = 0
    for i in range(10):
        b = dict(frames[i])
        if i >= 6:                                  # the camera jumps 1.3 m along x: the volume follows (integer-metre bounds)
            c2w = b["c2w"].clone()
            c2w[0, 3] += 1.3
            b["c2w"], b["c2w_dev"] = c2w, c2w.cuda()
        pipe.step(i, b)
        new_bnds = np.array(pipe.mv.vol_bnds)
        if not np.array_equal(new_bnds, bnds):
            back = [a.copy() for a in vol]
            orc.mv_shift(vol, back, dims, new_bnds[:, 0].astype(np.float32), dims, bnds[:, 0].astype(np.float32), cfg["volume"]["voxel_size"])
            bnds, moves = new_bnds, moves + 1
        orc.mv_integrate(*vol, dims, bnds[:, 0].astype(np.float32), cfg["volume"]["voxel_size"], pipe.K, b["c2w"].numpy(),
                         O.pack_color(b["rgb255"].cpu().numpy()), b["depth"].cpu().numpy(), cfg["volume"]["trunc"])
    torch.cuda.synchronize()
    assert moves == 1
    for g, r, nm in zip(pipe.mv._vols(), vol, ("tsdf", "weight", "colour")):
        _dev_equal(torch, g[:n], r, nm)
    assert float((vol[1] > 0).mean()) > 0.02


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["cafeteria", "apartment"])
def test_moving_volume_sizes_of_configs_4_and_5(name):
    import torch
    from oracle import tsdf as O
    from remixfusion_amd import _lib as L
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.traj import Trajectory
    from remixfusion_amd.model.Volume import moving_volume
    cfg = synthetic_config(name)
    ds, b = _frame(cfg, 4)
    mv = moving_volume(cfg, Trajectory(), ds.poses[0].numpy().astype(np.float64))
    dims = tuple(int(v) for v in mv.vol_dim)
    assert dims == {"cafeteria": (700, 700, 300), "apartment": (1600, 1600, 600)}[name]
    rgb255 = torch.floor(b["rgb"] * 255.0)
    mv.integrate(rgb255, b["depth"], ds.K(), b["c2w"].numpy(), None)
    n = int(np.prod(dims))
    vol = [np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)]
    u, c = O.load(True).mv_integrate_threads(*vol, dims, mv.vol_origin, mv.voxel_size, ds.K(), b["c2w"].numpy(),
                                             O.pack_color(rgb255.cpu().numpy()), b["depth"].cpu().numpy(), mv.trunc_margin, threads=16)
    assert u > 1e6 and c > 1e3        # (cafeteria: a 24 m hall seen from a 14 m volume -- little surface inside)
    for g, r, nm in zip(mv._vols(), vol, ("tsdf", "weight", "colour")):
        _dev_equal(torch, g[:n], r, f"{name} {nm}")
    # the slab rank 3 of 8 (configs 4 / 5 shard the volume over 4 / 8 GPUs) integrates on its own
    world, rank = 8, 3
    x0, x1 = dims[0] * rank // world, dims[0] * (rank + 1) // world
    plane = dims[1] * dims[2]
    H, W = b["depth"].shape
    lib = L.load()
    st = L.stream_ptr()
    t = torch.ones((x1 - x0) * plane, device="cuda")
    w, col = torch.zeros_like(t), torch.zeros_like(t)
    ws = torch.empty((lib.rfx_tsdf_integrate_workspace_bytes(x1 - x0, dims[1], dims[2], H, W) + 3) // 4, device="cuda")
    cpk = torch.empty(H * W, device="cuda")
    L.check(lib.rfx_pack_color(L.ptr(rgb255.reshape(-1, 3).contiguous()), L.ptr(cpk), H * W, st), "pack")
    L.check(lib.rfx_tsdf_integrate_slab(L.ptr(t), L.ptr(w), L.ptr(col), *dims, x0, x1, L.farr(L._F3, mv.vol_origin), mv.voxel_size,
                                        L.farr(L._F9, ds.K().reshape(-1)), L.farr(L._F16, b["c2w"].numpy().reshape(-1)), L.ptr(cpk),
                                        L.ptr(b["depth"].reshape(-1).contiguous()), H, W, float(mv.trunc_margin), 1.0, 1, 0,
                                        L.farr(L._F6, np.zeros(6, np.float32)), 0, L.ptr(ws), ws.numel() * 4, st), "slab")
    sl = slice(x0 * plane, x1 * plane)
    for g, r, nm in zip((t, w, col), vol, ("tsdf", "weight", "colour")):
        _dev_equal(torch, g, r[sl], f"{name} slab {nm}")


@pytest.mark.timeout(900)
def test_north_star_volume_1000_cubed():
    """(10 m)^3 at 1 cm on one GPU: V1 bit-exact against the oracle at 1e9 voxels, and the mapping loop (V1 every frame,
    5 + 5 iterations every 5 frames) above the 30 frames/s target."""
    import torch
    from oracle import tsdf as O
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config("stress10m")
    cfg["mapping"]["first_iters"] = 20
    pipe = MappingPipeline(cfg, n_frames=60)
    dims = tuple(int(v) for v in pipe.mv.vol_dim)
    assert dims == (1000, 1000, 1000)
    frames = pipe.prefetch(list(range(46)))
    pipe.start(frames[0])
    pipe.sync_volume()
    torch.cuda.synchronize()
    n = int(np.prod(dims))
    vol = [np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)]
    b = frames[0]
    u, c = O.load(True).mv_integrate_threads(*vol, dims, pipe.mv.vol_origin, pipe.mv.voxel_size, pipe.K, b["c2w"].numpy(),
                                             O.pack_color(b["rgb255"].cpu().numpy()), b["depth"].cpu().numpy(), pipe.mv.trunc_margin,
                                             threads=16)
    assert u > 1e7
    for g, r, nm in zip(pipe.mv._vols(), vol, ("tsdf", "weight", "colour")):
        _dev_equal(torch, g[:n], r, f"1000^3 {nm}")
    del vol
    for i in range(1, 6):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6, 46):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    fps = 40 / (time.perf_counter() - t0)
    print(f"stress10m (1000^3 voxels @ 1 cm, 640x480, {u} voxels updated by frame 0): {fps:.0f} frames/s")
    assert fps >= 30.0, fps


@pytest.mark.timeout(900)
def test_config3_scene0000_mapping_with_the_tracker_on():
    """BASELINE config 3 end to end at its own sizes (620x460, moving volume 250x250x150 @ 4 cm, T = 2^19, 117 samples per
    ray, 63^3-point TV lattice): poses come from the ROTracker, not from the ground truth, and the search runs on the
    REFERENCE's particle templates (round 5: tests/golden/pst_templates.npz, the 60 arrays of PFO/fps_uniform_sphere): the check
    is that the loop runs, keeps the pose error bounded on the depth-observable axis and leaves finite maps; kernel parity of
    the tracker is test_tracker_gpu.py's."""
    import random
    import warnings
    import torch
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    random.seed(0)
    cfg = synthetic_config("scene0000")
    cfg["synthetic"]["tracker"] = True
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
    cfg["mapping"]["first_iters"] = 30
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pipe = MappingPipeline(cfg, n_frames=24)
    assert pipe.tracker is not None and tuple(int(v) for v in pipe.mv.vol_dim) == (250, 250, 150)
    # the reference's particles (its TIFF directory, or the archive of the same arrays that travels with the repository): never generated
    assert pipe.tracker.RO_Tracker.PST_source.endswith(("fps_uniform_sphere", "pst_templates.npz"))
    frames = pipe.prefetch(list(range(16)))
    pipe.start(frames[0])
    for i in range(1, 16):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    assert int(pipe.slam.mapping_idx[0]) == 10 and int(pipe.slam.tracking_idx[0]) == 15
    ke = cfg["mapping"]["keyframe_every"]
    errs = []
    for i in (5, 10, 15):
        est = pipe.slam.est_c2w_data[i] if i % ke == 0 else pipe.slam.est_c2w_data_rel[i] @ pipe.slam.est_c2w_data[(i // ke) * ke]
        gt = frames[i]["c2w"]
        assert bool(torch.isfinite(est).all())
        errs.append(abs(float((est[:3, 3].cpu() - gt[:3, 3]) @ gt[:3, 2])))
    print("scene0000 + tracker: |translation error along the view axis| (m) at frames 5/10/15:", [round(e, 4) for e in errs])
    assert max(errs) < 0.10
    assert bool(torch.isfinite(pipe.model.embed_res_fn.params).all()) and bool(torch.isfinite(pipe.model.GBV.params).all())
    assert float((pipe.mv.weight_vol_gpu > 0).float().mean()) > 0.01


@pytest.mark.timeout(900)
def test_config3_tracker_follows_a_120_frame_sequence():
    """Sequence-level evidence for the ROTracker (reference model/ROtracker.py:713-866, mp_slam/tracker.py:55-134): 120 frames of
    the scene0000-sized stream with the tracker on, absolute trajectory error against the synthetic ground truth (no alignment:
    both start from the same pose).  The room is furnished (synthetic.clutter = 48 spheres): in the bare box room a translation
    along a flat wall changes no depth and a geometric tracker slides along it (measured: 1.8 cm of every 3 cm step lost,
    ATE 28 cm after 120 frames; tools/tracker_dbg.py), which says nothing about the tracker.  The search uses the reference's
    particle templates (round 5: the archive of its 60 PST arrays travels with the repository; round 4 measured this test on
    generated ones: ATE rmse 2.2-2.4 cm, max 3.3 cm, rotation rmse 0.8 deg at 4 cm voxels)."""
    import random
    import warnings
    import numpy as np
    import torch
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    N = 120
    random.seed(0)
    cfg = synthetic_config("scene0000")
    cfg["synthetic"].update({"tracker": True, "depth_noise": 0.0, "dropout": 0.0, "clutter": 48})
    cfg["mapping"]["first_iters"] = 50
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pipe = MappingPipeline(cfg, n_frames=N + 8)
    frames = pipe.prefetch(list(range(N)))
    pipe.start(frames[0])
    for i in range(1, N):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    assert int(pipe.slam.tracking_idx[0]) == N - 1 and int(pipe.slam.mapping_idx[0]) == N - 5
    ke = cfg["mapping"]["keyframe_every"]
    err, rot = [], []
    for i in range(1, N):
        est = pipe.slam.est_c2w_data[i] if i % ke == 0 else pipe.slam.est_c2w_data_rel[i] @ pipe.slam.est_c2w_data[(i // ke) * ke]
        est, gt = est.cpu().double(), frames[i]["c2w"].double()
        assert bool(torch.isfinite(est).all())
        err.append(float((est[:3, 3] - gt[:3, 3]).norm()))
        R = est[:3, :3].T @ gt[:3, :3]
        rot.append(float(torch.rad2deg(torch.acos(((R.trace() - 1) / 2).clamp(-1, 1)))))
    err, rot = np.array(err), np.array(rot)
    path = sum(float((frames[i]["c2w"][:3, 3] - frames[i - 1]["c2w"][:3, 3]).norm()) for i in range(1, N))
    rmse = float(np.sqrt((err ** 2).mean()))
    print(f"scene0000 + tracker, {N} frames, path {path:.2f} m: ATE rmse {rmse * 100:.2f} cm, max {err.max() * 100:.2f} cm, "
          f"final {err[-1] * 100:.2f} cm; rotation rmse {float(np.sqrt((rot ** 2).mean())):.2f} deg")
    assert path > 1.5
    assert rmse < 0.05 and err.max() < 0.08 and err[-1] < 0.06          # bounded, no drift: the last frame is no worse than the mean
    assert float(np.sqrt((rot ** 2).mean())) < 2.0
    assert bool(torch.isfinite(pipe.model.embed_res_fn.params).all())
