"""ONE scene over several GPUs, on the real kernels: slab forms of V1 / V2 against the whole volume (bit
for bit), and a world-2 run of the sharded pipeline -- two processes on this one GPU, gloo rendezvous on 127.0.0.1, device
tensors staged through the host -- against the single-process pipeline: moving volume identical, global volume identical,
losses and field parameters within the noise of float atomics."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _small_cfg(version="center", shard_field="auto"):
    from remixfusion_amd.config import synthetic_config
    cfg = synthetic_config("office0")
    cfg["mapping"]["shard_field"] = shard_field
    cfg["volume"]["version"] = version
    if version == "more":      # the layout the reference's 'more' logic is written for: one axis fixed to a range (no shipped config
        cfg["volume"]["z_config"] = {"fix": 1, "len": 3, "range": [-3, 3]}     # selects 'more'; without a fixed axis it leaves z empty)
        cfg["volume"]["third_len"] = cfg["volume"]["second_len"]              # same dimensions whichever axis leads (fixed allocation)
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 6, "sample": 512})
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.02})
    cfg["pipeline"] = {"mv_stream": False}
    return cfg


def test_shift_slabs_equal_the_whole_volume():
    """rfx_tsdf_shift_slab: each slab of the new volume gathered from exactly the old planes rfx_tsdf_shift_source_planes
    names (as if fetched from their owners) == rfx_tsdf_shift on the whole volume, for moves along every axis."""
    import ctypes as C
    import torch
    from remixfusion_amd import _lib as L
    lib = L.load()
    dims = (48, 40, 36)
    n = int(np.prod(dims))
    plane = dims[1] * dims[2]
    g = torch.Generator().manual_seed(3)
    old = [torch.rand(n, generator=g).cuda() for _ in range(3)]
    voxel = 0.05
    o_old = np.array([-1.0, -1.0, -1.0], np.float32)
    st = L.stream_ptr()
    for shift in ((1.0, 0.0, 0.0), (-1.0, 0.0, 0.0), (0.0, 1.0, -1.0), (2.0, -1.0, 0.0), (3.0, 0.0, 0.0)):
        o_new = (o_old + np.array(shift, np.float32)).astype(np.float32)
        whole = [torch.empty(n, device="cuda") for _ in range(3)]
        L.check(lib.rfx_tsdf_shift(*[L.ptr(t) for t in whole], *dims, L.farr(L._F3, o_new), *[L.ptr(t) for t in old], *dims,
                                   L.farr(L._F3, o_old), voxel, 0, st), "shift")
        for x0, x1 in ((0, 11), (11, 12), (12, 48)):
            a, b = C.c_int(0), C.c_int(0)
            L.check(lib.rfx_tsdf_shift_source_planes(x0, x1, L.farr(L._F3, o_new), dims[0], L.farr(L._F3, o_old), voxel, C.byref(a), C.byref(b)), "planes")
            a, b = a.value, b.value
            stage = [t[a * plane:b * plane].clone() for t in old]            # "received from the owners"
            slab = [torch.full(((x1 - x0) * plane,), -7.0, device="cuda") for _ in range(3)]
            L.check(lib.rfx_tsdf_shift_slab(*[L.ptr(t) for t in slab], *dims, x0, x1, L.farr(L._F3, o_new),
                                            *[(L.ptr(t) if b > a else None) for t in stage], *dims, a, b, L.farr(L._F3, o_old), voxel, 0, st),
                    "shift_slab")
            torch.cuda.synchronize()
            for k in range(3):
                assert torch.equal(slab[k], whole[k][x0 * plane:x1 * plane]), (shift, x0, x1, k)
        overlap = float((whole[1] != 0).float().mean())       # the volume is 2.4 x 2.0 x 1.8 m: a 3 m move leaves nothing
        assert (overlap == 0.0) if abs(shift[0]) >= 3 else (overlap > 0.05)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_FRAMES = 12
FAR_POSE_DX = 1.4          # metres: beyond volume.t_treshold, so the volume follows the camera (V7 + V2 across slabs)


def _run(pipe, frames, out):
    import torch
    pipe.start(frames[0])
    losses = []
    for i in range(1, N_FRAMES):
        pipe.step(i, frames[i])
    d = pipe.mapper._direct_iterations()
    # one more map and one more pose iteration, losses returned
    batch = pipe.dataset[N_FRAMES - 1]
    rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], -1).reshape(-1, 7).to(pipe.device)
    poses = pipe.slam.est_c2w_data[0:N_FRAMES:pipe.config["mapping"]["keyframe_every"]].clone()
    losses.append(d.map_gradients(rays, poses).clone().cpu())
    pipe.slam.map_optimizer.step()
    pipe.mapper.sync_field()                     # (sharded scene, table partitioned by level: whole again on every rank)
    out["mode"] = type(d).__name__
    out["losses"] = torch.stack(losses)
    out["hash"] = pipe.model.embed_res_fn.params.detach().cpu().clone()
    out["w1"] = pipe.model.decoder_res.fused_weights()[0].detach().cpu().clone()
    out["gbv"] = pipe.model.GBV.params.detach().cpu().clone()
    out["poses"] = pipe.slam.est_c2w_data[:N_FRAMES].detach().cpu().clone()
    # then move the volume: a pose FAR_POSE_DX further along x
    far = frames[N_FRAMES - 1]["c2w"].clone().numpy().astype(np.float64)
    far[0, 3] += FAR_POSE_DX
    moved, _ = pipe.mv.check_move_volume_new(N_FRAMES, far, pipe.traj, version=pipe.config["volume"]["version"])
    out["moved"] = bool(moved)
    out["bnds"] = np.array(pipe.mv.vol_bnds)
    if pipe.config["volume"]["version"] == "more":
        # version 'more': the camera now looks along another world axis -> the box is re-laid along it, and (a quirk kept from
        # the reference, model/Volume.py:1078) re-gridded from the BACK buffers without a fresh copy_volume()
        from conftest import look_at
        eye = far[:3, 3]
        fwd0 = far[:3, 2]
        turn = np.array([-fwd0[1], fwd0[0], 0.0])
        turn /= np.linalg.norm(turn)
        if turn[np.argmax(np.abs(turn))] < 0:      # (looking down a negative axis the reference's formula inverts the bounds)
            turn = -turn
        rot = look_at(tuple(eye), tuple(turn)).astype(np.float64)
        moved2, _ = pipe.mv.check_move_volume_new(N_FRAMES + 1, rot, pipe.traj, version="more")
        out["moved_more"] = bool(moved2)
        out["bnds_more"] = np.array(pipe.mv.vol_bnds)


def _worker(rank, world, port, out_dir, version, shard_field="auto"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from remixfusion_amd.dist import ShardedPipeline
    pipe = ShardedPipeline(_small_cfg(version, shard_field), dist, rank, world, n_frames=N_FRAMES + 4, seed=5)
    frames = pipe.prefetch(list(range(N_FRAMES)))
    out = {}
    _run(pipe, frames, out)
    whole = pipe.mv.gather_whole()
    out["mv"] = [torch.from_numpy(a.copy()) for a in whole]
    out["slab"] = pipe.mv._slab()
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("version,shard_field", [("center", "levels"), ("center", "replicas"), ("more", "levels")])
def test_sharded_scene_world2_equals_single_gpu(tmp_path, version, shard_field):
    import torch
    import torch.multiprocessing as mp
    from remixfusion_amd.pipeline import MappingPipeline
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), version, shard_field), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world))
    # ---- the single-GPU run
    pipe = MappingPipeline(_small_cfg(version), n_frames=N_FRAMES + 4, seed=5)
    frames = pipe.prefetch(list(range(N_FRAMES)))
    ref = {}
    _run(pipe, frames, ref)
    if version == "more":
        assert ref["moved_more"] and r0["moved_more"] and r1["moved_more"]
        assert np.array_equal(ref["bnds_more"], r0["bnds_more"]) and not np.array_equal(ref["bnds_more"], ref["bnds"])
    mv = [torch.from_numpy(a.copy()) for a in pipe.mv.get_volume_all()]
    assert ref["moved"] and r0["moved"] and r1["moved"] and np.array_equal(ref["bnds"], r0["bnds"])
    assert r0["slab"] == (0, 100) and r1["slab"] == (100, 200)
    # moving volume (after 12 integrated frames AND a move across the slab boundary): bit for bit, on both ranks
    for k, name in enumerate(("tsdf", "weight", "colour")):
        assert torch.equal(r0["mv"][k], mv[k]), name
        assert torch.equal(r1["mv"][k], mv[k]), name
    assert float((mv[1] > 0).float().mean()) > 0.01
    assert r0["mode"] == r1["mode"] == ("LevelShardedIterations" if shard_field == "levels" else "ShardedIterations")
    # the two ranks' fields identical to each other (levels: each level has ONE owner, copies synchronised; replicas:
    # all-reduced gradients, same Adam step): bit for bit
    for key in ("hash", "w1", "gbv", "poses", "losses"):
        assert torch.equal(r0[key], r1[key]), key
    # ... and equal to the single-GPU run up to the order of floating-point sums (atomics, all-reduce)
    assert torch.equal(r0["gbv"], ref["gbv"])
    dl = (r0["losses"][:, :4] - ref["losses"][:, :4]).abs() / ref["losses"][:, :4].abs().clamp_min(1e-12)
    dh = (r0["hash"] - ref["hash"]).abs().max() / ref["hash"].abs().max()
    dw = (r0["w1"] - ref["w1"]).abs().max() / ref["w1"].abs().max()
    dp = (r0["poses"] - ref["poses"]).abs().max()
    print(f"sharded vs single: loss rel {float(dl.max()):.2e}  hash {float(dh):.2e}  W1 {float(dw):.2e}  poses {float(dp):.2e}")
    # (two single-GPU runs of the same seed differ by ~1e-6 in the losses and up to ~1e-2 of the table's scale in single hash
    #  entries after these 20 Adam steps: float atomics)
    assert float(dl.max()) < 1e-3 and float(dh) < 3e-2 and float(dw) < 2e-3 and float(dp) < 1e-5


# ---- the volume's cold readers on a sharded volume: the truncated point cloud (V5) and the tracker's nearest-voxel reads (T3)
N_COLD = 9


def _cold_run(make_pipe):
    import random
    import warnings
    import torch
    out = {}
    cfg = _small_cfg()
    pipe = make_pipe(cfg)
    frames = pipe.prefetch(list(range(N_COLD)))
    pipe.start(frames[0])
    for i in range(1, N_COLD):
        pipe.step(i, frames[i])
    out["pc"] = pipe.mv.get_truncated_pc(pc_num=40000, trunc_tsdf=0.5)
    out["pc_small"] = pipe.mv.get_truncated_pc(pc_num=977, trunc_tsdf=0.9)          # many voxels per slot: the highest index must win
    del pipe
    random.seed(0)
    cfg = _small_cfg()
    cfg["synthetic"].update({"tracker": True, "clutter": 32})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pipe = make_pipe(cfg)
    frames = pipe.prefetch(list(range(N_COLD)))
    pipe.start(frames[0])
    for i in range(1, N_COLD):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    out["poses"] = pipe.slam.RO_c2w_data[:N_COLD].detach().cpu().clone()
    out["gt"] = torch.stack([frames[i]["c2w"] for i in range(N_COLD)])
    out["volume_class"] = type(pipe.mv).__name__
    return out


def _worker_cold(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from remixfusion_amd.dist import ShardedPipeline
    out = _cold_run(lambda cfg: ShardedPipeline(cfg, dist, rank, world, n_frames=N_COLD + 4, seed=5))
    torch.save(out, os.path.join(out_dir, f"cold{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_volume_cold_readers_world3(tmp_path):
    """get_truncated_pc (reference model/Volume.py:489-559, :622-653) and the tracker's reads of the volume
    (model/ROtracker.py:144-270) on a volume cut into 3 x-slabs (uneven: 200 planes) against the single-GPU volume."""
    import torch
    import torch.multiprocessing as mp
    from remixfusion_amd.pipeline import MappingPipeline
    world = 3
    mp.spawn(_worker_cold, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, f"cold{r}.pt"), weights_only=False) for r in range(world)]
    ref = _cold_run(lambda cfg: MappingPipeline(cfg, n_frames=N_COLD + 4, seed=5))
    for key in ("pc", "pc_small"):
        assert ref[key].shape[0] > 500 and ref[key].shape[1] == 7
        for r in rs:                                       # the merged cloud: the single-GPU cloud, bit for bit, on every rank
            assert r[key].shape == ref[key].shape and np.array_equal(r[key].view(np.uint32), ref[key].view(np.uint32)), key
    assert all(r["volume_class"] == "sharded_volume" for r in rs) and ref["volume_class"] == "moving_volume"
    for r in rs[1:]:
        assert torch.equal(r["poses"], rs[0]["poses"])     # every rank continues with the same sums: the same poses
    err = (ref["poses"][:, :3, 3] - ref["gt"][:, :3, 3]).norm(dim=1)
    # The slab sums ARE the single-GPU sums (round 5: 30-bit fixed-point terms added as integers, in the threads, between the
    # pixel slabs and over the ranks -- tests/test_tracker_gpu.py holds them to the oracle's exactly), so the search sees the
    # same fitness values, picks the same candidates and leaves the same pose: the trajectory on 3 slabs is the single-GPU
    # trajectory BIT FOR BIT.  (Up to round 4 the sums came from float atomics; the search amplified their last bit into
    # millimetres within a few frames and this test bounded both runs against the truth instead.)
    assert torch.equal(rs[0]["poses"], ref["poses"])
    assert float(err.max()) < 0.08
