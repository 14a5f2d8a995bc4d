"""BASELINE configs 4 and 5 as what BASELINE.json says they are: ONE scene over several ranks.

cafeteria (config 4: BS3D sizes -- 1280x720, moving volume 700x700x300 @ 2 cm, hash table 2^21, 63^3 TV lattice) and apartment
(config 5: uHumans2 sizes at 1 cm -- 720x480, 1600x1600x600 = 1.5e9 voxels, S = 117, 10 map iterations, a marching-cubes mesh
per keyframe) run through ``ShardedPipeline`` on 4 and 5 ranks -- that many processes on this one GPU, gloo rendezvous on 127.0.0.1,
device tensors staged through the host (the pool hands out 1-GPU boxes; with backend "nccl" the same code runs one rank per
GPU over RCCL) -- for two mapper steps and a volume move across the slab cuts, then once more in a single process:

  * moving volume: every rank's x-slab bit-identical to the same planes of the single-process volume (exact digests),
  * the hash table partitioned by level (mp_slam/sharded.py: per-point rows exchanged, never the table); after
    ``Mapper.sync_field()`` every rank holds the same table, decoder and global volume, bit for bit,
  * losses within the noise of float atomics of the single-process run,
  * config 5: the per-keyframe mesh hook of the reference's loop (mp_slam/mapper.py:908-918) produced a mesh on rank 0.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_FRAMES = 12
WORLDS = {"cafeteria": 4, "apartment": 5}       # BASELINE asks 4 and 8; a GPU box admits at most 6 processes on its card (gpurun's
FAR_POSE_DX = 1.4                               # process guard) and the test process is one of them: config 5 runs on 5 (uneven level ranges)


def _cfg(name):
    from remixfusion_amd.config import synthetic_config
    cfg = synthetic_config(name)
    cfg["mapping"].update({"first_iters": 8})
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.02})
    cfg["pipeline"] = {"mv_stream": False}
    return cfg


def _digest(t):
    """exact fingerprint of a float tensor: two integer sums over its bit patterns (order-independent, overflow wraps)"""
    import torch
    a = t.contiguous().reshape(-1).view(torch.int32).to(torch.int64)
    idx = torch.arange(a.numel(), device=a.device, dtype=torch.int64) % 65521 + 1
    return int(a.sum().item()), int((a * idx).sum().item())


def _run(pipe, frames, out, tmp):
    import torch
    pipe.config["data"]["output"] = tmp          # the in-loop meshes go under the test's directory
    if pipe.slam is not None:
        pipe.slam.config["data"]["output"] = tmp
    pipe.start(frames[0])
    for i in range(1, N_FRAMES):
        pipe.step(i, frames[i])
    d = pipe.mapper._direct_iterations()
    batch = pipe.dataset[N_FRAMES - 1]
    rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], -1).reshape(-1, 7).to(pipe.device)
    poses = pipe.slam.est_c2w_data[0:N_FRAMES:pipe.config["mapping"]["keyframe_every"]].clone()
    out["losses"] = d.map_gradients(rays, poses).clone().cpu()
    out["mode"] = type(d).__name__
    out["recv_bytes"] = getattr(d, "last_exchange", {}).get("recv_bytes")
    pipe.mapper.sync_field()                     # collective on a sharded scene: every rank's copy of the table whole again
    out["mapping_idx"] = int(pipe.slam.mapping_idx[0])
    out["hash"] = _digest(pipe.model.embed_res_fn.params.detach())
    out["w1"] = _digest(pipe.model.decoder_res.fused_weights()[0].detach())
    out["gbv"] = _digest(pipe.model.GBV.params.detach())
    out["hash_max"] = float(pipe.model.embed_res_fn.params.detach().abs().max())
    mesh = getattr(pipe.mapper, "last_mesh", None)
    out["mesh_faces"] = None if mesh is None else int(mesh["faces"].shape[0])
    far = frames[N_FRAMES - 1]["c2w"].clone().numpy().astype(np.float64)
    far[0, 3] += FAR_POSE_DX
    moved, _ = pipe.mv.check_move_volume_new(N_FRAMES, far, pipe.traj, version=pipe.config["volume"]["version"])
    out["moved"] = bool(moved)
    out["bnds"] = np.array(pipe.mv.vol_bnds)
    torch.cuda.synchronize()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, name):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from remixfusion_amd.dist import ShardedPipeline
    pipe = ShardedPipeline(_cfg(name), dist, rank, world, n_frames=N_FRAMES + 4, seed=5)
    frames = pipe.prefetch(list(range(N_FRAMES)))
    out = {}
    _run(pipe, frames, out, os.path.join(out_dir, f"rank{rank}"))
    n = pipe.mv._n()
    out["slab"] = pipe.mv._slab()
    out["mv"] = [_digest(t[:n]) for t in pipe.mv._vols()]
    out["mv_w_pos"] = float((pipe.mv.weight_vol_gpu[:n] > 0).float().mean())
    # the halo read: trilinear samples at points spread over the whole volume, incl. cells that straddle the slab cuts
    g = torch.Generator().manual_seed(3)
    b = torch.from_numpy(np.array(pipe.mv.vol_bnds)).float()
    pts = b[:, 0] + (b[:, 1] - b[:, 0]) * torch.rand((4000, 3), generator=g)
    cuts = pipe.mv._cuts()
    for k, c in enumerate(cuts[1:-1]):                       # points whose lower corner is the last plane of a slab
        pts[k * 50:(k + 1) * 50, 0] = float(pipe.mv.vol_origin[0]) + (c - 1 + 0.37) * pipe.mv.voxel_size
    res, valid = pipe.mv.tri_interpolate(pts.numpy())
    out["tri"] = res
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("name", ["cafeteria", "apartment"])
def test_config_runs_sharded_over_several_ranks_and_equals_the_single_process_run(name, tmp_path):
    import torch
    import torch.multiprocessing as mp
    from remixfusion_amd.pipeline import MappingPipeline
    WORLD = WORLDS[name]
    mp.spawn(_worker, args=(WORLD, _free_port(), str(tmp_path), name), nprocs=WORLD, join=True)
    rs = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(WORLD)]
    # ---- the single-process run of the same stream
    pipe = MappingPipeline(_cfg(name), n_frames=N_FRAMES + 4, seed=5)
    frames = pipe.prefetch(list(range(N_FRAMES)))
    ref = {}
    _run(pipe, frames, ref, os.path.join(str(tmp_path), "single"))
    dims = [int(v) for v in pipe.mv.vol_dim]
    assert dims == ([700, 700, 300] if name == "cafeteria" else [1600, 1600, 600])
    assert ref["mapping_idx"] == 10 and all(r["mapping_idx"] == 10 for r in rs)            # two mapper steps
    assert ref["moved"] and all(r["moved"] for r in rs) and all(np.array_equal(ref["bnds"], r["bnds"]) for r in rs)
    plane = dims[1] * dims[2]
    for r in rs:                                                                           # moving volume, slab by slab
        x0, x1 = r["slab"]
        for k, (t, nm) in enumerate(zip(pipe.mv._vols(), ("tsdf", "weight", "colour"))):
            assert _digest(t[x0 * plane:x1 * plane]) == tuple(r["mv"][k]), (nm, x0, x1)
    assert sum(r["mv_w_pos"] for r in rs) > 0
    assert all(r["mode"] == "LevelShardedIterations" for r in rs)                          # the table partitioned by level
    for key in ("hash", "w1", "gbv"):                                                      # after sync_field: copies bit-identical
        assert all(tuple(r[key]) == tuple(rs[0][key]) for r in rs), key
    # what an iteration moves between the ranks: per-point rows, not the table (161 / 166 MB of gradient at T = 2^21)
    assert all(r["recv_bytes"] is not None and r["recv_bytes"] < 24e6 for r in rs), [r["recv_bytes"] for r in rs]
    assert all(torch.equal(r["losses"], rs[0]["losses"]) for r in rs)
    assert tuple(rs[0]["gbv"]) == ref["gbv"]                                               # deterministic kernel, same keyframes
    dl = (rs[0]["losses"][:4] - ref["losses"][:4]).abs() / ref["losses"][:4].abs().clamp_min(1e-12)
    # the yardstick: the single-process run against ITSELF (a second run of the same seed).  The hash gradient is built with
    # float atomics, so after these 20 Adam steps two single-GPU runs already differ in the last digits of the losses
    pipe2 = MappingPipeline(_cfg(name), n_frames=N_FRAMES + 4, seed=5)
    ref2 = {}
    _run(pipe2, pipe2.prefetch(list(range(N_FRAMES))), ref2, os.path.join(str(tmp_path), "single2"))
    del pipe2
    noise = (ref2["losses"][:4] - ref["losses"][:4]).abs() / ref["losses"][:4].abs().clamp_min(1e-12)
    print(f"{name}: sharded vs single loss rel {float(dl.max()):.2e} (single vs single: {float(noise.max()):.2e}); "
          f"mesh faces {rs[0]['mesh_faces']} / {ref['mesh_faces']}")
    assert float(dl.max()) < max(4.0 * float(noise.max()), 2e-5)
    # halo read: every rank returns the single-process records, bit for bit
    g = torch.Generator().manual_seed(3)
    b = torch.from_numpy(np.array(pipe.mv.vol_bnds)).float()
    pts = b[:, 0] + (b[:, 1] - b[:, 0]) * torch.rand((4000, 3), generator=g)
    from remixfusion_amd.dist import slab_bounds
    cuts = slab_bounds(dims[0], WORLD)
    for k, c in enumerate(cuts[1:-1]):
        pts[k * 50:(k + 1) * 50, 0] = float(pipe.mv.vol_origin[0]) + (c - 1 + 0.37) * pipe.mv.voxel_size
    res, _ = pipe.mv.tri_interpolate(pts.numpy())
    for r in rs:
        assert np.array_equal(res.view(np.uint32), r["tri"].view(np.uint32))
    assert float((res[:, 0] != 1.0).mean()) > 0.001
    if name == "apartment":                                                                # config 5: a mesh per keyframe
        assert rs[0]["mesh_faces"] is not None and rs[0]["mesh_faces"] > 0 and ref["mesh_faces"] is not None
        assert all(r["mesh_faces"] is None for r in rs[1:])                                # rank 0 writes
        assert os.path.exists(os.path.join(str(tmp_path), "rank0", "apartment", "mesh_track10.ply"))
    else:
        assert ref["mesh_faces"] is None
