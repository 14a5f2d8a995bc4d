"""world_size-2 gloo test (CPU) of the multi-GPU path: scene partition geometry and the ghost-plane
exchange of the global volume between neighbouring partitions."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, R, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from remixfusion_amd.dist import ScenePartition
    from remixfusion_amd.config import synthetic_config
    part = ScenePartition(synthetic_config("office0"), rank, world, dist)
    g = torch.Generator().manual_seed(100 + rank)
    gbv = torch.rand(R ** 3 * 4, generator=g)
    gbw = torch.rand(R ** 3, generator=g)
    before_v, before_w = gbv.clone(), gbw.clone()
    part.exchange_halo(gbv, gbw, R)
    torch.save({"v": gbv, "w": gbw, "bv": before_v, "bw": before_w, "bound": part.config["mapping"]["bound"]},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_geometry_overlaps_by_two_cells():
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.dist import partition_config
    cfg = synthetic_config("office0")
    R = cfg["globalV"]["base_resolution"]
    parts = [partition_config(cfg, r, 4) for r in range(4)]
    ext = cfg["mapping"]["bound"][0][1] - cfg["mapping"]["bound"][0][0]
    cell = ext / R
    for a, b in zip(parts[:-1], parts[1:]):
        ax, bx = a["mapping"]["bound"][0], b["mapping"]["bound"][0]
        assert abs((ax[1] - bx[0]) - 2 * cell) < 1e-9            # two-cell overlap
        # vertex R-2 of the left slab sits on vertex 0 of the right slab (vertices at i/R of the extent)
        assert abs((ax[0] + (R - 2) * cell) - bx[0]) < 1e-9
        assert a["mapping"]["bound"][1:] == b["mapping"]["bound"][1:]
    assert parts[0]["mapping"]["bound"] == cfg["mapping"]["bound"]
    assert len({p["synthetic"]["seed"] for p in parts}) == 4


def test_ghost_plane_exchange_world2(tmp_path):
    R, world = 6, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, R, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(2))
    v0, v1 = r0["v"].view(R, R, R, 4), r1["v"].view(R, R, R, 4)
    b0, b1 = r0["bv"].view(R, R, R, 4), r1["bv"].view(R, R, R, 4)
    w0, w1 = r0["w"].view(R, R, R), r1["w"].view(R, R, R)
    # rank 0's last plane <- rank 1's plane 1 ; rank 1's plane 0 <- rank 0's plane R-2
    assert torch.equal(v0[:, :, R - 1], b1[:, :, 1]) and torch.equal(v1[:, :, 0], b0[:, :, R - 2])
    assert torch.equal(w0[:, :, R - 1], r1["bw"].view(R, R, R)[:, :, 1]) and torch.equal(w1[:, :, 0], r0["bw"].view(R, R, R)[:, :, R - 2])
    # owned planes are untouched; outer boundaries (no neighbour) too
    assert torch.equal(v0[:, :, :R - 1], b0[:, :, :R - 1]) and torch.equal(v1[:, :, 1:], b1[:, :, 1:])
    # after the exchange both ranks agree on the two overlap planes
    assert torch.equal(v0[:, :, R - 2], v1[:, :, 0]) and torch.equal(v0[:, :, R - 1], v1[:, :, 1])
