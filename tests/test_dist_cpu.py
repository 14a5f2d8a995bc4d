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


# ---- ONE scene over N ranks: host logic (slab cuts, move plan) and the collective helpers on gloo
def test_slab_bounds_and_shift_plan():
    from remixfusion_amd.dist import shift_plan, slab_bounds
    assert slab_bounds(800, 8) == [0, 100, 200, 300, 400, 500, 600, 700, 800]
    c = slab_bounds(7, 3)
    assert c[0] == 0 and c[-1] == 7 and all(b > a for a, b in zip(c[:-1], c[1:]))
    cuts = slab_bounds(800, 4)
    # the volume moves +1 m at 1 cm: new plane x reads old plane x + 100; one plane of slack either side
    need = [(max(0, cuts[r] + 100 - 1), min(800, cuts[r + 1] + 100 + 1)) for r in range(4)]
    plan = shift_plan(cuts, need)
    assert all(s != d for s, d, _, _ in plan)
    for r in range(4):                                   # every needed remote plane is sent exactly once
        got = sorted((p0, p1) for s, d, p0, p1 in plan if d == r)
        a, b = need[r]
        remote = [(max(a, cuts[s]), min(b, cuts[s + 1])) for s in range(4) if s != r and min(b, cuts[s + 1]) > max(a, cuts[s])]
        assert got == sorted(remote)
    assert (1, 0, 200, 301) in plan and (0, 1, 299, 200) not in plan
    assert shift_plan(cuts, [(c0, c1) for c0, c1 in zip(cuts[:-1], cuts[1:])]) == []      # no move: nothing to send


def _worker_collectives(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from remixfusion_amd.dist import all_reduce_sum_, broadcast_
    a = torch.full((5,), float(rank + 1))
    b = torch.arange(6, dtype=torch.float32).view(2, 3).t()          # non-contiguous view
    b = b * (rank + 1)
    d = torch.full((8,), 0.5 * (rank + 1), dtype=torch.float64)
    big = torch.arange(70000, dtype=torch.float32) * (rank + 1)       # above the bucket limit: reduced where it lies
    big_ptr = big.data_ptr()
    all_reduce_sum_(dist, [a, None, big, b, d])
    assert big.data_ptr() == big_ptr
    f = torch.full((4,), float(rank * 10 + 3))
    broadcast_(dist, f, 0)
    torch.save({"a": a, "b": b, "d": d, "f": f, "big": big}, os.path.join(out_dir, f"c{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_helpers_world2(tmp_path):
    world = 2
    mp.spawn(_worker_collectives, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        c = torch.load(os.path.join(tmp_path, f"c{r}.pt"))
        assert torch.equal(c["a"], torch.full((5,), 3.0))
        assert torch.equal(c["b"], torch.arange(6, dtype=torch.float32).view(2, 3).t() * 3)
        assert torch.equal(c["d"], torch.full((8,), 1.5, dtype=torch.float64))
        assert torch.equal(c["f"], torch.full((4,), 3.0))
        assert torch.equal(c["big"], torch.arange(70000, dtype=torch.float32) * 3)
