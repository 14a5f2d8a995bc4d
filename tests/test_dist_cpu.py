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


def _worker_async_collectives(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from remixfusion_amd.dist import all_reduce_sum_start, all_to_all_rows_start
    # the two-half forms of round 6: issue, do something that does not read the buffers, then join
    sums = torch.full((8,), 0.25 * (rank + 1), dtype=torch.float64)
    dw = torch.arange(5312, dtype=torch.float32) * (rank + 1)
    tv = torch.full((1,), float(rank + 1), dtype=torch.float64)
    finish = all_reduce_sum_start(dist, [sums, None, dw, tv])         # two dtype buckets: (sums, tv) concatenated, dw alone
    inp = torch.arange(10, dtype=torch.float32) + 100 * rank          # 3 + 7 (rank 0) / 6 + 4 (rank 1) elements to ranks 0 / 1
    in_splits = [3, 7] if rank == 0 else [6, 4]
    out_splits = [3, 6] if rank == 0 else [7, 4]
    out = torch.full((16,), -1.0)
    done = all_to_all_rows_start(dist, out, out_splits, inp, in_splits)
    unrelated = torch.ones(4).sum()                                    # (what the caller launches in between)
    done()
    finish()
    torch.save({"sums": sums, "dw": dw, "tv": tv, "out": out, "unrelated": unrelated}, os.path.join(out_dir, f"a{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_half_collectives_world2(tmp_path):
    """dist.all_reduce_sum_start / all_to_all_rows_start (mp_slam/sharded.py issues its exchanges with them): same results as the
    blocking helpers, the small tensors of several dtypes bucketed and copied back after the wait"""
    world = 2
    mp.spawn(_worker_async_collectives, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a0, a1 = (torch.load(os.path.join(tmp_path, f"a{r}.pt")) for r in range(2))
    for a in (a0, a1):
        assert torch.equal(a["sums"], torch.full((8,), 0.75, dtype=torch.float64))
        assert torch.equal(a["dw"], torch.arange(5312, dtype=torch.float32) * 3)
        assert torch.equal(a["tv"], torch.full((1,), 3.0, dtype=torch.float64))
    r0 = torch.arange(10, dtype=torch.float32)
    r1 = r0 + 100
    assert torch.equal(a0["out"][:9], torch.cat([r0[:3], r1[:6]])) and torch.equal(a0["out"][9:], torch.full((7,), -1.0))
    assert torch.equal(a1["out"][:11], torch.cat([r0[3:], r1[6:]])) and torch.equal(a1["out"][11:], torch.full((5,), -1.0))


# ---- the hash table partitioned by level (mp_slam/sharded.py::LevelShardedIterations): host logic and the exchange protocol
def _hash_desc(name):
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.encodings import make_grid_desc
    cfg = synthetic_config(name)
    b = np.array(cfg["mapping"]["bound"], dtype=np.float64)
    R = int((b[:, 1] - b[:, 0]).max() / cfg["grid"]["voxel_sdf"])
    pls = np.exp2(np.log2(R / 16) / 15)
    return make_grid_desc(16, 2, cfg["grid"]["hash_size"], 16, pls, True)[0]


def test_level_partition_is_contiguous_complete_and_optimal():
    import itertools
    from remixfusion_amd.dist import level_costs, level_partition, ray_partition
    for name in ("office0", "scene0000", "cafeteria", "apartment"):
        desc = _hash_desc(name)
        cost = level_costs(desc)
        assert len(cost) == 16 and (name == "office0") == all(c == 1.0 for c in cost)      # T = 2^16: no binned level
        for world in range(1, 17):
            cuts = level_partition(desc, world)
            assert cuts[0] == 0 and cuts[-1] == 16 and len(cuts) == world + 1
            assert all(b > a for a, b in zip(cuts[:-1], cuts[1:]))                           # every rank keeps at least one level
            got = max(sum(cost[a:b]) for a, b in zip(cuts[:-1], cuts[1:]))
            if world <= 4:                                                                   # brute force: the smallest possible maximum
                best = min(max(sum(cost[a:b]) for a, b in zip((0,) + c, c + (16,))) for c in itertools.combinations(range(1, 16), world - 1))
                assert abs(got - best) < 1e-9, (name, world, cuts)
    with __import__("pytest").raises(ValueError):
        level_partition(_hash_desc("office0"), 17)
    assert ray_partition(2304, 4) == [0, 576, 1152, 1728, 2304]
    r = ray_partition(2148, 8)
    assert r[0] == 0 and r[-1] == 2148 and max(b - a for a, b in zip(r[:-1], r[1:])) - min(b - a for a, b in zip(r[:-1], r[1:])) <= 1


def test_field_exchange_model_prefers_level_rows_to_the_dense_gradient():
    """bytes a rank receives per map iteration at cafeteria sizes (T = 2^21, 2 148 rays x 59 + 63^3 lattice)"""
    from remixfusion_amd.dist import field_exchange_model
    desc = _hash_desc("cafeteria")
    mdl = field_exchange_model(desc, 2148 * 59, 63 ** 3, 4)
    assert 160e6 < mdl["table_bytes"] < 170e6
    assert mdl["replicas"]["recv_bytes"] > 240e6                     # 2 (N-1)/N of 166 MB
    assert mdl["levels"]["recv_bytes"] < 7e6                         # 2 x 3/4 x (1.27e5 / 4) x 128 B
    assert mdl["points"]["recv_bytes"] < mdl["replicas"]["recv_bytes"] and mdl["points"]["scatter_share"] == 1.0
    assert mdl["levels"]["scatter_share"] < 0.32 < 0.7 < mdl["replicas"]["scatter_share"]      # the lattice alone is 2/3 of the replicas' rank 0
    assert mdl["levels"]["recv_bytes"] < mdl["points"]["recv_bytes"]


def test_auto_mode_follows_the_time_model():
    from remixfusion_amd.dist import choose_field_mode
    assert choose_field_mode(_hash_desc("office0"), 2304 * 59, 31 ** 3, 2)["mode"] == "replicas"      # 6.6 MB table: all-reduce it
    for name, S, world in (("scene0000", 117, 2), ("cafeteria", 59, 2), ("cafeteria", 59, 4), ("apartment", 117, 8)):
        c = choose_field_mode(_hash_desc(name), 2304 * S, 63 ** 3, world)
        assert c["mode"] == "levels" and c["estimated_seconds"]["levels"] < (0.6 if name == "scene0000" else 0.2) * c["estimated_seconds"]["replicas"], (name, world)
        assert c["estimated_seconds"]["levels"] < c["estimated_seconds"]["points"]
        # ... and the whole map iteration on N GPUs is estimated below the one-GPU iteration, which the replicated table is not
        # (round 6: the one-GPU scatter got 20 % faster, which leaves N = 2 less to divide: scene0000's modelled iteration on two
        #  GPUs is 0.97 of the one-GPU one, cafeteria's 0.9)
        assert c["iteration_seconds"]["levels"] < (1.0 if world == 2 else 0.7) * c["iteration_seconds_one_gpu"], (name, world)
        assert c["iteration_seconds_one_gpu"] < c["iteration_seconds"]["replicas"], (name, world)
    assert choose_field_mode(_hash_desc("cafeteria"), 2304 * 59, 63 ** 3, 1)["mode"] == "replicas"     # one rank: nothing to partition


def _worker_level_exchange(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from remixfusion_amd.dist import all_to_all_rows_, level_exchange_splits, level_partition
    n, S = 37, 5                                   # rays do not divide by the world: uneven shares
    cuts = level_partition(_hash_desc("scene0000"), world)
    sp = level_exchange_splits(n, S, cuts, rank)
    rs, m, k = sp["rays"], sp["m"], cuts[rank + 1] - cuts[rank]
    pt = torch.arange(n * S, dtype=torch.float32)
    # ---- features out: feat_send[p, 2 j + c] of the own level cuts[rank] + j encodes (point, level, component)
    lv = torch.arange(cuts[rank], cuts[rank + 1], dtype=torch.float32)
    feat_send = (pt[:, None, None] * 1000 + lv[None, :, None] * 10 + torch.arange(2.0)[None, None, :]).reshape(n * S, 2 * k)
    feat_recv = torch.full((m * S * 32,), -1.0)
    all_to_all_rows_(dist, feat_recv, sp["feat"][1], feat_send.reshape(-1), sp["feat"][0])
    ok = True
    for q in range(world):                         # the addressing rfx_ba_shard_render uses (csrc/rfx_ba.hip: shard_rows)
        a, e = cuts[q], cuts[q + 1]
        blk = feat_recv[m * S * 2 * a: m * S * 2 * e].view(m * S, 2 * (e - a))
        for l in range(a, e):
            for c in range(2):
                want = (pt[rs[rank] * S: rs[rank + 1] * S] * 1000 + l * 10 + c)
                ok &= bool(torch.equal(blk[:, 2 * (l - a) + c], want))
    # ---- gradients back: demb_send block q [m S, 2 k_q] encodes (own point, level of q, component)
    own_pt = pt[rs[rank] * S: rs[rank + 1] * S]
    blocks = []
    for q in range(world):
        lq = torch.arange(cuts[q], cuts[q + 1], dtype=torch.float32)
        blocks.append((own_pt[:, None, None] * 1000 + lq[None, :, None] * 10 + torch.arange(2.0)[None, None, :] + 0.5).reshape(-1))
    demb_recv = torch.full((n * S * 2 * k,), -1.0)
    all_to_all_rows_(dist, demb_recv, sp["demb"][1], torch.cat(blocks), sp["demb"][0])
    ok &= bool(torch.equal(demb_recv.view(n * S, 2 * k), feat_send + 0.5))          # all points, own levels, point order
    # ---- pose phase: every rank's partial d loss / d x01 of the own points, in rank order
    dx_send = (pt[:, None] * 10 + torch.arange(3.0)[None, :] + 100000.0 * rank).reshape(-1)
    dx_recv = torch.full((world * m * S * 3,), -1.0)
    all_to_all_rows_(dist, dx_recv, sp["dx"][1], dx_send, sp["dx"][0])
    for q in range(world):
        want = (own_pt[:, None] * 10 + torch.arange(3.0)[None, :] + 100000.0 * q).reshape(-1)
        ok &= bool(torch.equal(dx_recv[q * m * S * 3:(q + 1) * m * S * 3], want))
    torch.save({"ok": ok, "m": m, "cuts": cuts}, os.path.join(out_dir, f"x{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@__import__("pytest").mark.parametrize("world", [3, 8])
def test_level_exchange_protocol(tmp_path, world):
    """the three all-to-alls of a level-partitioned iteration on 3 and on 8 gloo ranks (BASELINE config 5's world; uneven ray
    shares and level ranges): every (point, level) lands where the phases of csrc/rfx_ba.hip read it"""
    mp.spawn(_worker_level_exchange, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"x{r}.pt")) for r in range(world)]
    assert all(r["ok"] for r in res)
    assert sum(r["m"] for r in res) == 37 and res[0]["cuts"] == res[world - 1]["cuts"]
