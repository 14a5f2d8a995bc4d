"""Multi-GPU: spatial partition of the scene across the GPUs of one node (one process per GPU,
torch.distributed over RCCL/xGMI).

The reference is single-GPU (no torch.distributed / NCCL anywhere), so this is new functionality
with no reference semantics to match; the 1-GPU result of each partition is unchanged.

Design (BASELINE configs 4/5): the scene bound is cut into ``world`` slabs along x.
Rank r owns slab r: its own moving TSDF volume (follows the camera that maps that region), its own
global explicit volume GBV/GBW and residual field over the slab's bound.  The integrate kernels
need no exchange.  Neighbouring slabs overlap by two GBV cells; after every keyframe integration
the owned boundary planes are exchanged point-to-point so both sides hold identical values in the
overlap (ghost planes):

    my plane R-2  --->  right neighbour's plane 0        my plane 1  --->  left neighbour's plane R-1

That is 2 x (R*R*(4+1) floats) = 1.6 MB per neighbour per keyframe -- far below one xGMI link
(~153 GB/s), so plain send/recv (no ring collective) on a side stream is the right primitive.
"""
from __future__ import annotations

import copy
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .model.Volume import moving_volume


def partition_config(cfg: Dict, rank: int, world: int) -> Dict:
    """config of slab ``rank``: bound / room / marching-cubes bound shifted along x so that adjacent
    slabs overlap by exactly two GBV cells; the synthetic stream of each slab gets its own seed."""
    c = copy.deepcopy(cfg)
    R = c["globalV"]["base_resolution"]
    (x0, x1) = c["mapping"]["bound"][0]
    ext = x1 - x0
    cell = ext / R
    shift = rank * (ext - 2 * cell)
    for key in ("bound", "marching_cubes_bound"):
        c["mapping"][key][0] = [c["mapping"][key][0][0] + shift, c["mapping"][key][0][1] + shift]
    room = c["synthetic"]["room"]
    room[0] = [room[0][0] + shift, room[0][1] + shift]
    c["synthetic"]["seed"] = int(c["synthetic"]["seed"]) + 1000 * rank
    c["synthetic"]["partition"] = {"rank": rank, "world": world, "shift_x": shift, "overlap_cells": 2}
    return c


def boundary_planes(params: torch.Tensor, R: int, feat: int):
    """views of the x-planes of a tcnn dense grid (x fastest, features interleaved): [z, y, x, f]."""
    return params.view(R, R, R, feat)


class ScenePartition:
    def __init__(self, cfg: Dict, rank: int, world: int, dist=None):
        self.rank, self.world, self.dist = rank, world, dist
        self.config = partition_config(cfg, rank, world)
        self.left: Optional[int] = rank - 1 if rank > 0 else None
        self.right: Optional[int] = rank + 1 if rank < world - 1 else None
        self._bufs: Dict[str, torch.Tensor] = {}
        self._stream = None

    def make_volume(self, config, traj, pose0, device):
        from .model.Volume import moving_volume
        return moving_volume(config, traj, pose0, device=device)

    def exchange_halo(self, gbv: torch.Tensor, gbw: torch.Tensor, R: int) -> None:
        """make the two overlap planes on each side consistent with the owning neighbour."""
        exchange_planes(self.dist, self.rank, self.left, self.right, gbv, gbw, R)


def exchange_planes(dist, rank: int, left: Optional[int], right: Optional[int], gbv: torch.Tensor, gbw: torch.Tensor,
                    R: int) -> None:
    """point-to-point ghost-plane exchange (works on any backend: nccl=RCCL on GPUs, gloo on CPU)."""
    if dist is None or (left is None and right is None):
        return
    v, w = boundary_planes(gbv, R, 4), boundary_planes(gbw, R, 1)

    def pack(ix: int) -> torch.Tensor:
        return torch.cat([v[:, :, ix, :].reshape(-1), w[:, :, ix, :].reshape(-1)]).contiguous()

    def unpack(buf: torch.Tensor, ix: int) -> None:
        n4 = R * R * 4
        with torch.no_grad():
            v[:, :, ix, :] = buf[:n4].view(R, R, 4)
            w[:, :, ix, :] = buf[n4:].view(R, R, 1)

    # gloo (CPU rehearsal of the multi-GPU path) cannot move device tensors: stage through the host
    host_staged = gbv.is_cuda and dist.get_backend() == "gloo"
    if host_staged:
        _pack = pack

        def pack(ix: int) -> torch.Tensor:       # noqa: F811
            return _pack(ix).cpu()
    ops, recvs = [], []
    if right is not None:
        send_r = pack(R - 2)
        recv_r = torch.empty_like(send_r)
        ops += [dist.P2POp(dist.isend, send_r, right), dist.P2POp(dist.irecv, recv_r, right)]
        recvs.append((recv_r, R - 1))
    if left is not None:
        send_l = pack(1)
        recv_l = torch.empty_like(send_l)
        ops += [dist.P2POp(dist.isend, send_l, left), dist.P2POp(dist.irecv, recv_l, left)]
        recvs.append((recv_l, 0))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for buf, ix in recvs:
        unpack(buf.to(gbv.device) if host_staged else buf, ix)


def make_shard(cfg: Dict, rank: int, world: int, dist=None) -> ScenePartition:
    return ScenePartition(cfg, rank, world, dist)


# ============================================================================================================
# ONE scene over N GPUs (BASELINE configs 4 / 5)
# ============================================================================================================
# One camera stream.  The moving TSDF volume is cut into contiguous x-slabs (x is the slowest axis of the reference
# layout, model/Volume.py:224-226: idx = z + y*Dz + x*Dy*Dz, so a slab is one contiguous range of the three arrays);
# every rank integrates the SAME frame into its slab (rfx_tsdf_integrate_slab: global indices and coordinates, hence
# bit-identical to the unsharded volume) -- V1 needs no voxel exchange.  When the volume follows the camera (V2,
# model/Volume.py:796-855) a shift along x crosses slab boundaries: each rank fetches the old x-planes its new slab
# reads from the ranks that own them (point-to-point, contiguous plane ranges) and gathers locally
# (rfx_tsdf_shift_slab).  The residual field and the global explicit volume are small (6.6 MB .. 166 MB, 160 MB) and
# replicated: each rank renders a strided share of an iteration's ray batch, the loss sums are all-reduced before the
# loss weights are formed (so the loss is that of the whole batch), and the gradients (hash table, decoder, poses) are
# all-reduced before the identical Adam step on every rank.  Collectives: broadcast (frame, 8*H*W B), send/recv
# (volume move), all-reduce (56 B of loss sums; gradients).  torch.distributed: backend "nccl" = RCCL over xGMI on the
# GPUs, "gloo" for the CPU / single-GPU rehearsal (device tensors are staged through the host there).

def slab_bounds(dx: int, world: int) -> List[int]:
    """x-plane cuts [c_0 = 0, ..., c_world = dx]: slab r is [c_r, c_{r+1}); sizes differ by at most one plane."""
    return [(dx * r) // world for r in range(world + 1)]


# RFX_DIST_FORCE_COLLECTIVES=1: issue every collective even in a world of one rank.  A 1-GPU box cannot hold two RCCL ranks
# (one communicator rank per device), but a one-rank RCCL communicator executes the same calls on the same device tensors
# (views, split sizes, dtypes, stream ordering): tests/test_rccl_one_rank_gpu.py and `RFX_FORCE_SHARDED=1 bench.py` use it.
_FORCE_COLLECTIVES = os.environ.get("RFX_DIST_FORCE_COLLECTIVES") == "1"


def _alone(dist) -> bool:
    """no collective needed: no process group, or a world of one (unless forced, above)"""
    return dist is None or (dist.get_world_size() == 1 and not _FORCE_COLLECTIVES)


def _host_staged(dist, t: torch.Tensor) -> bool:
    return dist is not None and t.is_cuda and dist.get_backend() == "gloo"


def broadcast_(dist, t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """in-place broadcast of a device tensor (gloo: via the host)."""
    if _alone(dist):
        return t
    if _host_staged(dist, t):
        h = t.cpu()
        dist.broadcast(h, src)
        if dist.get_rank() != src:
            t.copy_(h)
    else:
        dist.broadcast(t, src)
    return t


_BUCKET_MAX_ELEMS = 1 << 16      # tensors up to this size share one flattened bucket per dtype; larger ones go in place


def _all_reduce_flat(dist, flat) -> None:
    if _host_staged(dist, flat):
        h = flat.cpu()
        dist.all_reduce(h)
        flat.copy_(h)
    else:
        dist.all_reduce(flat)


def all_reduce_sum_(dist, tensors) -> None:
    """in-place sum over ranks of a list of tensors.  The big ones (the hash gradient: 6.6 ... 166 MB) are all-reduced where
    they lie -- no concatenation, no copy back; only the small ones (decoder weight gradients, loss sums, pose gradients)
    share one flattened bucket per dtype, so that they cost one collective instead of one each."""
    if _alone(dist):
        return
    small = {}
    for t in tensors:
        if t is None:
            continue
        if t.is_contiguous() and t.numel() > _BUCKET_MAX_ELEMS:
            _all_reduce_flat(dist, t.view(-1))
        else:
            small.setdefault(t.dtype, []).append(t)
    for ts in small.values():
        if len(ts) == 1 and ts[0].is_contiguous():
            _all_reduce_flat(dist, ts[0].view(-1))
            continue
        flat = torch.cat([t.reshape(-1) for t in ts])
        _all_reduce_flat(dist, flat)
        o = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[o:o + n].view_as(t))
            o += n


def all_to_all_rows_(dist, out: torch.Tensor, out_splits: List[int], inp: torch.Tensor, in_splits: List[int]) -> None:
    """all-to-all of flat fp32 buffers with per-rank element counts: inp is cut into in_splits (piece q goes to rank q), out
    receives the pieces of ranks 0..world-1 in turn (out_splits).  nccl (RCCL): on the device; gloo: through the host."""
    all_to_all_rows_start(dist, out, out_splits, inp, in_splits)()


def _done():
    return None


def all_to_all_rows_start(dist, out: torch.Tensor, out_splits: List[int], inp: torch.Tensor, in_splits: List[int]):
    """all_to_all_rows_ in two halves: the exchange is ISSUED here and the returned callable makes the current stream wait for
    it.  Under nccl (RCCL) the collective runs on the process group's own stream (which first waits for everything already
    queued on the current stream), so kernels launched between the two halves run WHILE the rows travel -- whatever does not
    read `out` (round 6: the TV lattice's lookups beside the feature rows).  gloo / a world of one: done on return."""
    if _alone(dist):
        out[:sum(out_splits)].copy_(inp[:sum(in_splits)])
        return _done
    o, i = out[:sum(out_splits)], inp[:sum(in_splits)]
    if _host_staged(dist, inp):
        ho = torch.empty(o.shape, dtype=o.dtype)
        dist.all_to_all_single(ho, i.cpu(), output_split_sizes=list(out_splits), input_split_sizes=list(in_splits))
        o.copy_(ho)
        return _done
    work = dist.all_to_all_single(o, i, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), async_op=True)
    return work.wait


def all_reduce_sum_start(dist, tensors):
    """all_reduce_sum_ in two halves (see all_to_all_rows_start): the small tensors' bucket is all-reduced on the process
    group's stream; the returned callable waits for it and copies the sums back.  What is launched in between must not read
    the tensors (round 6: the table scatter beside the all-reduce of the decoder gradients and the loss sums, which only
    the optimizer step / the loss report need)."""
    if _alone(dist):
        return _done
    ts = [t for t in tensors if t is not None]
    if not ts:
        return _done
    if any(_host_staged(dist, t) for t in ts) or any(t.numel() > _BUCKET_MAX_ELEMS for t in ts):
        all_reduce_sum_(dist, ts)
        return _done
    by_dtype = {}
    for t in ts:
        by_dtype.setdefault(t.dtype, []).append(t)
    pending = []
    for group in by_dtype.values():
        flat = group[0].view(-1) if len(group) == 1 and group[0].is_contiguous() else torch.cat([t.reshape(-1) for t in group])
        pending.append((dist.all_reduce(flat, async_op=True), flat, group))

    def finish():
        for work, flat, group in pending:
            work.wait()
            if len(group) == 1 and flat.data_ptr() == group[0].data_ptr():
                continue
            o = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[o:o + n].view_as(t))
                o += n
    return finish


# ---- the residual field over N GPUs: the hash table partitioned by LEVEL (mp_slam/sharded.py) -----------------------------
_BIN_SEG, _BIN_MIN_SEGMENTS = 8192, 12       # csrc/rfx_field.hip: levels of >= 12 segments of 8 192 entries take the binned scatter
# What keeping a level costs a rank per map iteration, relative to a small (LDS-swept) level, measured with the ranks of worlds
# of 2, 4 and 8 played on one GPU at cafeteria and apartment sizes (tools/shard_phase_times.py, profiles/r4_shard_phases.txt):
# a small level ~30 us (the 250 k-point TV lattice collides on its few cells), a binned DENSE level ~60 us (its segments are
# slabs of space, a scene fills a few of them), a binned HASHED level of 2^21 entries ~78 us: ~37 us of scatter + its Adam
# step, 7 arrays streamed over its 2.1e6 entries.  The costs are not exactly additive (grouped launches, bandwidth shared
# between phases); these three numbers reproduce the best partitions found by measurement at N = 2, 4 and 8.
_SMALL_LEVEL_COST, _BINNED_DENSE_COST, _BINNED_HASHED_COST = 1.0, 2.0, 2.6
_SWEEP_UNITS_BINNED = 3.1                    # the TIME model's units (choose_field_mode): a binned level's scatter against a small
                                             # level's, single GPU (round 6, profiles/r6_scatter_16_levels.txt: cafeteria's 494 us =
                                             # 4 small levels at 12 us + 12 binned ones at 37 us; round 3: 62 us against 17 us = 3.6)


def _binned(desc, l: int) -> bool:
    return -(-int(desc.size[l]) // _BIN_SEG) >= _BIN_MIN_SEGMENTS


def level_costs(desc) -> List[float]:
    """relative cost of keeping one hash level (lookups + gradient scatter + TV term + Adam step): every level sees every
    point, so the cost is per level; large levels take the binned scatter and a bandwidth-bound Adam step."""
    return [(_BINNED_HASHED_COST if int(desc.hashed[l]) else _BINNED_DENSE_COST) if _binned(desc, l) else _SMALL_LEVEL_COST
            for l in range(int(desc.n_levels))]


def _sweep_units(desc) -> float:
    return sum(_SWEEP_UNITS_BINNED if _binned(desc, l) else 1.0 for l in range(int(desc.n_levels)))


def level_partition(desc, world: int) -> List[int]:
    """cuts [0 = c_0 < c_1 < ... < c_world = n_levels]: rank q keeps the CONTIGUOUS levels [c_q, c_{q+1}) (one contiguous
    range of the table, of its gradient and of the Adam state), chosen to minimise the largest rank's cost."""
    L = int(desc.n_levels)
    if not 1 <= world <= L:
        raise ValueError(f"a {L}-level table can be partitioned over at most {L} ranks (asked: {world})")
    cost = level_costs(desc)
    pre = [0.0]
    for c in cost:
        pre.append(pre[-1] + c)
    INF = float("inf")
    best = [[INF] * (L + 1) for _ in range(world + 1)]       # best[q][l]: smallest possible maximum over q ranks keeping levels [0, l)
    arg = [[0] * (L + 1) for _ in range(world + 1)]
    best[0][0] = 0.0
    for q in range(1, world + 1):
        for l in range(q, L - (world - q) + 1):
            for a in range(q - 1, l):
                v = max(best[q - 1][a], pre[l] - pre[a])
                if v < best[q][l]:
                    best[q][l], arg[q][l] = v, a
    cuts = [L]
    for q in range(world, 0, -1):
        cuts.append(arg[q][cuts[-1]])
    return cuts[::-1]


def ray_partition(n: int, world: int) -> List[int]:
    """rank q renders the contiguous rays [n q / world, n (q + 1) / world) of an iteration's batch"""
    return [(n * q) // world for q in range(world + 1)]


def level_exchange_splits(n: int, S: int, cuts: List[int], rank: int) -> Dict:
    """element counts (fp32) of an iteration's three all-to-alls on a level-partitioned table, for `rank` of
    world = len(cuts) - 1: name -> (what goes to rank q, what comes from rank q), q = 0..world-1.
      feat   own levels' features of every rank's points out; every rank's levels for the own points in (blocks [m S, 2 k_q])
      demb   the way back: gradient rows of the own points, per owning rank, out; all points' rows for the own levels in
      dx     (pose phase) d loss / d x01 through the own levels of every rank's points out; world partial sums for the own points in"""
    world = len(cuts) - 1
    rs = ray_partition(n, world)
    nq = [rs[q + 1] - rs[q] for q in range(world)]
    kq = [cuts[q + 1] - cuts[q] for q in range(world)]
    m, k = nq[rank], kq[rank]
    feat = ([nq[q] * S * 2 * k for q in range(world)], [m * S * 2 * kq[q] for q in range(world)])
    return {"rays": rs, "m": m, "feat": feat, "demb": (feat[1], feat[0]),
            "dx": ([nq[q] * S * 3 for q in range(world)], [m * S * 3] * world)}


def field_exchange_model(desc, n_points: int, n_lattice: int, world: int, selected: float = 0.63) -> Dict:
    """bytes each rank RECEIVES per map iteration under the three ways of spreading the field over `world` GPUs, and what
    part of the single-GPU scatter work a rank still does:
      replicas   the table replicated, its dense gradient all-reduced (ring: 2 (N-1)/N of the buffer)
      points     the verdict's alternative: all-gather of the selected points' (x01, d_emb) rows, every rank scatters ALL of them
      levels     the table partitioned by level: features out, feature gradients back, each rank scatters its levels only"""
    N = world
    table_bytes = 4 * int(desc.n_feat) * sum(int(desc.size[l]) for l in range(int(desc.n_levels)))
    row = 8 * int(desc.n_levels)                 # features (or their gradient) of one point, all levels
    f = (N - 1) / N
    cost = level_costs(desc)
    cuts = level_partition(desc, N)
    share = max(sum(cost[cuts[q]:cuts[q + 1]]) for q in range(N)) / sum(cost)
    return {
        "replicas": {"recv_bytes": 2 * f * table_bytes,          # (the lattice is rank 0's: its share of the points is the largest)
                     "scatter_share": (n_points / N + n_lattice) / max(n_points + n_lattice, 1)},
        "points": {"recv_bytes": f * selected * n_points * (row + 12), "scatter_share": 1.0},
        "levels": {"recv_bytes": 2 * f * (n_points / N) * row, "scatter_share": share},
        "table_bytes": table_bytes, "level_cuts": cuts,
    }


# constants of the time model below (one MI355X node): what a rank can receive per second over xGMI under a collective
# (7 links x ~50 GB/s per direction; ring all-reduce and direct all-to-all both priced at a conservative 60 GB/s), the fixed
# cost of one small collective, and the single-GPU scatter's cost per (point, unit of level cost) -- 78 us for office0's
# 1.6e5 points x 16 units, 570 us for cafeteria's 3.8e5 x 47.2 (profiles/r3_notes.md): 31 ps either way.
XGMI_RECV_BYTES_PER_S = 60e9
COLLECTIVE_LATENCY_S = 30e-6
SCATTER_S_PER_POINT_UNIT = 31e-12
DECODER_S_PER_POINT = 0.93e-9          # forward + backward chain + weight gradients: 47 + 39 + 41 us at 1.36e5 points
ITERATION_FIXED_S = 50e-6              # the kernel boundaries of one iteration (9-12 launches)


def choose_field_mode(desc, n_points: int, n_lattice: int, world: int) -> Dict:
    """mapping.shard_field = auto: the cheaper of "replicas" and "levels" by estimated time per map iteration spent on what
    differs between them -- exchange (bytes / link rate + a latency per collective: 2 against 3) + the rank's share of the
    scatter.  Small tables (office0: 6.6 MB, below the per-point rows of two all-to-alls at N = 2) keep the all-reduce;
    from T = 2^19 on the level partition wins by an order of magnitude.  "points" (all-gather of point gradients, every rank
    scattering all of them) is priced for the record: it never beats "levels", whose exchange is smaller and whose scatter
    is divided."""
    mdl = field_exchange_model(desc, n_points, n_lattice, world)
    t_scatter = (n_points + n_lattice) * _sweep_units(desc) * SCATTER_S_PER_POINT_UNIT
    est = {}
    for mode, n_coll in (("replicas", 2), ("points", 2), ("levels", 3)):
        est[mode] = mdl[mode]["recv_bytes"] / XGMI_RECV_BYTES_PER_S + n_coll * COLLECTIVE_LATENCY_S + mdl[mode]["scatter_share"] * t_scatter
    ok_levels = 1 < world <= int(desc.n_levels) == 16
    mode = "levels" if ok_levels and est["levels"] <= est["replicas"] else "replicas"
    # whole map iteration, one GPU against N: the decoder's share divides by N in every mode; the sharded forms pay half as
    # many kernel boundaries again (the phases between the collectives)
    t_one = t_scatter + n_points * DECODER_S_PER_POINT + ITERATION_FIXED_S
    t_iter = {k: v + n_points * DECODER_S_PER_POINT / world + 1.5 * ITERATION_FIXED_S for k, v in est.items()}
    return {"mode": mode, "estimated_seconds": est, "single_gpu_scatter_seconds": t_scatter, "model": mdl,
            "iteration_seconds_one_gpu": t_one, "iteration_seconds": t_iter}


def shift_plan(cuts: List[int], need: List) -> List:
    """who sends which old x-planes to whom when the volume moves: need[r] = (a_r, b_r) are the old planes rank r's new
    slab reads; cuts are the (unchanged) slab boundaries of the old volume.  Returns [(src, dst, p0, p1)] with src != dst,
    in a canonical order every rank derives identically."""
    plan = []
    world = len(cuts) - 1
    for dst in range(world):
        a, b = need[dst]
        for src in range(world):
            p0, p1 = max(a, cuts[src]), min(b, cuts[src + 1])
            if p1 > p0 and src != dst:
                plan.append((src, dst, p0, p1))
    return plan


class sharded_volume(moving_volume):
    """x-slab ``rank`` of ``world`` of a moving TSDF volume: the reference's ``moving_volume`` interface
    (model/Volume.py:19-1408) over one slab.  Host bound logic is inherited unchanged (every rank follows the same camera,
    so all ranks take the same decisions); device state is the slab only."""

    def __init__(self, cfg, traj, init_pose, rank: int, world: int, dist=None, device=None, start=0):
        self.rank, self.world, self.dist = int(rank), int(world), dist
        super().__init__(cfg, traj, init_pose, start=start, device=device)

    # -- geometry of the slab
    def _cuts(self):
        return slab_bounds(int(self.vol_dim[0]), self.world)

    def _slab(self):
        c = self._cuts()
        return c[self.rank], c[self.rank + 1]

    def _n(self) -> int:
        x0, x1 = self._slab()
        return (x1 - x0) * int(self.vol_dim[1]) * int(self.vol_dim[2])

    def _alloc_voxels(self) -> int:
        """voxels to allocate: the widest slab this rank can get (slab widths differ by at most one plane)"""
        d = self.vol_dim
        return (int(d[0]) // self.world + 1) * int(d[1]) * int(d[2])

    # -- reads of the volume that need a neighbour's plane (the 1-plane halo)
    def _halo_plane(self):
        """plane x1 of tsdf and colour from the right neighbour (None on the last rank); every rank sends its first plane left"""
        plane = int(self.vol_dim[1]) * int(self.vol_dim[2])
        halo = None
        if self.dist is None or self.world == 1:             # no neighbour
            return halo
        ops = []
        host = _host_staged(self.dist, self.tsdf_vol_gpu)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        if self.rank > 0:
            first = torch.stack([self.tsdf_vol_gpu[:plane], self.color_vol_gpu[:plane]])
            ops.append(self.dist.P2POp(self.dist.isend, first.cpu() if host else first.contiguous(), self.rank - 1))
        if self.rank < self.world - 1:
            buf = torch.empty((2, plane), dtype=torch.float32, device="cpu" if host else self.device)
            ops.append(self.dist.P2POp(self.dist.irecv, buf, self.rank + 1))
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()
        if self.rank < self.world - 1:
            halo = buf.to(self.device)
        return halo

    def tri_interpolate(self, query_pc):
        """Trilinear tsdf/rgb at world points (reference model/Volume.py:760-794, kernel :337-458) on the sharded volume: a point
        is evaluated by the rank that owns the plane of its lower corner, which reads the upper plane from its slab or from
        the halo plane its right neighbour sends; the records are then summed over the ranks (exactly one rank contributes to
        each), so every rank returns what the single-GPU volume returns, bit for bit."""
        from . import _lib
        from ._lib import _F3, check, farr, ptr, stream_ptr
        self._wait_for_producer()
        pts = self._dev(query_pc).reshape(-1, 3)
        n = pts.shape[0]
        halo = self._halo_plane()
        out = torch.zeros((n, 5), dtype=torch.float32, device=self.device)
        inside = torch.zeros(n, dtype=torch.uint8, device=self.device)
        d = self.vol_dim
        x0, x1 = self._slab()
        check(_lib.load().rfx_tsdf_trilerp_slab(ptr(self.tsdf_vol_gpu), ptr(self.color_vol_gpu), int(d[0]), int(d[1]), int(d[2]), x0, x1,
                                                ptr(halo[0]) if halo is not None else None, ptr(halo[1]) if halo is not None else None,
                                                farr(_F3, self.vol_origin), self.voxel_size, ptr(pts), n, ptr(out), inside.data_ptr(),
                                                stream_ptr(self.device)), "rfx_tsdf_trilerp_slab")
        out = out * inside.unsqueeze(1).to(out.dtype)          # rows of other ranks: exact zeros
        all_reduce_sum_(self.dist, [out])
        result = out.cpu().numpy()
        notvalid = (result[:, 0] == 10.0) & (result[:, 1] == 0.0) & (result[:, 2] == 0.0) & (result[:, 3] == 0.0)
        return result, ~notvalid

    def get_truncated_pc(self, pc_num=5000000, trunc_tsdf=0.5):
        """Near-surface voxels as a point cloud (reference model/Volume.py:622-653, kernel :489-559) from a sharded volume.  The
        reference's slot scatter (`voxel_idx % pc_num`) keeps, in this build deterministically, the voxel with the HIGHEST global
        index per slot: each rank fills the slots from its slab (rfx_tsdf_truncated_pc_slab: global indices), the rank with the
        highest x-planes that hit a slot wins it (all-reduce MAX of rank + 1 over the hit flags), and the winners' records are
        summed (every other rank contributes zeros): every rank returns the single-GPU cloud, bit for bit.  Cold path
        (training.surface_weight = 0 in every shipped configuration); two all-reduces of 4 and 28 bytes per slot."""
        from . import _lib
        from ._lib import _F3, check, farr, ptr, stream_ptr
        self._wait_for_producer()
        pc = torch.zeros((pc_num, 7), dtype=torch.float32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        hit = torch.zeros(pc_num, dtype=torch.uint8, device=self.device)
        d = self.vol_dim
        x0, x1 = self._slab()
        check(_lib.load().rfx_tsdf_truncated_pc_slab(ptr(self.tsdf_vol_gpu), ptr(self.color_vol_gpu), int(d[0]), int(d[1]), int(d[2]), x0, x1,
                                                     farr(_F3, self.vol_origin), self.voxel_size, float(self.trunc_margin), int(pc_num),
                                                     float(trunc_tsdf), ptr(pc), cnt.data_ptr(), hit.data_ptr(), self.index_decode,
                                                     stream_ptr(self.device)), "rfx_tsdf_truncated_pc_slab")
        if not _alone(self.dist):
            owner = hit.to(torch.float32) * float(self.rank + 1)
            if _host_staged(self.dist, owner):
                h = owner.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX)
                owner = h.to(self.device)
            else:
                self.dist.all_reduce(owner, op=self.dist.ReduceOp.MAX)
            pc = pc * (owner == float(self.rank + 1)).unsqueeze(1).to(pc.dtype)
            all_reduce_sum_(self.dist, [pc])
        truncated_pc = pc.cpu().numpy()
        valid = (truncated_pc[:, 0] != 0.0) & (truncated_pc[:, 1] != 0.0) & (truncated_pc[:, 2] != 0.0)
        return truncated_pc[valid, :]

    def track_evaluate(self, vertex4, normal3, R, T, cand, search_size, n_cand, K9, H, W, level, level_index, value, count):
        """the tracker's nearest-voxel reads (reference model/ROtracker.py:244-259) on a sharded volume: every rank evaluates the
        pixels whose nearest voxel lies in its slab (rfx_track_evaluate_slab), the fixed-point sums and hit counts (int64) are
        added over the ranks -- integers: exactly the single-GPU values -- and every rank continues with the same numbers."""
        from . import _lib
        from ._lib import _F3, _F6, _F9, check, farr, ptr, stream_ptr
        self._wait_for_producer()
        d = self.vol_dim
        x0, x1 = self._slab()
        check(_lib.load().rfx_track_evaluate_slab(ptr(self.tsdf_vol_gpu), int(d[0]), int(d[1]), int(d[2]), x0, x1, farr(_F3, self.vol_origin),
                                                  float(self.voxel_size), ptr(vertex4), ptr(normal3), farr(_F9, R), farr(_F3, T), ptr(cand),
                                                  farr(_F6, search_size), int(n_cand), farr(_F9, K9), int(H), int(W), int(level),
                                                  int(level_index), ptr(value), ptr(count), stream_ptr(self.device)), "rfx_track_evaluate_slab")
        all_reduce_sum_(self.dist, [value, count])

    def track_search_volume(self):
        self._wait_for_producer()
        d = self.vol_dim
        return {"tsdf": self.tsdf_vol_gpu, "dim": (int(d[0]), int(d[1]), int(d[2])), "slab": self._slab(),
                "origin": self.vol_origin, "voxel": float(self.voxel_size)}

    def track_search_reduce(self, sums):
        """value_q30 / count (int64 [2, rows]) of one evaluation of the device-side search (rfx_track_search_evaluate), added
        over the slabs: one collective, exact"""
        all_reduce_sum_(self.dist, [sums])

    def copy_volume(self):
        """front -> back on this slab; remembers the layout of the copy (see update_tsdf_swap_rot_trans)"""
        super().copy_volume()
        self._back_dim = tuple(int(v) for v in self.vol_dim)

    def update_tsdf_swap_rot_trans(self, vol_bnds, old_bnds):
        """V2 across slabs: fetch the old planes this slab reads from their owners, then gather locally.

        The gather reads the BACK buffers.  version='more' re-grids without a copy_volume() of its own (a quirk kept from the
        reference, model/Volume.py:1078): the back buffers then hold whatever the last copy left.  Slab r of that stale copy is
        rank r's back slab as long as the slab cuts have not changed since, i.e. as long as the dimensions are the ones the
        copy was made with; otherwise the planes a rank would send are not the ones the single-GPU gather reads: refuse."""
        import ctypes as C
        from . import _lib
        from ._lib import _F3, check, farr, ptr, stream_ptr
        lib = _lib.load()
        old_cuts = self._cuts()
        old_dim = np.ceil((old_bnds[:, 1] - old_bnds[:, 0]) / self.voxel_size).copy(order="C").astype(int)
        back_dim = getattr(self, "_back_dim", None)
        if back_dim is not None and tuple(int(v) for v in old_dim) != back_dim:
            raise _lib.RfxError(f"sharded_volume: the back copy was made with dimensions {back_dim}, the re-gridding reads it as "
                                f"{tuple(int(v) for v in old_dim)}: slab layouts differ (copy_volume() first)")
        old_origin = old_bnds[:, 0].copy(order="C").astype(np.float32)
        self._set_geometry(vol_bnds)
        d = [int(v) for v in self.vol_dim]
        cuts = self._cuts()
        plane = int(old_dim[1]) * int(old_dim[2])
        need = []
        for r in range(self.world):
            a, b = C.c_int(0), C.c_int(0)
            check(lib.rfx_tsdf_shift_source_planes(cuts[r], cuts[r + 1], farr(_F3, self.vol_origin), int(old_dim[0]),
                                                   farr(_F3, old_origin), self.voxel_size, C.byref(a), C.byref(b)), "source_planes")
            need.append((a.value, b.value))
        a, b = need[self.rank]
        backs = self._backs()                      # copy_volume() put the old slab there (old planes [oc0, oc1))
        oc0, oc1 = old_cuts[self.rank], old_cuts[self.rank + 1]
        stage = [torch.empty(max(b - a, 0) * plane, dtype=torch.float32, device=self.device) for _ in range(3)]
        # own planes: device copy
        p0, p1 = max(a, oc0), min(b, oc1)
        if p1 > p0:
            for s_, bk in zip(stage, backs):
                s_[(p0 - a) * plane:(p1 - a) * plane].copy_(bk[(p0 - oc0) * plane:(p1 - oc0) * plane])
        # remote planes: point-to-point, canonical order
        plan = shift_plan(old_cuts, need)
        if plan and self.dist is not None:
            host = _host_staged(self.dist, stage[0])
            ops, recvs = [], []
            for (src, dst, q0, q1) in plan:
                for k in range(3):
                    if src == self.rank:
                        buf = backs[k][(q0 - oc0) * plane:(q1 - oc0) * plane]
                        ops.append(self.dist.P2POp(self.dist.isend, buf.cpu() if host else buf.contiguous(), dst))
                    elif dst == self.rank:
                        view = stage[k][(q0 - a) * plane:(q1 - a) * plane]
                        buf = torch.empty(view.shape, dtype=torch.float32) if host else view
                        ops.append(self.dist.P2POp(self.dist.irecv, buf, src))
                        if host:
                            recvs.append((view, buf))
            if ops:
                if self.device.type == "cuda":
                    torch.cuda.current_stream(self.device).synchronize()     # the back copy is complete before it is sent
                for req in self.dist.batch_isend_irecv(ops):
                    req.wait()
                for view, buf in recvs:
                    view.copy_(buf)
        t, w, c = self._vols()
        x0, x1 = cuts[self.rank], cuts[self.rank + 1]
        check(lib.rfx_tsdf_shift_slab(ptr(t), ptr(w), ptr(c), d[0], d[1], d[2], x0, x1, farr(_F3, self.vol_origin),
                                      ptr(stage[0]) if b > a else None, ptr(stage[1]) if b > a else None, ptr(stage[2]) if b > a else None,
                                      int(old_dim[0]), int(old_dim[1]), int(old_dim[2]), a, b, farr(_F3, old_origin),
                                      self.voxel_size, self.index_decode, stream_ptr(self.device)), "rfx_tsdf_shift_slab")

    def get_volume_all(self):
        """this rank's slab (three flat arrays, z fastest); ``gather_whole`` assembles the volume on every rank"""
        self._wait_for_producer()
        n = self._n()
        return self.tsdf_vol_gpu[:n].cpu().numpy(), self.weight_vol_gpu[:n].cpu().numpy(), self.color_vol_gpu[:n].cpu().numpy()

    def gather_whole(self):
        """the whole volume as three host arrays on every rank (tests / meshing): all_gather of the slabs"""
        parts = self.get_volume_all()
        if _alone(self.dist):
            return parts
        out = []
        cuts = self._cuts()
        plane = int(self.vol_dim[1]) * int(self.vol_dim[2])
        width = max(cuts[r + 1] - cuts[r] for r in range(self.world)) * plane
        where = self.device if self.dist.get_backend() == "nccl" else "cpu"      # RCCL moves device tensors only, gloo host ones
        for p in parts:
            mine = torch.zeros(width, dtype=torch.float32, device=where)
            mine[:p.size] = torch.from_numpy(p).to(where)
            bufs = [torch.empty(width, dtype=torch.float32, device=where) for _ in range(self.world)]
            self.dist.all_gather(bufs, mine)
            out.append(np.concatenate([bufs[r][:(cuts[r + 1] - cuts[r]) * plane].cpu().numpy() for r in range(self.world)]))
        return tuple(out)


class SceneShard:
    """what Mapper / ShardedIterations need to know about the process group"""

    def __init__(self, dist, rank: int, world: int):
        self.dist, self.rank, self.world = dist, int(rank), int(world)


def ShardedPipeline(config: Dict, dist, rank: int, world: int, device: str = "cuda:0", n_frames: Optional[int] = None, seed: int = 0):
    """MappingPipeline over ONE scene on ``world`` GPUs: rank 0 is the camera (it renders / reads the frame and broadcasts
    depth + colour), every rank integrates the frame into its x-slab of the moving volume, integrates keyframes into its
    replica of the global volume, and takes a strided share of each iteration's ray batch (mp_slam/sharded.py)."""
    from .pipeline import MappingPipeline

    class _Sharded(MappingPipeline):
        def __init__(self):
            self._shard = SceneShard(dist, rank, world)
            super().__init__(config, device=device, n_frames=n_frames, seed=seed)
            self.mapper.scene_shard = self._shard
            self.mapper._direct = None

        def _make_volume(self, config, traj, pose0):
            return sharded_volume(config, traj, pose0, rank, world, dist, device=self.device)

        def prefetch(self, ids):
            """frames resident in HBM on every rank: rank 0 produces them, the others receive depth and colour"""
            frames = super().prefetch(ids) if rank == 0 else None
            if rank != 0:
                H, W = self.dataset.H, self.dataset.W
                frames = {}
                for i in ids:
                    frames[i] = {"frame_id": i, "c2w": self.dataset.poses[i], "direction": self.dataset.rays_d,
                                 "rgb": torch.empty((H, W, 3), dtype=torch.float32, device=self.device),
                                 "depth": torch.empty((H, W), dtype=torch.float32, device=self.device)}
            for i in ids:
                broadcast_(dist, frames[i]["rgb"], 0)
                broadcast_(dist, frames[i]["depth"], 0)
                if rank != 0:
                    b = frames[i]
                    b["rgb255"] = torch.floor(b["rgb"] * 255.0)
                    b["c2w_dev"] = b["c2w"].to(self.device)
                    self.dataset._cache[i] = {k: b[k] for k in ("frame_id", "c2w", "rgb", "depth", "direction")}
            if self.mv_stream is not None:
                self.mv_stream.wait_stream(torch.cuda.current_stream(self.device))
            return frames

        def start(self, batch0, first_iters=None):
            """frame 0 on every rank (replicated: 200-1000 small iterations), then the replicas are made identical once:
            rank 0's parameters and Adam state (float atomics make the ranks' first-frame results differ in the last bits);
            from here on all-reduced gradients keep them identical"""
            super().start(batch0, first_iters)
            self.sync_replicas()

        def sync_replicas(self):
            # rank 0's table is whole only while no level-partitioned iteration has run since the last sync_field(): bring the
            # ranges home from their owners first, or the broadcast would overwrite them with rank 0's stale copies
            if self.mapper is not None and hasattr(self.mapper, "sync_field"):
                self.mapper.sync_field()
            with torch.no_grad():
                for prm in self.model.parameters():
                    if prm.numel():
                        broadcast_(dist, prm.data, 0)
                for opt in (self.slam.map_optimizer, self.slam.rba_optimizer):
                    for st in opt.state.values():
                        for k in ("exp_avg", "exp_avg_sq"):
                            if k in st:
                                broadcast_(dist, st[k], 0)

    return _Sharded()
