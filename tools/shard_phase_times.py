"""dev helper (one GPU): what ONE rank of a level-partitioned scene computes per BA iteration, for worlds of 1, 2, 4, 8 --
the ranks of a world are played in turn in this process (tests/test_level_shard_gpu.py), each phase timed with HIP events;
the collectives between the phases are tensor copies here and are NOT in the figures (DESIGN.md section 5 prices them).
Prints, per world: the slowest rank's lookup / render / scatter (/ pose) phase, the sliced Adam step, and their sum against
the single-GPU one-call iteration.  usage: python tools/shard_phase_times.py [cafeteria|apartment|scene0000|office0] [frames]"""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_level_shard_gpu as TL
from remixfusion_amd import _lib as L
from remixfusion_amd.dist import level_partition, slab_bounds
name = sys.argv[1] if len(sys.argv) > 1 else "cafeteria"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 11
lib = L.load()
cfg, pipe, fr = TL._pipeline(name, frames, small=False)
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
print(f"{name}: {n} rays x {S} = {n * S} points, lattice {(int(tr['smooth_pts']) - 1) ** 3}, table {enc.params.numel() * 4 / 1e6:.1f} MB, K = {K}", flush=True)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(c) for a, c in ev])) * 1e3


def adam_us(lo, hi):
    p = enc.params.detach()
    g, m1, m2 = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    t = L.AdamTensor(p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, m1.data_ptr() + 4 * lo, m2.data_ptr() + 4 * lo, hi - lo, 0.9, 0.99, 0.1, 0.01,
                     1e-15, 0.0, -0.0, 1.0)          # step size 0: the table is left as it is
    arr = (L.AdamTensor * 1)(t)
    return timed(lambda: L.check(lib.rfx_adam_step(arr, 1, st), "adam"))


for phase, clamp, map_grads, pose in (("map", False, True, False), ("pose", True, False, True)):
    B = direct._buffers(n, K, dev)
    random.seed(11)
    d = direct._fill(B, cur, poses_all.data_ptr(), K, clamp, B.p.dposes if pose else None, map_grads, None)
    d = type(d).from_buffer_copy(d)
    dt = torch.zeros_like(enc.params); dw = torch.zeros_like(B.t.dw_flat); dp = torch.zeros((K, 4, 4), device=dev); lc = torch.zeros(8, device=dev)
    d1 = type(d).from_buffer_copy(d)
    d1.d_hash, d1.d_w = (dt.data_ptr(), dw.data_ptr()) if map_grads else (None, None)
    d1.d_poses16 = dp.data_ptr() if pose else None
    d1.losses8 = lc.data_ptr()
    one = timed(lambda: L.check(lib.rfx_ba_forward_backward(C.byref(d1), B.p.ws, B.ws_bytes, st), "single"))
    one_adam = adam_us(0, enc.params.numel()) if map_grads else 0.0
    print(f"  {phase} iteration on ONE GPU: {one:7.1f} us + Adam on the table {one_adam:6.1f} us = {one + one_adam:7.1f} us", flush=True)
    for world in (2, 4, 8):
        cuts = level_partition(enc.desc, world)
        R = [TL._alloc_rank(lib, L, direct, B, d, q, world, cuts, n, S, K, dev, map_grads, pose) for q in range(world)]
        T = {k: [0.0] * world for k in ("lookup", "render", "scatter", "pose", "adam")}
        call = lambda fn, r: L.check(fn(C.byref(r["desc"]), C.byref(r["shard"]), r["wsp"], B.ws_bytes, st), "phase")
        for q, r in enumerate(R):
            T["lookup"][q] = timed(lambda: call(lib.rfx_ba_shard_lookup, r))
        for q, r in enumerate(R):
            a, e = r["rs"][q] * S, r["rs"][q + 1] * S
            r["feat_recv"][:r["m"] * S * 32].copy_(torch.cat([o["feat_send"][a:e].reshape(-1) for o in R]))
        for q, r in enumerate(R):
            T["render"][q] = timed(lambda: call(lib.rfx_ba_shard_render, r))
        for q, r in enumerate(R):
            parts = []
            for o in R:
                off = o["m"] * S * 2 * cuts[q]
                parts.append(o["demb_send"][off:off + o["m"] * S * 2 * r["k"]].view(o["m"] * S, 2 * r["k"]))
            r["demb_recv"].copy_(torch.cat(parts, 0))
        for q, r in enumerate(R):
            def scat():
                if map_grads:          # the phase accumulates: the own range is zeroed by the lookup phase in a real iteration
                    lo = int(enc.desc.offset[cuts[q]]) * 2
                    hi = (int(enc.desc.offset[cuts[q + 1] - 1]) + int(enc.desc.size[cuts[q + 1] - 1])) * 2
                call(lib.rfx_ba_shard_scatter, r)
            T["scatter"][q] = timed(scat)
            if map_grads:
                lo = int(enc.desc.offset[cuts[q]]) * 2
                hi = (int(enc.desc.offset[cuts[q + 1] - 1]) + int(enc.desc.size[cuts[q + 1] - 1])) * 2
                T["adam"][q] = adam_us(lo, hi)
        if pose:
            for q, r in enumerate(R):
                a, e = r["rs"][q] * S, r["rs"][q + 1] * S
                for j, o in enumerate(R):
                    r["dx_recv"][j, :r["m"] * S].copy_(o["dx_send"][a:e])
            for q, r in enumerate(R):
                T["pose"][q] = timed(lambda: call(lib.rfx_ba_shard_pose, r))
        tot = [sum(T[k][q] for k in T) for q in range(world)]
        worst = int(np.argmax(tot))
        recv = 4 * ((n // world) * S * 32 * 2) * (world - 1) // world + (4 * (n // world) * S * 3 * (world - 1) if pose else 0)
        print(f"  world {world}: level cuts {cuts}; slowest rank {worst}: lookup {T['lookup'][worst]:6.1f} render {T['render'][worst]:6.1f} "
              f"scatter {T['scatter'][worst]:6.1f} pose {T['pose'][worst]:6.1f} adam {T['adam'][worst]:6.1f} = {tot[worst]:7.1f} us "
              f"({(one + one_adam) / tot[worst]:4.2f}x one GPU, before the exchange of {recv / 1e6:.1f} MB per rank); ranks' totals "
              f"{[round(t) for t in tot]}", flush=True)
        del R
        torch.cuda.empty_cache()
