#!/usr/bin/env python3
"""bench.py -- RGB-D frames/s of the mapping hot path on synthetic 640x480 streams (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is one RGB-D frame of the stream through the mapping path: TSDF integrate into the
moving 1 cm volume every frame, and -- whenever the reference's mapper would wake (every
map_every = 5 frames) -- keyframe integration into the global volume plus iters (5) map and
BA_iters (5) pose optimisation steps of the residual field (forward, backward, Adam), exactly
the reference schedule.  All frames are rendered and resident in HBM before the timed region.
Workload at N=1: BASELINE config 2 (office0 bound, 640x480, 800x800x600 voxels @ 1 cm, ground-truth-initialised poses).
At N>1 the workload is ONE scene mapped by the N GPUs together (north_star: the scene volume partitioned spatially over the
GPUs of one node) -- BASELINE config 4 (cafeteria: 1280x720, 700x700x300 voxels @ 2 cm, hash 2^21) at N = 2 and 4, config 5
(apartment: 720x480, 1600x1600x600 voxels @ 1 cm, S = 117, a marching-cubes mesh per keyframe) at N = 8: the moving volume cut
into x-slabs that all integrate the frame rank 0 broadcasts, the hash table partitioned by LEVEL (each rank looks up,
scatters into and steps its own levels for every sample point; per-point feature rows travel, all-to-all over RCCL, never
the table), each rank running the decoder on a share of every ray batch (remixfusion_amd/dist.py, mp_slam/sharded.py).
`value` = frames/s of that one camera stream (strong scaling: the scene does not grow with N).  Because the scene differs
from the N = 1 line's, the N > 1 line carries `n1_same_workload`: the SAME scene, stream and schedule on one GPU, measured in
the same process right after the sharded run (every rank maps it alone on its own GPU; rank 0's figure is printed) -- the
number a scaling efficiency has to be formed against -- and `exchange`: the bytes a rank receives per iteration.  `--rooms`
adds, as a side field, the round-1 figure of N independent rooms (one spatial partition of an N-times larger scene per GPU).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured live with HIP events on
the launch stream) and `cpu_baseline` (the C / torch CPU oracle timed on this host's cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# The host side of the frame loop is a handful of tiny numpy / torch-CPU ops per frame.  On a box that exposes 256 cores
# through a 16-CPU quota, default-sized BLAS / OpenMP pools (256 spinning threads) burn the quota and get the whole process
# throttled (cpu.stat nr_throttled), which shows up as 2x swings in frames/s.  Small pools, set before numpy/torch load.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA (v_mfma_f32_32x32x2_f32) dense peak
MLP_FLOP_PER_POINT = 2 * (81 * 32 + 32 * 16 + 66 * 32 + 32 * 3)     # 10 624
PMC_V1_FILE = "r6_pmc_v1_frame25.json"                                  # written by tools/r6_measure.sh (office0, the driver's frame)


def pmc_traffic_file(config):
    """committed rocprofv3 --pmc summary of THIS workload (tools/r6_measure.sh: one per BASELINE config)"""
    return "r6_pmc_traffic.json" if config == "office0" else f"r6_pmc_traffic_{config}.json"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="office0", help="N = 1 workload (and the rooms of --rooms)")
    ap.add_argument("--first-iters", type=int, default=None, help="override mapping.first_iters (default: config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="N = 1, office0: skip the short runs of BASELINE configs 3-5 at their one-GPU sizes (scene0000, cafeteria, apartment) "
                         "whose frames/s and binned-scatter roofline the line carries as `other_configs`")
    ap.add_argument("--side-steps", type=int, default=20, help="timed frames of each side-config run (after 10 warm-up frames)")
    ap.add_argument("--render-frames", type=int, default=3)
    ap.add_argument("--frame-times", action="store_true", help="print host and GPU time per timed frame to stderr")
    ap.add_argument("--no-process-warmup", action="store_true", help="skip the throwaway pipeline that loads kernels / primes the allocator")
    ap.add_argument("--pos-fp16", action="store_true",
                    help="opt in to OneBlob outputs rounded to fp16 on the fp16 matrix pipe (NOT the reference's precision, which "
                         "is fp32: model/encodings.py:73); reported in dtype/config")
    ap.add_argument("--rooms", action="store_true", help="N>1: also time N independent rooms (one per GPU), reported as a side field")
    ap.add_argument("--one-scene-timeout", type=float, default=900.0, help="N>1: seconds the sharded run may take before the watchdog ends it")
    ap.add_argument("--sharded-config", default=None, help="N>1: synthetic config of the ONE scene (default: cafeteria, apartment at N >= 8)")
    ap.add_argument("--shard-field", default="auto", choices=("auto", "levels", "replicas"),
                    help="N>1: hash table partitioned by level (per-point rows exchanged) or replicated (dense gradient all-reduced)")
    ap.add_argument("--no-n1", action="store_true", help="N>1: skip the one-GPU run of the same scene (n1_same_workload)")
    ap.add_argument("--no-mv-stream", action="store_true", help="V1 on the mapper's stream instead of a stream of its own (A/B)")
    ap.add_argument("--stage-events-every", type=int, default=5,
                    help="N = 1: every k-th BA iteration of the timed region records HIP events at its stage boundaries INSIDE the "
                         "one-call iteration (rfx_ba_desc.stage_events): the live per-stage times behind `roofline`; 0 = none")
    ap.add_argument("--stagewise-every", type=int, default=-1,
                    help="issue every k-th BA iteration stage by stage so that HIP events see the individual entry points (0: never; "
                         "default: about three such iterations in the timed region, spaced so that both phases are sampled)")
    ap.add_argument("--unused-gradients", action="store_true",
                    help="pose iterations also compute the map gradients the reference's backward produces and then zeroes "
                         "(mapping.unused_gradients); results are the same, only slower")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of individual librfx calls on torch's current stream (= the launch stream)."""

    def __init__(self):
        self.records = {}
        self.spread = {}
        self.enabled = False
        self.every = 4

    def wrap(self, lib, name, record_as=None):
        fn = getattr(lib, name)
        timer = self
        rec = record_as or name

        count = [0]

        def timed(*a):
            count[0] += 1
            if not timer.enabled or (count[0] % timer.every):      # sample every k-th call: keeps event overhead out of the fps
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            timer.records.setdefault(rec, []).append((e0, e1, a))
            return rc

        setattr(lib, name, timed)

    def summary(self):
        out = {}
        for name, evs in self.records.items():
            ms = np.array([e0.elapsed_time(e1) for e0, e1, _ in evs])
            med = float(np.median(ms))
            keep = ms <= 5.0 * med              # a launch that sat behind an unrelated stall (seen: one 100 ms sample
            out[name] = (len(ms), float(np.mean(ms[keep])), evs)     # among 50 of 0.1 ms) would otherwise own the mean
            self.spread[name] = (med, float(np.max(ms)), int((~keep).sum()))
        return out


class StageTimer:
    """HIP events recorded INSIDE rfx_ba_forward_backward at its stage boundaries (rfx_ba_desc.stage_events, ABI 10), on the
    stream the iteration is launched on: the device time of every stage of the very launches the frame loop runs -- fused
    prologue, shared launches and all -- for every k-th iteration of the timed region.  (Rounds 2-6 issued a few iterations
    stage by stage instead, one foreign call per entry point with events around each: ~0.15 ms more per such iteration, 3.5 % of
    the driver's 20-frame window against ~1 % for this: tools/r6_ab_stagewise.sh.)  A stage is reported under the name of the
    entry point that runs the same launches when called alone."""
    ORDER = {"map": ("start", "prologue", "forward", "loss", "chain", "weights", "scatter"),
             "pose": ("start", "prologue", "forward", "loss", "chain", "dx_table", "dx", "pose"),
             "pose+map": ("start", "prologue", "forward", "loss", "chain", "weights", "dx_table", "dx", "pose", "scatter")}
    NAMES = {("map", "chain"): "rfx_field_backward_chain_weights", ("pose", "chain"): "rfx_field_backward_chain_inputs",
             ("pose+map", "chain"): "rfx_field_backward_chain", "prologue": "ba_prologue", "forward": "rfx_field_forward",
             "loss": "ba_composite_loss_grad", "weights": "rfx_field_backward_weights", "scatter": "rfx_field_backward_scatter_merged",
             "dx_table": "rfx_field_backward_scatter", "dx": "rfx_field_backward_dx", "pose": "ba_pose_chain"}

    def __init__(self, lib, every, S, n_lattice, unused_gradients=False, pool=0):
        from remixfusion_amd import _lib
        self._lib_mod, self.lib, self.every, self.S, self.n_lattice = _lib, lib, int(every), int(S), int(n_lattice)
        self.unused = bool(unused_gradients)
        self.enabled, self.count, self.samples, self.pool, self.spread = False, 0, [], [], {}
        for _ in range(pool):                      # created before the timed region: hipEventCreate is host time
            self.pool.append(self._new_set())

    def _new_set(self):
        arr = (C.c_void_p * self._lib_mod.BA_STAGE_EVENTS)()
        for i in range(self._lib_mod.BA_STAGE_EVENTS):
            ev = C.c_void_p()
            self._lib_mod.check(self.lib.rfx_event_create(C.byref(ev)), "rfx_event_create")
            arr[i] = ev.value
        return arr

    def __call__(self, phase, n_rays):             # DirectIterations.stage_events
        # An event costs the loop ~2.7 us (tools/r6_ab_stagewise.sh: every iteration timed = 300 events in the driver's 20-frame
        # window = -6 %), so one iteration in `every` is timed: the third, eighth, ... of the timed region -- with the reference's
        # 5 map + 5 pose iterations per mapper step and every = 5 that is the third map and the third pose iteration of each step.
        if not self.enabled or self.every <= 0:
            return None
        self.count += 1
        if self.count % self.every != 3 % self.every:
            return None
        arr = self.pool.pop() if self.pool else self._new_set()
        self.samples.append(("pose+map" if phase == "pose" and self.unused else phase, int(n_rays), arr))
        return C.addressof(arr)

    def summary(self):
        """name -> (samples, mean ms, [(None, None, {"points": ...})]) like KernelTimer.summary(); fills self.spread"""
        ev_i = self._lib_mod.BA_EV
        per = {}
        for phase, n, arr in self.samples:
            order = self.ORDER[phase]
            for a, b in zip(order[:-1], order[1:]):
                ms = C.c_float()
                if self.lib.rfx_event_elapsed_ms(arr[ev_i[a]], arr[ev_i[b]], C.byref(ms)) != 0:
                    continue                       # an event this iteration did not record
                name = self.NAMES.get((phase, b)) or self.NAMES[b]
                pts = n * self.S + (self.n_lattice if b == "scatter" else 0)
                per.setdefault(name, []).append((float(ms.value), {"points": pts, "phase": phase}))
        out, self.spread = {}, {}
        for name, rows in per.items():
            ms = np.array([r[0] for r in rows])
            med = float(np.median(ms))
            keep = ms <= 5.0 * med
            out[name] = (len(ms), float(np.mean(ms[keep])), [(None, None, r[1]) for r in rows])
            self.spread[name] = (med, float(np.max(ms)), int((~keep).sum()))
        return out

    def close(self):
        for _, _, arr in self.samples:
            self.pool.append(arr)
        for arr in self.pool:
            for i in range(len(arr)):
                if arr[i]:
                    self.lib.rfx_event_destroy(arr[i])
        self.pool, self.samples = [], []


def _points(e, index):
    """points of one timed sample: KernelTimer keeps the call's arguments, StageTimer the count itself"""
    return e[2]["points"] if isinstance(e[2], dict) else e[2][index]


def cpu_baseline(cfg, frame, model_points: int):
    """Oracle timed on the host: TSDF integrate of one full frame into the full-size volume (C, the voxel range cut over
    `cores` threads) + one optimisation iteration of the field on a point sample (torch CPU, `cores` threads), scaled to
    the per-frame schedule.  A reported baseline, not a target."""
    from oracle import field_oracle as FO
    from oracle import tsdf as OT
    cores = min(os.cpu_count() or 1, 16)     # torch-CPU ops of this size stop scaling (and thrash) beyond ~16 threads
    torch.set_num_threads(cores)
    # --- V1 on the full-size volume (single-thread C)
    vol = cfg["volume"]
    dims = tuple(int(round(2 * vol[k]["len"] / vol["voxel_size"])) for k in ("x_config", "y_config", "z_config"))
    n = int(np.prod(dims))
    c2w = frame["c2w"].cpu().numpy()
    center = np.round(c2w[:3, 3])
    origin = center - np.array([vol["x_config"]["len"], vol["y_config"]["len"], vol["z_config"]["len"]])
    t, w, c = np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    cam = cfg["cam"]
    K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1]], np.float32)
    cpk = OT.pack_color(frame["rgb255"].cpu().numpy())
    depth = frame["depth"].cpu().numpy()
    t_runs = []
    for rep in range(3):                     # the same frame three times (the later runs update an already touched volume: the same
        t0 = time.time()                     # voxels, the same arithmetic); the median is reported
        u_c = OT.load().mv_integrate_threads(t, w, c, dims, origin.astype(np.float32), vol["voxel_size"], K, c2w, cpk, depth, vol["trunc"],
                                             threads=cores)
        t_runs.append(time.time() - t0)
        if rep == 0:
            upd, col = u_c
    t_v1 = float(np.median(t_runs))
    del t, w, c
    # --- one field iteration (forward + backward) on a sample of the points, torch CPU
    S = cfg["training"]["n_range_d"] + cfg["training"]["n_samples_d"]
    n_pts_iter = model_points
    sample = int(min(n_pts_iter, 1 << 18))         # round 5: the whole iteration's points (round 4 timed 32 768 of 165 727 and scaled)
    meta = FO.hashgrid_meta_from_config(cfg["grid"]["hash_size"], int(max(b[1] - b[0] for b in cfg["mapping"]["bound"]) / cfg["grid"]["voxel_sdf"]))
    g = torch.Generator().manual_seed(0)
    R = cfg["globalV"]["base_resolution"]
    fp = FO.FieldParams(hash_meta=meta, hash_table=(torch.rand(meta.n_params, generator=g) * 2e-4 - 1e-4).requires_grad_(True),
                        gbv=torch.rand(R ** 3 * 4, generator=g), gbw=torch.rand(R ** 3, generator=g), gbv_res=R,
                        W1=(torch.randn(32, 81, generator=g) * 0.1).requires_grad_(True),
                        W2=(torch.randn(16, 32, generator=g) * 0.1).requires_grad_(True),
                        W3=(torch.randn(32, 66, generator=g) * 0.1).requires_grad_(True),
                        W4=(torch.randn(3, 32, generator=g) * 0.1).requires_grad_(True),
                        c_trunc=cfg["training"]["c_trunc"], trunc=cfg["training"]["trunc"])
    x = torch.rand((sample, 3), generator=g)
    FO.query_color_sdf(fp, x[:256]).square().sum().backward()     # warm the allocator / thread pool
    t_runs = []
    for _ in range(5):
        t0 = time.time()
        raw = FO.query_color_sdf(fp, x)
        raw.square().sum().backward()
        t_runs.append(time.time() - t0)
    t_iter_sample = float(np.median(t_runs))       # median of five: the host's run-to-run noise moved the mean of two by +-20 %
    t_iter = t_iter_sample * n_pts_iter / sample
    m = cfg["mapping"]
    iters_per_frame = (m["iters"] + m["BA_iters"]) / m["map_every"]
    t_frame = t_v1 + iters_per_frame * t_iter
    return {"value": round(1.0 / t_frame, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"C oracle mv_integrate of 1 frame into {dims[0]}x{dims[1]}x{dims[2]} voxels on {cores} threads ({t_v1:.2f} s, "
                      f"{upd} voxels updated; median of 3) + torch-CPU oracle field fwd+bwd on {sample} of {n_pts_iter} points/iter "
                      f"on {cores} cores ({t_iter_sample:.2f} s, median of 5), {iters_per_frame:g} iters/frame",
            "v1_seconds": round(t_v1, 3), "field_iter_seconds_scaled": round(t_iter, 2)}


def side_configs(args):
    """BASELINE configs 3-5 at their one-GPU sizes, each as a short run of this script in a CHILD process (a fresh pipeline, its
    own allocator; this process keeps the GPU initialised and starts a child, it never exec's): frames/s and the roofline of the
    E1 backward scatter, whose binned form (tables of 2^19-2^21 entries) the office0 workload never runs.  Measured live, like
    everything else in the line; `traffic` from the committed PMC passes of each config (profiles/r6_pmc_traffic_<config>.json)."""
    import subprocess
    out = {}
    for name in ("scene0000", "cafeteria", "apartment"):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", name, "--steps", str(args.side_steps), "--warmup", "10",
               "--no-cpu-baseline", "--render-frames", "0", "--no-side-configs"]
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
            lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
            if res.returncode != 0 or not lines:
                out[name] = {"error": (res.stderr or "no output")[-300:]}
                continue
            d = json.loads(lines[-1])
            sc = d["rooflines"].get("field_backward_scatter", {})
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                         "metric": d["metric"], "workload": d["config"]["workload"], "dominant_call": d["dominant_call"],
                         "field_backward_scatter": {k: sc.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_scatter_bytes_only",
                                                                           "traffic", "points_per_launch", "avg_ms")},
                         "calls_timed": {k: v["calls_timed"] for k, v in d["kernels"].items() if "scatter" in k},
                         "seconds": round(time.perf_counter() - t0, 1)}
        except Exception as e:        # noqa: BLE001 -- a side figure must not cost the headline line
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def field_rooflines(summ, cfg, merged_scatter=True):
    """rooflines of the field kernels from HIP-event timings of their entry points (KernelTimer.summary())."""
    out = {}

    def mfma_roofline(name, flop_per_point, point_arg_index):
        cnt, ms, evs = summ[name]
        pts = float(np.mean([_points(e, point_arg_index) for e in evs]))
        ach = flop_per_point * pts / (ms * 1e-3) / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None, "points_per_launch": int(pts),
                "avg_ms": round(ms, 4), "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}

    if "rfx_field_forward" in summ:
        out["field_forward"] = mfma_roofline("rfx_field_forward", MLP_FLOP_PER_POINT, 2)
    chain_name = next((k for k in ("rfx_field_backward_chain_weights", "rfx_field_backward_chain") if k in summ), None)
    if chain_name:
        # recompute-forward + dX chain (the map phase runs the _weights variant: of dX1 only d_emb); counted as one
        # forward's FLOPs of algorithmic work; dW is the _weights stage
        out["field_backward_chain"] = mfma_roofline(chain_name, MLP_FLOP_PER_POINT, 2)
    if "rfx_field_backward_weights" in summ:
        out["field_backward_weights"] = mfma_roofline("rfx_field_backward_weights", MLP_FLOP_PER_POINT, 0)
    scat = "rfx_field_backward_scatter_merged" if "rfx_field_backward_scatter_merged" in summ else "rfx_field_backward_scatter"
    if scat in summ:
        # algorithmic bytes per point: 12 (x) + 128 (dfeat) read, 16 levels x 8 corners x 8 B scattered;
        # the merged call scatters the ray samples AND the TV lattice points in one sweep
        cnt, ms, evs = summ[scat]
        pts = float(np.mean([e[2]["points"] if isinstance(e[2], dict) else e[2][2] + (e[2][5] if scat.endswith("merged") else 0) for e in evs]))
        nbytes = pts * (12 + 128 + 1024)
        ach = nbytes / (ms * 1e-3) / 1e9
        f64 = cfg["grid"]["hash_size"] <= 17
        binned = cfg["grid"]["hash_size"] >= 19         # levels of >= 12 (dense: 8) segments of 8 192 entries exist (csrc/rfx_field.hip: level_is_binned)
        out["field_backward_scatter"] = {
            "kernel": (f"scatter_stage_kernel + grid_scatter_lds_kernel (the levels of few segments) + bin_sort_kernel + bin_reduce_kernel (the binned levels) ({scat})"
                       if binned else f"scatter_stage_kernel + grid_scatter_lds_kernel ({scat})"),
            "bound": "hbm", "achieved": round(ach, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
            "points_per_launch": int(pts), "avg_ms": round(ms, 4), "algorithmic_bytes_per_point": 12 + 128 + 1024,
            "frac_scatter_bytes_only": round(pts * 1024 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),      # SURVEY 8d's 1 024 B per point alone
            "note": ("binned levels: one walk over the points sorts 16-byte x-pair records by table segment inside each block's own "
                     "region (bin_sort), one block per segment adds its runs into 128 KB of LDS (double accumulators) and writes the "
                     "segment back (bin_reduce): the records -- 64 B per point and level written and read once -- are the traffic, "
                     "moved at 2.3-3.4 TB/s (profiles/r6_notes.md)" if binned else
                     "LDS-privatised: corner sums accumulate in 128 KB of LDS per table segment (double accumulators over "
                     "8 192 entries), then one contiguous global atomic per non-zero entry; a block = (segment, range of the "
                     "staged rows), the parts per level chosen so that all blocks are resident at once, dearest first (round 6); "
                     "bound by the walk's index arithmetic (every hashed level is walked once per segment), not HBM"),
            # algorithmic adds = points x 16 levels x 8 corners x 2 features, priced against the LDS atomic of the
            # accumulator type this table size uses (tools/micro/lds_atomic.hip: a full-wave ds_add_f64 retires in 19
            # clocks per CU, ds_add_f32 in 169; x 256 CUs x 2.4 GHz)
            "lds_atomic": {"accumulator": "f64" if f64 else "f32", "algorithmic_lane_adds": int(pts * 256),
                           "achieved_per_s": round(pts * 256 / (ms * 1e-3), 0),
                           "peak_per_s_measured": 2.07e12 if f64 else 2.03e11,
                           "frac": round(pts * 256 / (ms * 1e-3) / (2.07e12 if f64 else 2.03e11), 3)}}
    return out


FIELD_ENTRY_POINTS = ("rfx_tsdf_integrate_rgb", "rfx_field_forward", "rfx_field_backward_chain", "rfx_field_backward_chain_inputs",
                      "rfx_field_backward_chain_weights", "rfx_field_backward_weights", "rfx_field_backward_scatter",
                      "rfx_field_backward_scatter_merged", "rfx_field_backward_dx", "rfx_render_rays", "rfx_gbv_integrate",
                      "rfx_grid_encode_forward", "rfx_grid_encode_backward", "rfx_composite_forward", "rfx_mapping_loss_forward",
                      "rfx_mapping_loss_backward", "rfx_tv_forward", "rfx_tv_backward")


def wrap_entry_points(timer, lib):
    for name in FIELD_ENTRY_POINTS:
        timer.wrap(lib, name, "rfx_tsdf_integrate" if name == "rfx_tsdf_integrate_rgb" else None)
    # the optimisation steps run the forward / chain pair that shares its hash lookups (include/rfx.h): same work items,
    # reported under the un-suffixed names (points per launch = argument 2 either way)
    timer.wrap(lib, "rfx_field_forward_stash", "rfx_field_forward")
    for name in ("rfx_field_backward_chain", "rfx_field_backward_chain_inputs", "rfx_field_backward_chain_weights"):
        timer.wrap(lib, name + "_stashed", name)
    for name in ("rfx_ba_shard_lookup", "rfx_ba_shard_render", "rfx_ba_shard_scatter", "rfx_ba_shard_pose"):      # N > 1, table partitioned by level
        timer.wrap(lib, name)


def shard_rooflines(summ, exchange, k_own, n_lattice):
    """rooflines of the level-partitioned iteration's phases on rank 0 (HIP events around the library calls): the render phase
    (decoder forward + backward chain + weight gradients of the own rays: 3 x 10 624 FLOP per point) against the
    fp32 MFMA peak, the scatter phase (own levels, all points + lattice: 12 B + per level 8 B of gradient read and 8 corners x
    8 B scattered) against HBM"""
    out = {}
    pts_all = exchange["points_per_iteration"]
    if "rfx_ba_shard_render" in summ and exchange.get("rays_own"):
        cnt, ms, _ = summ["rfx_ba_shard_render"]
        pts = exchange["rays_own"] * (pts_all // exchange["rays_per_iteration"])
        ach = 3 * MLP_FLOP_PER_POINT * pts / (ms * 1e-3) / 1e12
        out["shard_render"] = {"kernel": "rfx_ba_shard_render (stash_put, field_forward<.,2>, composite_loss_grad, field_backward, field_dw_recompute, demb_rows)",
                               "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None, "points_per_launch": int(pts), "avg_ms": round(ms, 4),
                               "note": "map and pose iterations mixed (the pose phase runs no weight gradients)"}
    if "rfx_ba_shard_scatter" in summ:
        cnt, ms, _ = summ["rfx_ba_shard_scatter"]
        pts = pts_all + n_lattice
        nbytes = pts * (12 + k_own * (8 + 64))
        ach = nbytes / (ms * 1e-3) / 1e9
        out["shard_scatter"] = {"kernel": f"rfx_ba_shard_scatter ({k_own} own levels: scatter_stage + grid_scatter_lds / bin_* kernels; pose phase: grid_encode_dx_lp)",
                                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                "traffic": None, "points_per_launch": int(pts), "avg_ms": round(ms, 4)}
    return out


def sharded_workload(args, world):
    return args.sharded_config or ("apartment" if world >= 8 else "cafeteria")


def metric_name(cfg):
    cam, v = cfg["cam"], cfg["volume"]
    return f"RGB-D frames/sec mapping ({cam['W']}x{cam['H']}, {v['voxel_size'] * 100:g}cm TSDF)"


def sharded_cfg(args, world):
    from remixfusion_amd.config import synthetic_config
    cfg = synthetic_config(sharded_workload(args, world))
    if args.first_iters is not None:
        cfg["mapping"]["first_iters"] = args.first_iters
    cfg["mapping"]["unused_gradients"] = bool(args.unused_gradients)
    cfg["mapping"]["shard_field"] = args.shard_field
    if args.pos_fp16:
        cfg["pos"]["fp16_opt_in"] = True
    cfg["data"]["output"] = os.path.join(ROOT, "gpurun_out", "bench_meshes")      # config 5 writes a mesh per keyframe
    return cfg


def run_same_scene_alone(args, world, device):
    """the N > 1 line's scene, stream, schedule and step counts on ONE GPU (MappingPipeline, no process group): what the
    one-scene figure has to be compared with.  Every rank runs it on its own GPU at the same time (no rank waits for another
    inside a collective meanwhile); no collective is issued in here."""
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = sharded_cfg(args, world)
    n_frames = 1 + args.warmup + args.steps
    pipe = MappingPipeline(cfg, device=device, n_frames=n_frames + 8, seed=0)
    frames = pipe.prefetch(list(range(n_frames)))
    import gc
    gc.collect()                            # (a full pass costs 60-110 ms in a torch process: not inside the timed frames)
    pipe.start(frames[0])
    for i in range(1, 1 + args.warmup):
        pipe.step(i, frames[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1 + args.warmup, n_frames):
        pipe.step(i, frames[i])
    pipe.mapper.wait_meshes()               # in-loop mesh exports run on a worker thread: they belong to the timed region
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    del pipe, frames
    return el


def run_one_scene(args, dist, rank, world, device, timer):
    """frames/s of ONE camera stream mapped by `world` GPUs together (remixfusion_amd/dist.py): the moving volume in
    x-slabs, every rank integrating the frame rank 0 broadcast; keyframes integrated into every replica of the global
    volume; each BA iteration's ray batch shared out, loss sums and gradients all-reduced.  Barrier +
    torch.cuda.synchronize() on both sides of the timed region, elapsed = max over ranks."""
    from remixfusion_amd.dist import ShardedPipeline, broadcast_, field_exchange_model
    name = sharded_workload(args, world)
    cfg = sharded_cfg(args, world)
    n_frames = 1 + args.warmup + args.steps
    pipe = ShardedPipeline(cfg, dist, rank, world, device=device, n_frames=n_frames + 8, seed=0)
    frames = pipe.prefetch(list(range(n_frames)))            # rank 0 renders, the others receive (resident before timing)
    import gc
    gc.collect()                            # (a full pass costs 60-110 ms in a torch process: not inside the timed frames)
    pipe.start(frames[0])
    direct = pipe.mapper._direct_iterations()
    for i in range(1, 1 + args.warmup):
        pipe.step(i, frames[i])

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    barrier()
    timer.enabled = True
    it0 = dict(direct.iterations)
    t0 = time.perf_counter()
    for i in range(1 + args.warmup, n_frames):
        # the camera is rank 0: in the timed loop the frame's depth and colour travel to the other ranks (16 H W bytes)
        for k in ("depth", "rgb255"):
            broadcast_(dist, frames[i][k], 0)
        pipe.step(i, frames[i])
    pipe.mapper.wait_meshes()               # (rank 0's exports of the timed frames finish inside the timed region)
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    tt = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    iters = {k: direct.iterations[k] - it0[k] for k in it0}
    x0, x1 = pipe.mv._slab()
    cam, tr, m = cfg["cam"], cfg["training"], cfg["mapping"]
    S = tr["n_range_d"] + tr["n_samples_d"]
    hash_mb = int(pipe.model.embed_res_fn.params.numel() * 4 / 1e6 * 10) / 10
    meshes = "" if cfg["mesh"]["only_final"] else f", a marching-cubes mesh every {cfg['mesh']['vis']} frames (rank 0)"
    levels = type(direct).__name__ == "LevelShardedIterations"
    n_rays = direct._n_rays()
    P3 = (int(tr["smooth_pts"]) - 1) ** 3
    model = field_exchange_model(pipe.model.embed_res_fn.desc, n_rays * S, P3, world)
    info = {"workload": f"{name} (BASELINE config {'5' if name == 'apartment' else '4' if name == 'cafeteria' else '?'}), ONE scene on "
                        f"{world} GPUs: {cam['W']}x{cam['H']} RGB-D, moving TSDF volume {'x'.join(str(int(v)) for v in pipe.mv.vol_dim)} @ "
                        f"{cfg['volume']['voxel_size']} m in {world} x-slabs of {x1 - x0} planes, GBV 200^3 replicated, hash 2^{cfg['grid']['hash_size']} "
                        f"x16 levels ({hash_mb} MB) " + (f"partitioned by level {model['level_cuts']}" if levels else "replicated")
                        + f", {S} samples/ray, {m['iters']} map + {m['BA_iters']} pose iters every {m['map_every']} frames" + meshes,
            "partition": ("every rank integrates the frame rank 0 broadcasts into its x-slab (no voxel exchange; point-to-point plane "
                          "exchange when the volume follows the camera); "
                          + ("rank q keeps a contiguous range of the hash levels: it looks them up for all points of a batch, receives their "
                             "gradient rows, scatters and takes their Adam step; each rank runs the decoder on a contiguous share of the rays"
                             if levels else "each rank renders rays r, r + N, ... of every batch; loss sums (64 B) and gradients "
                             "all-reduced, identical Adam step on every replica")),
            "collectives_per_frame": "broadcast 16*H*W B (depth + rgb); per BA iteration "
                                     + ("2 all-to-alls of per-point rows (8 B per point and level) + all-reduce of 21 KB decoder gradients and "
                                        "64 B loss sums; pose iterations: + all-to-all of 12 B per point and all-reduce of the pose gradients"
                                        if levels else f"all-reduce 64 B + gradients ({hash_mb} MB hash table, 21 KB decoder)"),
            "backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (rehearsal: device tensors staged through the host)"),
            "unused_gradients": bool(args.unused_gradients), "pos_fp16_opt_in": bool(args.pos_fp16)}
    exchange = {"field": "levels" if levels else "replicas", "rays_per_iteration": n_rays, "points_per_iteration": n_rays * S,
                "recv_bytes_per_iteration_model": {k: int(model[k]["recv_bytes"]) for k in ("replicas", "points", "levels")},
                "scatter_share_of_one_gpu_model": {k: round(model[k]["scatter_share"], 3) for k in ("replicas", "points", "levels")}}
    choice = getattr(pipe.mapper, "field_mode_choice", None)
    if choice is not None:
        exchange["auto_choice_estimated_us"] = {k: round(v * 1e6, 1) for k, v in choice["estimated_seconds"].items()}
    # the time model's prediction beside the measurement (dist.choose_field_mode; constants in dist.py: a MODEL until RCCL ranks
    # have met): one map iteration on one GPU and on `world`, and the mapper's share of a frame if every iteration cost that
    from remixfusion_amd.dist import choose_field_mode
    mdl_t = choose_field_mode(pipe.model.embed_res_fn.desc, n_rays * S, P3, world)
    mode_now = "levels" if levels else "replicas"
    per_frame = (m["iters"] + m["BA_iters"]) / m["map_every"]
    exchange["model"] = {"map_iteration_us_one_gpu": round(mdl_t["iteration_seconds_one_gpu"] * 1e6, 1),
                         "map_iteration_us_n_gpus": round(mdl_t["iteration_seconds"][mode_now] * 1e6, 1),
                         "mapper_ms_per_frame_n_gpus_if_every_iteration_cost_that": round(mdl_t["iteration_seconds"][mode_now] * per_frame * 1e3, 3),
                         "constants": "dist.py: 60 GB/s received per rank under a collective, 30 us per collective, scatter and decoder "
                                      "rates measured on one GPU"}
    if levels:
        exchange["rays_own"], exchange["k_own"], exchange["n_lattice"] = direct.last_exchange.get("rays_own"), direct.k_own, P3
        exchange["recv_bytes_last_iteration_rank0"] = direct.last_exchange.get("recv_bytes")
        its = max(1, direct.iterations["map"] + direct.iterations["pose"])
        exchange["recv_bytes_per_iteration_rank0_mean"] = int(direct.exchanged_bytes / its)
    del pipe, frames
    return elapsed, iters, cfg, info, exchange


def run_rooms(args, dist, rank, world, device):
    """side figure (--rooms): every rank maps its own spatial partition of an N-times larger scene and exchanges the ghost
    planes of the global volume with its neighbours (the round-1 multi-GPU form): total frames/s of the N rooms."""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.dist import make_shard
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config(args.config)
    if args.first_iters is not None:
        cfg["mapping"]["first_iters"] = args.first_iters
    shard = make_shard(cfg, rank, world, dist)
    n_frames = 1 + args.warmup + args.steps
    pipe = MappingPipeline(shard.config, device=device, n_frames=n_frames + 8, seed=rank, shard=shard)
    frames = pipe.prefetch(list(range(n_frames)))
    pipe.start(frames[0])
    for i in range(1, 1 + args.warmup):
        pipe.step(i, frames[i])
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1 + args.warmup, n_frames):
        pipe.step(i, frames[i])
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    tt = torch.tensor([time.perf_counter() - t0], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return {"value": round(args.steps * world / float(tt.item()), 2), "unit": "frames/s (sum over N independent rooms)",
            "workload": f"{args.config}: one room per GPU, ghost planes of the global volume exchanged per keyframe"}


def main_sharded(args, dist, rank, world, device):
    """N > 1: the one-scene run under a watchdog thread.  A stalled collective cannot be interrupted from Python, so on a
    timeout the thread prints a line saying so (rank 0: the JSON line with "error") and ends the process with status 3 on
    every rank: a hang never reads as success.  An exception on one rank is reported the same way; its peers are inside a
    collective that rank will never join and end through their own watchdogs.  Nothing is restarted."""
    import threading
    from remixfusion_amd import _lib
    done = threading.Event()
    try:
        metric = metric_name(sharded_cfg(args, world))
    except Exception:                   # an unknown configuration is reported below, by the run itself
        metric = "RGB-D frames/sec mapping"
    base = {"metric": metric, "value": None, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": {"workload": sharded_workload(args, world)}}

    def fail(reason):
        print("[bench] " + reason, file=sys.stderr, flush=True)
        if rank == 0:
            out = dict(base)
            out["error"] = reason
            print(json.dumps(out), flush=True)
        os._exit(3)

    def watchdog():
        if not done.wait(args.one_scene_timeout):
            fail(f"rank {rank}: the sharded run did not finish within {args.one_scene_timeout} s (stalled collective?)")

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        lib = _lib.load()
        timer = KernelTimer()
        wrap_entry_points(timer, lib)
        timer.every = 4
        elapsed, iters, cfg, info, exchange = run_one_scene(args, dist, rank, world, device, timer)
        n1 = None
        if not args.no_n1:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            alone = run_same_scene_alone(args, world, device)              # symmetric: every rank, no collective inside
            ta = torch.tensor([alone, -alone], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(ta, op=dist.ReduceOp.MAX)
            n1 = {"value": round(args.steps / alone, 2), "unit": "frames/s", "ms_per_step": round(alone / args.steps * 1e3, 3),
                  "n_gpus": 1, "slowest_rank_value": round(args.steps / float(ta[0].item()), 2),
                  "fastest_rank_value": round(args.steps / -float(ta[1].item()), 2),
                  "what": "the same scene, stream, schedule, --steps and --warmup on ONE GPU (MappingPipeline), measured by every rank "
                          "on its own GPU right after the sharded run; value = rank 0's"
                          + ("; REHEARSAL: the ranks share one GPU here, so these runs contended with each other" if dist.get_backend() != "nccl" else "")}
        rooms = run_rooms(args, dist, rank, world, device) if args.rooms else None
    except Exception as e:          # noqa: BLE001 -- reported, then a non-zero exit
        import traceback
        traceback.print_exc()
        fail(f"rank {rank}: {type(e).__name__}: {e}"[:400])
    # what the process group itself saw: the number of ranks, the backend and each rank's device (all-gathered), so that the
    # line proves N ranks ran on N devices (tests/test_bench_gpu.py)
    try:
        seen = [None] * dist.get_world_size()
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        dist.all_gather_object(seen, {"rank": rank, "device_index": torch.cuda.current_device(), "pid": os.getpid(),
                                      "device_uuid": str(getattr(props, "uuid", "")), "device_name": props.name})
    except Exception as e:          # noqa: BLE001
        fail(f"rank {rank}: all_gather_object of the rank devices: {type(e).__name__}: {e}"[:400])
    done.set()
    if rank == 0:
        summ = timer.summary()
        per_kernel = {k: {"calls_timed": c, "avg_ms": round(ms, 4), "median_ms": round(timer.spread[k][0], 4), "max_ms": round(timer.spread[k][1], 4),
                          "outliers_dropped": timer.spread[k][2]} for k, (c, ms, _) in summ.items()}
        rl = field_rooflines(summ, cfg)
        if exchange.get("field") == "levels":
            rl.update(shard_rooflines(summ, exchange, exchange["k_own"], exchange["n_lattice"]))
        # dominant entry point of rank 0's share by summed device time (every 4th call was timed)
        step_kernels = {k: v for k, v in summ.items() if k != "rfx_render_rays"}
        dominant = max(step_kernels, key=lambda k: step_kernels[k][0] * step_kernels[k][1]) if step_kernels else None
        key = {"rfx_field_forward": "field_forward", "rfx_field_backward_chain": "field_backward_chain",
               "rfx_field_backward_chain_weights": "field_backward_chain", "rfx_field_backward_chain_inputs": "field_backward_chain",
               "rfx_field_backward_weights": "field_backward_weights", "rfx_field_backward_scatter": "field_backward_scatter",
               "rfx_field_backward_scatter_merged": "field_backward_scatter", "rfx_ba_shard_render": "shard_render",
               "rfx_ba_shard_scatter": "shard_scatter"}.get(dominant)
        roofline = rl.get(key) if key else None
        if roofline is None and rl:
            roofline = max(rl.values(), key=lambda r: r["avg_ms"])
        out = dict(base)
        out.update({"value": round(args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                    "dtype": "f32 (OneBlob columns rounded to fp16: --pos-fp16 opt-in)" if args.pos_fp16 else "f32",
                    "config": info, "roofline": roofline, "rooflines": rl, "kernels": per_kernel, "dominant_call": dominant,
                    "iterations_timed": iters, "cpu_baseline": None, "n1_same_workload": n1, "exchange": exchange,
                    "ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "rank_devices": seen,
                    "distinct_devices": len({(r["device_uuid"] or r["device_index"]) for r in seen}),
                    "launched_by": "bench.py itself (child ranks)" if os.environ.get("RFX_BENCH_SELF_LAUNCHED") == "1" else "external launcher",
                    "speedup_vs_n1_same_workload": round((args.steps / elapsed) / n1["value"], 3) if n1 else None,
                    "note": "rooflines are rank 0's share of each launch (1/N of the batch); cpu_baseline is reported at N = 1 only"})
        if rooms is not None:
            out["independent_rooms"] = rooms
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` without an external launcher: start the N ranks as FRESH child processes (torch.distributed.run,
    rendezvous on 127.0.0.1) and leave with their status.  This parent has not touched the GPU (no HIP call, no
    torch.cuda.is_available()) and never does; nothing is exec'ed over a running process.  The children's stderr is this
    process's own; of their stdout the JSON line passes through unchanged (see below)."""
    import subprocess
    port = os.environ.get("RFX_BENCH_MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RFX_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    # the children's stdout is filtered: ONE JSON line is the contract, and libraries of the ranks write there too (gloo's
    # "[Gloo] Rank 0 is connected to ..." notices): only lines that are JSON objects pass, the rest goes to stderr
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in child.stdout:
            t = line.strip()
            is_json = False
            if t.startswith("{") and t.endswith("}"):
                try:
                    json.loads(t)
                    is_json = True
                except ValueError:
                    pass
            (sys.stdout if is_json else sys.stderr).write(line)
            (sys.stdout if is_json else sys.stderr).flush()
        rc = child.wait()
    except KeyboardInterrupt:
        child.terminate()              # the exact child we started, never a pattern
        rc = child.wait()
    if rc != 0:
        print(f"[bench] the {args.gpus}-rank run ended with status {rc}", file=sys.stderr, flush=True)
    sys.exit(rc if 0 <= rc < 256 else 1)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args)            # before any torch.cuda call
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
                         "they must agree (the line's n_gpus is the number of ranks that ran)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    local_rank %= max(torch.cuda.device_count(), 1)      # rehearsal: several ranks may share one GPU (gloo)
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    dist = None
    # RFX_FORCE_SHARDED=1 (with RFX_DIST_FORCE_COLLECTIVES=1): the N > 1 code path with ONE rank -- a one-rank RCCL communicator
    # executes every collective of the one-scene run on a 1-GPU box (a rehearsal of the calls, not a measurement of anything)
    force_sharded = world == 1 and os.environ.get("RFX_FORCE_SHARDED") == "1"
    if world > 1 or force_sharded:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        backend = os.environ.get("RFX_DIST_BACKEND", "nccl")     # nccl == RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod

    if world > 1 or force_sharded:
        return main_sharded(args, dist, rank, world, device)

    from remixfusion_amd import _lib
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline

    lib = _lib.load()
    timer = KernelTimer()
    wrap_entry_points(timer, lib)

    cfg = synthetic_config(args.config)
    if args.first_iters is not None:
        cfg["mapping"]["first_iters"] = args.first_iters
    cfg["mapping"]["unused_gradients"] = bool(args.unused_gradients)
    if args.pos_fp16:
        cfg["pos"]["fp16_opt_in"] = True
    if args.no_mv_stream:
        cfg.setdefault("pipeline", {})["mv_stream"] = False
    n_frames = 1 + args.warmup + args.steps
    shard = None
    if not args.no_process_warmup:
        # Process-level warm-up, independent of --warmup: a throwaway pipeline of the same configuration runs two mapper
        # steps (one-call and stage-by-stage issue, both phases) and one fused render, so that code-object loads, the
        # first-use hipFuncSetAttribute calls and the allocator's first large blocks are not charged to the first timed
        # mapper step when --warmup is shorter than one keyframe interval.  It shares no state with the measured pipeline.
        import copy
        wcfg = copy.deepcopy(cfg)
        wcfg["mapping"]["first_iters"] = 4
        wp = MappingPipeline(wcfg, device=device, n_frames=20, seed=rank + 1000, shard=None)
        wf = wp.prefetch(list(range(12)))
        wp.start(wf[0])
        wd = wp.mapper._direct_iterations() if wp.mapper is not None else None
        if wd is not None:
            wd.stagewise_every = 3
        for i in range(1, 12):
            wp.step(i, wf[i])
        if wp.model is not None:
            wp.model.train()
            b = wf[11]
            c2w_w = b["c2w"].to(device)
            rd = torch.sum(b["direction"].reshape(-1, 3).to(device).unsqueeze(1) * c2w_w[None, :3, :3], -1).reshape(-1, 3).contiguous()
            wp.model.render_fused(c2w_w[:3, -1].repeat(rd.shape[0], 1).contiguous(), rd, b["depth"].reshape(-1, 1).to(device))
        torch.cuda.synchronize()
        del wp, wf, wd
        import gc
        gc.collect()
    pipe = MappingPipeline(cfg, device=device, n_frames=n_frames + 8, seed=rank, shard=shard)
    frames = pipe.prefetch(list(range(n_frames)))
    import gc
    gc.collect()        # a full pass costs 60-110 ms in a torch process: keep it out of the timed frames.  Done HERE, with the
                        # first-frame mapping (tens of ms of GPU work) still to come: a pause of that length right before the
                        # timed region -- or before the few warm-up frames -- lets the GPU idle and read 3-7 % lower
    pipe.start(frames[0])
    # The BA iterations are issued by one library call each (rfx_ba_forward_backward).  Every `--stage-events-every`-th one of
    # the timed region records HIP events at its stage boundaries inside that call (StageTimer): the per-stage device times of
    # the launches the loop runs.  `--stagewise-every k` (k > 0) issues every k-th iteration stage by stage instead -- one
    # foreign call per entry point with KernelTimer's events around each, ~0.15 ms more per such iteration (rounds 2-6's way).
    direct = pipe.mapper._direct_iterations() if pipe.mapper is not None else None
    stages = None
    if direct is not None:
        if args.stagewise_every > 0:
            direct.stagewise_every = args.stagewise_every
            direct.before_stagewise = pipe.sync_volume      # time the entry points without V1 running on the other stream
            timer.every = 1
        elif args.stage_events_every > 0 and hasattr(direct, "stage_events"):
            tr_ = cfg["training"]
            stages = StageTimer(lib, args.stage_events_every, tr_["n_range_d"] + tr_["n_samples_d"], (int(tr_["smooth_pts"]) - 1) ** 3,
                                unused_gradients=args.unused_gradients, pool=2 * args.steps // args.stage_events_every + 2)
            direct.stage_events = stages
    for i in range(1, 1 + args.warmup):
        pipe.step(i, frames[i])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    timer.enabled = True
    if stages is not None:
        stages.enabled = True
    it0 = dict(direct.iterations) if direct is not None else None
    t0 = time.perf_counter()
    frame_marks = []
    if args.frame_times:
        gc_log, gc_t = [], [0.0]

        def _gc_cb(phase, info):
            if phase == "start":
                gc_t[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], (time.perf_counter() - gc_t[0]) * 1e3, (time.perf_counter() - t0) * 1e3))
        gc.callbacks.append(_gc_cb)
    for i in range(1 + args.warmup, n_frames):
        pipe.step(i, frames[i])
        if args.frame_times:                             # dev aid: where inside the timed region the time goes
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            frame_marks.append((i, time.perf_counter() - t0, ev))
    if pipe.mapper is not None:
        pipe.mapper.wait_meshes()           # in-loop mesh exports run on a worker thread: they belong to the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if stages is not None:
        stages.enabled = False
    if args.frame_times and rank == 0:
        gc.callbacks.remove(_gc_cb)
        for g, dur, at in gc_log:
            print(f"gc generation {g}: {dur:.3f} ms at +{at:.3f} ms", file=sys.stderr)
        prev_h, prev_e = 0.0, None
        for i, h, ev in frame_marks:
            g = frame_marks[0][2].elapsed_time(ev)
            print(f"frame {i:4d} host +{(h - prev_h) * 1e3:7.3f} ms  gpu mark {g:9.3f} ms (+{0.0 if prev_e is None else g - prev_e:7.3f})",
                  file=sys.stderr)
            prev_h, prev_e = h, g
    iters = {k: direct.iterations[k] - it0[k] for k in it0} if direct is not None else {"map": 0, "pose": 0}
    if dist is not None:
        tt = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- V1 alone: in the timed region it runs on its own stream and shares the GPU with the mapper's kernels, which
    #      stretches its launches; time the same call on an otherwise idle GPU as well (after the timed region)
    v1_alone_ms = None
    if getattr(pipe, "mv_stream", None) is not None:
        f = frames[n_frames - 1]
        pose_np = f["c2w"].numpy().astype(np.float64)
        torch.cuda.synchronize()
        evs = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pipe.mv.integrate(f["rgb255"], f["depth"], pipe.K, pose_np, pipe.mv.vol_bnds)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        v1_alone_ms = float(np.median([a.elapsed_time(b) for a, b in evs[2:]]))

    # ---- rays/s of the fused full-frame renderer (second headline number)
    render = None
    if pipe.model is not None and args.render_frames > 0:
        pipe.model.train()
        b = frames[n_frames - 1]
        ray_d = b["direction"].reshape(-1, 3).to(device)
        c2w = b["c2w"].to(device)
        rays_d = torch.sum(ray_d.unsqueeze(1) * c2w[None, :3, :3], -1).reshape(-1, 3).contiguous()
        rays_o = c2w[:3, -1].repeat(rays_d.shape[0], 1).contiguous()
        td = b["depth"].reshape(-1, 1).to(device)
        pipe.model.render_fused(rays_o, rays_d, td)
        torch.cuda.synchronize()
        timer.enabled = True
        t1 = time.perf_counter()
        for _ in range(args.render_frames):
            pipe.model.render_fused(rays_o, rays_d, td)
        torch.cuda.synchronize()
        render = rays_d.shape[0] * args.render_frames / (time.perf_counter() - t1)
        timer.enabled = False

    # ---- roofline of the dominant kernel (by summed device time over the timed region)
    summ = timer.summary()
    spread = dict(timer.spread)
    if stages is not None:
        summ.update(stages.summary())
        spread.update(stages.spread)
        stages.close()
    cam, tr = cfg["cam"], cfg["training"]
    S = tr["n_range_d"] + tr["n_samples_d"]
    per_kernel = {k: {"calls_timed": c, "avg_ms": round(ms, 4), "median_ms": round(spread[k][0], 4), "max_ms": round(spread[k][1], 4),
                      "outliers_dropped": spread[k][2]}
                  for k, (c, ms, _) in summ.items()}
    step_kernels = {k: v for k, v in summ.items() if k != "rfx_render_rays"}
    # device time per entry point over the timed region = avg x number of launches; the BA-iteration stages were only
    # visible (hence counted) in every `stagewise_every`-th iteration
    # the BA-iteration stages were only visible in every `stagewise_every`-th iteration: their launch counts come from the
    # number of map / pose iterations issued in the timed region
    per_frame = ("rfx_tsdf_integrate", "rfx_gbv_integrate", "rfx_render_rays")
    map_only = ("rfx_field_backward_chain_weights", "rfx_field_backward_weights", "rfx_field_backward_scatter_merged", "rfx_tv_forward",
                "rfx_tv_backward", "rfx_grid_encode_forward")
    pose_only = ("rfx_field_backward_chain_inputs", "rfx_field_backward_scatter", "rfx_field_backward_dx", "ba_pose_chain")
    if args.unused_gradients:                      # the pose phase then runs the map-gradient stages as well (on the full chain)
        map_only, pose_only = ("rfx_field_backward_chain_weights",), pose_only + ("rfx_field_backward_chain",)
    launches = {}
    for k, (cnt, ms, _) in step_kernels.items():
        if k in per_frame or direct is None:
            launches[k] = cnt
        elif k in map_only:
            launches[k] = iters["map"]
        elif k in pose_only:
            launches[k] = iters["pose"]
        else:
            launches[k] = iters["map"] + iters["pose"]
    dominant = max(step_kernels, key=lambda k: launches[k] * step_kernels[k][1]) if step_kernels else None
    roofline = None
    extra_rooflines = field_rooflines(summ, cfg)
    if "rfx_render_rays" in summ:
        cnt, ms, evs = summ["rfx_render_rays"]
        pts = float(evs[0][2][6]) * S
        ach = MLP_FLOP_PER_POINT * pts / (ms * 1e-3) / 1e12
        extra_rooflines["render_rays"] = {"kernel": "rfx_render_rays", "bound": "mfma", "achieved": round(ach, 3),
                                          "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                                          "points_per_launch": int(pts), "avg_ms": round(ms, 4)}
    if "rfx_tsdf_integrate" in summ:
        # algorithmic bytes: 16 U + 8 C + 8 H W; U, C counted by the CPU oracle on one timed frame
        cnt, ms, _ = summ["rfx_tsdf_integrate"]
        uc = None
        if not args.no_cpu_baseline:
            try:
                from oracle import tsdf as OT
                mv = pipe.mv
                f = frames[n_frames - 1]
                n = int(np.prod(mv.vol_dim))
                tt, ww, cc = np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
                uc = OT.load().mv_integrate(tt, ww, cc, mv.vol_dim, mv.vol_origin, mv.voxel_size, pipe.K,
                                            f["c2w"].cpu().numpy(), OT.pack_color(f["rgb255"].cpu().numpy()),
                                            f["depth"].cpu().numpy(), mv.trunc_margin)
                del tt, ww, cc
            except Exception as e:   # the oracle is optional at bench time
                uc = None
                print(f"[bench] oracle voxel count unavailable: {e}", file=sys.stderr)
        if uc is not None:
            nbytes = 16 * uc[0] + 8 * uc[1] + 8 * cam["H"] * cam["W"]
            ach = nbytes / (ms * 1e-3) / 1e9
            extra_rooflines["tsdf_integrate"] = {"kernel": "rfx_tsdf_integrate_rgb (mv_frame + mv_rows + mv_chunks kernels)", "bound": "hbm",
                                                 "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                                 "updated_voxels": int(uc[0]), "colour_voxels": int(uc[1]),
                                                 "algorithmic_bytes": int(nbytes), "avg_ms": round(ms, 4)}
            if v1_alone_ms:
                extra_rooflines["tsdf_integrate"].update({
                    "note": "avg_ms is measured in the timed region, where V1 runs on its own stream concurrently with the "
                            "mapper's kernels; *_alone: the same call on an otherwise idle GPU, after the timed region",
                    "avg_ms_alone": round(v1_alone_ms, 4), "achieved_alone": round(nbytes / (v1_alone_ms * 1e-3) / 1e9, 1),
                    "frac_alone": round(nbytes / (v1_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    # HBM traffic per launch: NOT measured by this run (PMC counters need rocprofv3 passes of their own).  The figure is
    # taken from the committed summary of those passes, profiles/r6_pmc_traffic[_<config>].json (tools/summarize_pmc.py), which
    # carries the digest of the kernel sources the passes ran on (remixfusion_amd.build.sources_digest): when this tree's
    # digest differs -- the binary timed here is not the one the counters saw -- traffic stays null and says so.
    from remixfusion_amd.build import sources_digest
    digest_now = sources_digest()
    try:
        PMC_TRAFFIC_FILE = pmc_traffic_file(args.config)
        pmc_all = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
        pmc, tag = pmc_all.get("kernels", {}), pmc_all.get("measured_at_commit", "unknown")
        stale = pmc_all.get("kernel_sources_digest") != digest_now
        if stale:
            pmc = {}
            for r_ in extra_rooflines.values():
                r_["traffic_source"] = (f"null: profiles/{PMC_TRAFFIC_FILE} was measured on kernel sources {pmc_all.get('kernel_sources_digest')} "
                                        f"(commit {tag}), this tree is {digest_now}")
        for rk, kns in (("field_backward_scatter", ("rfx::grid_scatter_lds_kernel", "rfx::scatter_stage_kernel", "rfx::bin_sort_kernel", "rfx::bin_reduce_kernel")),
                        ("field_forward", ("rfx::field_forward_kernel<false, 1>",)),
                        ("field_backward_chain", ("rfx::field_backward_kernel<false, true, false, true>",)),
                        ("field_backward_weights", ("rfx::field_dw_recompute_kernel", "rfx::field_dw_reduce_kernel")),
                        ("render_rays", ("rfx::render_rays_kernel<false, 1>",)),
                        ("tsdf_integrate", ("rfx::mv_chunks_kernel", "rfx::mv_rows_kernel", "rfx::mv_frame_kernel"))):
            keys = [k for k in pmc if any(kn in k for kn in kns)]
            if rk in extra_rooflines and keys:
                extra_rooflines[rk]["traffic"] = int(sum(pmc[k]["hbm_bytes"] for k in keys))
                extra_rooflines[rk]["traffic_source"] = (f"profiles/{PMC_TRAFFIC_FILE}, rocprofv3 --pmc passes at commit {tag} on these kernel sources ({digest_now}; not this run): "
                                                         "FETCH_SIZE x2 (gfx950 counts 128-B reads at 64 B) + WRITE_SIZE; " + " + ".join(keys))
    except Exception:
        pass
    # V1: traffic of the SAME frame the algorithmic bytes above were counted on (frame 1 + warmup + steps - 1), when the
    # committed passes cover it (profiles/r6_pmc_v1_frame25.json: the driver's settings) and ran on these kernel sources
    try:
        v1p = json.load(open(os.path.join(ROOT, "profiles", PMC_V1_FILE)))
        if "tsdf_integrate" in extra_rooflines and v1p.get("kernel_sources_digest") != digest_now:
            extra_rooflines["tsdf_integrate"]["traffic"] = None
            extra_rooflines["tsdf_integrate"]["traffic_source"] = (f"null: profiles/{PMC_V1_FILE} was measured on kernel sources "
                                                                   f"{v1p.get('kernel_sources_digest')}, this tree is {digest_now}")
        elif "tsdf_integrate" in extra_rooflines and v1p["frame"] == n_frames - 1 and v1p["config"] == args.config:
            extra_rooflines["tsdf_integrate"]["traffic"] = int(sum(k["hbm_bytes"] for k in v1p["kernels"].values()))
            extra_rooflines["tsdf_integrate"]["traffic_raw_counters"] = int(sum(k["hbm_bytes_raw"] for k in v1p["kernels"].values()))
            extra_rooflines["tsdf_integrate"]["traffic_source"] = (
                f"profiles/{PMC_V1_FILE}: rocprofv3 --pmc passes of tools/pmc_v1.py at commit {v1p['measured_at_commit']} on frame "
                f"{v1p['frame']} (this frame; not this run): 2 x FETCH_SIZE + WRITE_SIZE = the bytes the L2s pulled in (every miss is a whole "
                "128-byte line tallied at 64, calibrated on V1's own access shape in round 5: profiles/r5_fetch_calib.txt) -- Infinity-Cache hits "
                "included: ~80 MB of it are the frame's images fetched once per XCD; traffic_raw_counters = FETCH_SIZE + WRITE_SIZE")
        elif "tsdf_integrate" in extra_rooflines:
            extra_rooflines["tsdf_integrate"]["traffic"] = None         # passes of other frames are not comparable
            extra_rooflines["tsdf_integrate"].pop("traffic_source", None)
    except Exception:
        pass
    key = {"rfx_field_forward": "field_forward", "rfx_field_backward_chain": "field_backward_chain",
           "rfx_field_backward_chain_weights": "field_backward_chain", "rfx_field_backward_chain_inputs": "field_backward_chain",
           "rfx_field_backward_weights": "field_backward_weights", "rfx_field_backward_scatter": "field_backward_scatter",
           "rfx_field_backward_scatter_merged": "field_backward_scatter",
           "rfx_tsdf_integrate": "tsdf_integrate"}.get(dominant)
    roofline = extra_rooflines.get(key) if key else None
    if roofline is None and extra_rooflines:
        roofline = next(iter(extra_rooflines.values()))

    base = None
    if not args.no_cpu_baseline and world == 1:          # the host baseline is reported at N = 1 only
        m = cfg["mapping"]
        n_rays = m["sample"] + max(m["sample"] // 8, m["min_pixels_cur"])
        tv_pts = (tr["smooth_pts"] - 1) ** 3
        base = cpu_baseline(cfg, frames[n_frames - 1], n_rays * S + tv_pts)

    vol_dim_str = "x".join(str(int(v)) for v in pipe.mv.vol_dim)
    streams_str = "V1 on its own HIP stream, concurrent with the mapper" if getattr(pipe, "mv_stream", None) is not None else "one stream"
    other = None
    if world == 1 and args.config == "office0" and not args.no_side_configs:
        del pipe, frames
        import gc as _gc
        _gc.collect()
        torch.cuda.empty_cache()
        other = side_configs(args)

    fps = args.steps * world / elapsed
    out = {
        "metric": metric_name(cfg), "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (OneBlob columns rounded to fp16: --pos-fp16 opt-in)" if args.pos_fp16 else "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {cam['W']}x{cam['H']} RGB-D, moving TSDF volume "
                               f"{vol_dim_str} @ {cfg['volume']['voxel_size']} m, GBV 200^3, "
                               f"hash 2^{cfg['grid']['hash_size']} x16 levels, {S} samples/ray, "
                               f"{cfg['mapping']['iters']} map + {cfg['mapping']['BA_iters']} pose iters every {cfg['mapping']['map_every']} frames, poses initialised from the ground-truth trajectory and refined by the RBA pose MLP",
                   "unused_gradients": bool(args.unused_gradients), "pos_fp16_opt_in": bool(args.pos_fp16),
                   "streams": streams_str,
                   "note": "pose iterations step only the pose MLP (reference mapper.py:494-499); the map gradients its backward also "
                           "produces and zeroes are computed only with --unused-gradients (same parameters and poses either way)",
                   "partition": "single volume"},
        "render_rays_per_s": round(render, 1) if render else None,
        "roofline": roofline, "rooflines": extra_rooflines, "kernels": per_kernel, "dominant_call": dominant,
        "iterations_timed": iters,
        "kernel_timing": (f"HIP events recorded at the stage boundaries INSIDE the one-call iteration (rfx_ba_desc.stage_events), every "
                          f"{args.stage_events_every}th BA iteration of the timed region, on the launch stream, V1 running beside "
                          "them on its own stream as in the loop; V1 / G1 / render: events around their entry points"
                          if stages is not None else
                          f"stage-by-stage issue of every {args.stagewise_every}th BA iteration with HIP events around each entry point"
                          if args.stagewise_every > 0 else "V1 / G1 / render only"),
        "cpu_baseline": base,
        "other_configs": other,
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
