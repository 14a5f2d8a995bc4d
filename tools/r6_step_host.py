"""dev helper (GPU box): where a mapper step's time goes between its iterations, at bench.py's office0 settings, WITHOUT a profiler.
Host: wall time of the step's preamble (everything before the first BA iteration), of each iteration's issue, of the tail.
GPU: HIP events (fence-free) on the mapper's stream at step entry, in front of the first iteration, behind the last iteration and
at step exit: the device-side length of preamble / iterations / tail, and the distance from one step's exit to the next one's entry."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from remixfusion_amd import _lib
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline

cfg = synthetic_config(os.environ.get("CONFIG", "office0"))
nf = int(os.environ.get("FRAMES", 46))
if os.environ.get("NO_MV_STREAM"):
    cfg.setdefault("pipeline", {})["mv_stream"] = False
if os.environ.get("WARM"):            # bench.py's process warm-up: a throwaway pipeline of the same configuration first
    import copy, gc
    w = copy.deepcopy(cfg); w["mapping"]["first_iters"] = 4
    wp = MappingPipeline(w, n_frames=20, seed=1000)
    wf = wp.prefetch(list(range(12))); wp.start(wf[0])
    for i in range(1, 12): wp.step(i, wf[i])
    torch.cuda.synchronize(); del wp, wf; gc.collect()
pipe = MappingPipeline(cfg, n_frames=nf + 8)
if os.environ.get("NO_V1"):          # what the loop costs without the volume's work (an upper bound for any V1 speed-up)
    pipe.mv.integrate = lambda *a, **k: None
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
lib = _lib.load()
st = _lib.stream_ptr(pipe.device)


def ev():
    e = C.c_void_p()
    _lib.check(lib.rfx_event_create(C.byref(e)), "ev")
    return e


def rec(e):
    torch.cuda.current_stream().synchronize if False else None
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventRecord(e, C.c_void_p(st))


hip = C.CDLL("libamdhip64.so")
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
mp = pipe.mapper
direct = mp._direct_iterations()
log = []
cur = {}
orig_step, orig_map, orig_pose = mp.step, direct.map_iteration, direct.pose_iteration


def step(i):
    cur.clear()
    cur.update(t_in=time.perf_counter(), e_in=ev(), its=[], e_first=None)
    hip.hipEventRecord(cur["e_in"], st)
    orig_step(i)
    cur["e_out"] = ev()
    hip.hipEventRecord(cur["e_out"], st)
    cur["t_out"] = time.perf_counter()
    log.append(dict(cur))


def wrap(fn):
    def f(*a):
        if cur["e_first"] is None:
            cur["e_first"] = ev(); hip.hipEventRecord(cur["e_first"], st); cur["t_first"] = time.perf_counter()
        t0 = time.perf_counter()
        r = fn(*a)
        cur["its"].append(time.perf_counter() - t0)
        if os.environ.get("ITS"): cur.setdefault("its_list", []).append(round(1e6 * (time.perf_counter() - t0)))
        cur["e_last"] = ev(); hip.hipEventRecord(cur["e_last"], st); cur["t_last"] = time.perf_counter()
        return r
    return f


mp.step, direct.map_iteration, direct.pose_iteration = step, wrap(orig_map), wrap(orig_pose)
torch.cuda.synchronize()
t0 = time.perf_counter()
walls = []
for i in range(1, nf):
    t = time.perf_counter()
    pipe.step(i, frames[i])
    walls.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
print(f"{nf - 1} frames in {(time.perf_counter() - t0) * 1e3:.2f} ms = {(time.perf_counter() - t0) * 1e3 / (nf - 1):.3f} ms/frame")
print("host ms per frame:", " ".join(f"{w:.2f}" for w in walls))


def el(a, b):
    ms = C.c_float()
    lib.rfx_event_elapsed_ms(a, b, C.byref(ms))
    return ms.value * 1e3


print("step: host preamble / iterations(sum, max) / tail us | gpu preamble / iterations / tail us | gpu exit->next entry us")
for k, s in enumerate(log):
    nxt = el(s["e_out"], log[k + 1]["e_in"]) if k + 1 < len(log) else float("nan")
    print(f"  {k:2d}: host {1e6 * (s['t_first'] - s['t_in']):7.0f} / {1e6 * sum(s['its']):7.0f} {1e6 * max(s['its']):5.0f} / {1e6 * (s['t_out'] - s['t_last']):6.0f}"
          f" | gpu {el(s['e_in'], s['e_first']):7.0f} / {el(s['e_first'], s['e_last']):7.0f} / {el(s['e_last'], s['e_out']):6.0f} | {nxt:7.0f}" + (f"  its {s.get('its_list')}" if os.environ.get("ITS") else ""))
