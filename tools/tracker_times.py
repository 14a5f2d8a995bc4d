"""dev helper: wall time of the tracker's per-frame search and of the mapper step on the scene0000-sized stream (tracker on)"""
import os, sys, time, random, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
N = 61
random.seed(0)
cfg = synthetic_config("scene0000"); cfg["synthetic"].update({"tracker": True, "depth_noise": 0.0, "dropout": 0.0, "clutter": 48})
cfg["mapping"]["first_iters"] = 50
cfg.setdefault("pipeline", {})["gc_freeze"] = os.environ.get("GC_FREEZE", "1") == "1"
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe = MappingPipeline(cfg, n_frames=N + 8)
frames = pipe.prefetch(list(range(N)))
pipe.start(frames[0])
tt, tm = [], []
trk, mp = pipe.tracker.tracking, pipe.mapper.step
def trk2(*a):
    t0 = time.perf_counter(); r = trk(*a); tt.append(time.perf_counter() - t0); return r
def mp2(*a):
    t0 = time.perf_counter(); r = mp(*a); tm.append(time.perf_counter() - t0); return r
pipe.tracker.tracking, pipe.mapper.step = trk2, mp2
import gc
_g = {}
def _gc_cb(phase, info):
    if phase == 'start': _g['t'] = time.perf_counter()
    else:
        dt = time.perf_counter() - _g['t']
        if dt > 1e-3: print(f'gc generation {info["generation"]}: {dt * 1e3:.1f} ms, collected {info["collected"]}', flush=True)
gc.callbacks.append(_gc_cb)
for i in range(1, 11): pipe.step(i, frames[i])
torch.cuda.synchronize(); tt.clear(); tm.clear(); t0 = time.perf_counter()
for i in range(11, N): pipe.step(i, frames[i])
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("slowest mapper.step calls ms:", sorted(round(t * 1e3, 1) for t in tm)[-3:], "slowest tracking:", sorted(round(t * 1e3, 1) for t in tt)[-3:])
print(f"{(N - 11) / el:.1f} frames/s; tracking() per frame {np.mean(tt) * 1e3:.2f} ms (host wall, incl. its syncs); mapper.step per call {np.mean(tm) * 1e3:.2f} ms host wall, {len(tm)} calls; "
      f"sum per frame {(sum(tt) + sum(tm)) / (N - 11) * 1e3:.2f} ms of {el / (N - 11) * 1e3:.2f}")
