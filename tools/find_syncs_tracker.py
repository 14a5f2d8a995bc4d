"""dev helper: find_syncs.py with the tracker on (scene0000 sizes, clutter 48): synchronising torch calls in the steady-state loop,
and the host wall time of mapper.step per call"""
import sys, os, warnings, traceback, collections, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
random.seed(0)
cfg = synthetic_config("scene0000"); cfg["synthetic"].update({"tracker": True, "depth_noise": 0.0, "dropout": 0.0, "clutter": 48})
cfg["mapping"]["first_iters"] = 50
nf = 61
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe = MappingPipeline(cfg, n_frames=nf + 8)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
for i in range(1, 11): pipe.step(i, frames[i])
torch.cuda.synchronize()
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "remixfusion_amd" in f.filename]
    seen[(str(message)[:60], " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-4:]))] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
tm = []
mp = pipe.mapper.step
def mp2(*a):
    t0 = time.perf_counter(); r = mp(*a); tm.append(time.perf_counter() - t0); return r
pipe.mapper.step = mp2
torch.cuda.set_sync_debug_mode("warn")
t0 = time.perf_counter()
for i in range(11, nf): pipe.step(i, frames[i])
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"{(nf - 11) / (time.perf_counter() - t0):.1f} frames/s; mapper.step host wall per call ms:", [round(t * 1e3, 2) for t in tm])
for (m, where), c in seen.most_common(): print(f"{c:4d}x  {m}  @ {where}")
