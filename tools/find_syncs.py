"""dev helper: list the host<->device synchronisation points of the steady-state frame loop
(torch.cuda.set_sync_debug_mode("warn") prints one warning, with a Python stack, per synchronising call)."""
import sys, os, warnings, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config(sys.argv[1] if len(sys.argv) > 1 else "office0")
cfg["mapping"]["first_iters"] = 20
nf = 41
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
for i in range(1, 21): pipe.step(i, frames[i])
torch.cuda.synchronize()
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "remixfusion_amd" in f.filename]
    seen[(str(message)[:60], " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:]))] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for i in range(21, nf): pipe.step(i, frames[i])
torch.cuda.set_sync_debug_mode("default")
for (m, where), c in seen.most_common(): print(f"{c:4d}x  {m}  @ {where}")
print("frames", nf - 21)
