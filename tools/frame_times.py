"""dev helper: per-frame wall time (synchronised) of the office0 stream and the frames at which the volume moves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0")
if len(sys.argv) > 1:          # a throwaway pipeline first, like bench.py's process warm-up
    import copy, gc
    wcfg = copy.deepcopy(cfg); wcfg["mapping"]["first_iters"] = 4
    wp = MappingPipeline(wcfg, n_frames=20, seed=1000)
    wf = wp.prefetch(list(range(12))); wp.start(wf[0])
    for i in range(1, 12): wp.step(i, wf[i])
    torch.cuda.synchronize(); del wp, wf; gc.collect()
pipe = MappingPipeline(cfg, n_frames=140)
orig = pipe.mv._set_geometry          # called once per move (not update_tsdf_swap_rot_trans: an instance override would
moves = []                            # switch the volume back to copy + gather)
def spy(*a, **k):
    moves.append(pipe.frames_done + 1); return orig(*a, **k)
pipe.mv._set_geometry = spy
frames = pipe.prefetch(list(range(130)))
pipe.start(frames[0], first_iters=20)
ts = []
for i in range(1, 130):
    torch.cuda.synchronize(); t0 = time.time()
    pipe.step(i, frames[i])
    torch.cuda.synchronize(); ts.append(time.time() - t0)
print("volume moves at frames", moves)
ts = np.array(ts) * 1e3
for a in list(range(0, 30, 5)) + [85]:
    print(a + 1, "-", a + 5, " ".join(f"{v:.2f}" for v in ts[a:a + 5]))
