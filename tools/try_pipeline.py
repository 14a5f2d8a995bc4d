"""dev helper: run the mapping pipeline for a few frames and print timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline

name = sys.argv[1] if len(sys.argv) > 1 else "office0"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 41
first = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = synthetic_config(name)
t0 = time.time()
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
torch.cuda.synchronize(); print("setup+prefetch s", round(time.time() - t0, 2))
t0 = time.time()
pipe.start(frames[0], first_iters=first)
torch.cuda.synchronize(); print("first frame mapping s", round(time.time() - t0, 2), "iters", first)
t0 = time.time()
for i in range(1, nf):
    pipe.step(i, frames[i])
host = time.time() - t0
torch.cuda.synchronize()
dt = time.time() - t0
print(f"host enqueue {host:.3f}s of {dt:.3f}s (GPU trailing {dt - host:.3f}s)")
print(f"{nf - 1} frames in {dt:.3f}s -> {(nf - 1) / dt:.1f} fps; mapping_idx {int(pipe.slam.mapping_idx[0]) if pipe.slam else None}")
if pipe.slam is None:
    print("tsdf-only: updated voxels", int((pipe.mv.weight_vol_gpu > 0).sum())); sys.exit(0)
m = pipe.model
m.train()
with torch.no_grad():
    b = frames[nf - 1]
    rgb, dep = pipe.slam.render_single(nf - 1, b["depth"][None], b["rgb"][None], b["c2w"], b["direction"], gap=4)
    valid = b["depth"][::4, ::4] > 0
    print("render depth L1 (m):", float((dep - b["depth"][::4, ::4]).abs()[valid].mean()),
          "rgb L1:", float((rgb - b["rgb"][::4, ::4]).abs().mean()))
    torch.cuda.synchronize(); t0 = time.time()
    rgb, dep = pipe.slam.render_single(nf - 1, b["depth"][None], b["rgb"][None], b["c2w"], b["direction"])
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"full-frame render {dt * 1e3:.2f} ms -> {rgb.shape[0] * rgb.shape[1] / dt / 1e6:.1f} Mrays/s")
