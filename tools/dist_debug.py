"""dev helper: run-to-run variation of the single-GPU pipeline vs the sharded pipeline at world = 1 (no process group)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_dist_gpu import _small_cfg, N_FRAMES
from remixfusion_amd.pipeline import MappingPipeline
from remixfusion_amd.dist import ShardedPipeline

def run(kind):
    cfg = _small_cfg()
    pipe = MappingPipeline(cfg, n_frames=N_FRAMES + 4, seed=5) if kind == "single" else ShardedPipeline(cfg, None, 0, 1, n_frames=N_FRAMES + 4, seed=5)
    frames = pipe.prefetch(list(range(N_FRAMES)))
    pipe.start(frames[0])
    snaps = [pipe.model.embed_res_fn.params.detach().clone()]
    for i in range(1, N_FRAMES):
        pipe.step(i, frames[i])
        if i % 5 == 1 and i > 1:
            snaps.append(pipe.model.embed_res_fn.params.detach().clone())
    d = pipe.mapper._direct_iterations()
    batch = pipe.dataset[N_FRAMES - 1]
    rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], -1).reshape(-1, 7).to(pipe.device)
    poses = pipe.slam.est_c2w_data[0:N_FRAMES:cfg["mapping"]["keyframe_every"]].clone()
    lc = d.map_gradients(rays, poses).clone().cpu()
    return snaps, lc, type(d).__name__

a = run("single"); b = run("single"); c = run("sharded")
print("iter classes", a[2], c[2])
for name, x, y in (("single vs single", a, b), ("single vs sharded(world=1)", a, c)):
    print(name, "losses", x[1][:4].tolist(), y[1][:4].tolist())
    for k, (s, t) in enumerate(zip(x[0], y[0])):
        print(f"   snap {k}: max |d hash| {float((s - t).abs().max()):.3e}  (scale {float(s.abs().max()):.3e})")
