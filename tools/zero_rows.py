"""dev helper: fraction of sample points whose loss gradient d_raw is exactly zero in real BA iterations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
cfg = synthetic_config(name); cfg["mapping"]["first_iters"] = 50
pipe = MappingPipeline(cfg, n_frames=100)
frames = pipe.prefetch(list(range(90)))
pipe.start(frames[0])
d = pipe.mapper._direct_iterations(); d.stagewise_every = 1
for i in range(1, 80):
    pipe.step(i, frames[i])
    if i % 5 == 1 and i > 5:
        torch.cuda.synchronize()
        for k, B in d._cache.items():
            if k[0] != "stage": continue
            dr = B.t.d_raw.view(-1, 4)
            z = (dr == 0).all(dim=1).float().mean().item()
            zs = (dr[:, 3] == 0).float().mean().item()
            print(f"frame {i} rays {B.t.o.shape[0]}: points {dr.shape[0]}, d_raw rows all-zero {z:.3f}, d_sdf zero {zs:.3f}")
            break
