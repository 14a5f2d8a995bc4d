"""dev helper: the frame loop on the autograd formulation, with torch.optim.Adam, and with host-side ray sampling."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
for kw in ({"direct_iterations": False}, {"direct_iterations": False, "fused_adam": False}, {"device_sampling": False}):
    cfg = synthetic_config("office0"); cfg["mapping"].update(kw); cfg["mapping"]["first_iters"] = 50
    pipe = MappingPipeline(cfg, n_frames=50)
    frames = pipe.prefetch(list(range(41)))
    pipe.start(frames[0])
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(1, 41): pipe.step(i, frames[i])
    torch.cuda.synchronize()
    b = frames[35]; pipe.model.train()
    with torch.no_grad():
        rgb, dep = pipe.slam.render_single(35, b["depth"][None], b["rgb"][None], pipe.slam.est_c2w_data[35], b["direction"], gap=4)
    valid = b["depth"][::4, ::4] > 0
    print(kw, f"{40 / (time.time() - t0):.1f} fps", "direct" if pipe.mapper._direct_iterations() else "autograd",
          f"depth L1 {float((dep - b['depth'][::4, ::4]).abs()[valid].mean()) * 1e3:.1f} mm")
