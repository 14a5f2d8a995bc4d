"""dev helper: long run of the frame loop -- throughput per 100 frames, allocator footprint, map quality at mapped keyframes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 501
cfg = synthetic_config(name)
if len(sys.argv) > 3 and sys.argv[3] == "tracker":           # config 3: the tracker (device-side search) drives the poses
    cfg["synthetic"].update({"tracker": True, "clutter": 48})
pipe = MappingPipeline(cfg, n_frames=nf + 10)
frames = pipe.prefetch(list(range(nf)))
pipe.start(frames[0])
torch.cuda.synchronize(); last = time.time()
for i in range(1, nf):
    pipe.step(i, frames[i])
    if i % 100 == 0:
        torch.cuda.synchronize(); now = time.time()
        print(f"frame {i}: {100 / (now - last):.1f} fps, allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, keyframes {len(pipe.mapper.keyframe)}", flush=True)
        last = now
pipe.model.train()
with torch.no_grad():
    for fid in (100, (nf // 2) // 5 * 5, (nf - 6) // 5 * 5):
        b = frames[fid]
        est = pipe.slam.est_c2w_data[fid].to("cuda")
        rgb, dep = pipe.slam.render_single(fid, b["depth"][None], b["rgb"][None], est, b["direction"], gap=4)
        valid = b["depth"][::4, ::4] > 0
        drift = float((est[:3, 3].cpu() - b["c2w"][:3, 3]).norm())
        print(f"keyframe {fid}: depth L1 {float((dep - b['depth'][::4, ::4]).abs()[valid].mean()) * 1e3:.1f} mm, "
              f"rgb L1 {float((rgb - b['rgb'][::4, ::4]).abs().mean()):.4f}, pose drift {drift * 1e3:.1f} mm")
