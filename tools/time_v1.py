"""dev helper: V1 kernel timing on the bench frames (HIP events around rfx_tsdf_integrate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.Volume import moving_volume
class T: kfx = kfy = kfz = 0.0; first = 0
cfg = synthetic_config("office0")
ds = get_dataset(cfg, device="cuda", n_frames=64)
mv = moving_volume(cfg, T(), ds.poses[0].numpy().astype(np.float64))
frames = [ds[i] for i in range(0, 60, 3)]
K = ds.K()
rgb = [torch.floor(b["rgb"] * 255 + 0.5) for b in frames]
for b, c in zip(frames[:3], rgb): mv.integrate(c, b["depth"], K, b["c2w"].numpy(), None)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(frames) + 1)]
ev[0].record()
for i, (b, c) in enumerate(zip(frames, rgb)):
    mv.integrate(c, b["depth"], K, b["c2w"].numpy(), None); ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(frames))]
print("integrate call ms: mean %.4f min %.4f max %.4f" % (np.mean(ms), np.min(ms), np.max(ms)))
