"""dev helper: time rfx_tsdf_integrate at BASELINE config-2 size with HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.Volume import moving_volume

class T: kfx = kfy = kfz = 0.0; first = 0
cfg = synthetic_config("office0")
ds = get_dataset(cfg, device="cuda", n_frames=40)
mv = moving_volume(cfg, T(), ds.poses[0].numpy().astype(np.float64))
print("dims", mv.vol_dim, "bnds", mv.vol_bnds.tolist())
frames = [ds[i] for i in range(0, 40, 4)]
K = ds.K()
for b in frames[:2]:
    mv.integrate(torch.floor(b["rgb"] * 255 + 0.5), b["depth"], K, b["c2w"].numpy(), None)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(frames) + 1)]
ev[0].record()
for i, b in enumerate(frames):
    mv.integrate(torch.floor(b["rgb"] * 255 + 0.5), b["depth"], K, b["c2w"].numpy(), None)
    ev[i + 1].record()
torch.cuda.synchronize()
print("ms per integrate (incl. pack + prepass):", [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(len(frames))])
print("updated voxels (w>0):", int((mv.weight_vol_gpu > 0).sum()))
