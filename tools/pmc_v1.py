"""dev helper (GPU box, run under rocprofv3 --pmc): V1 on the frame bench.py times last at the driver's settings
(frame 25 of the office0 stream, after frames 0..24 have been integrated), five launches, plus two calibration sweeps of
known size over the same 3.84e8-voxel volume: rfx_tsdf_filter with threshold 0 (reads every weight once, one dword per
lane, writes nothing: 1 536 000 000 B) and rfx_tsdf_copy (three arrays, 16 B per lane: 4 608 000 000 B read and written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.Volume import moving_volume
class T: kfx = kfy = kfz = 0.0; first = 0
cfg = synthetic_config("office0")
last = int(sys.argv[1]) if len(sys.argv) > 1 else 25
ds = get_dataset(cfg, device="cuda", n_frames=last + 2)
mv = moving_volume(cfg, T(), ds.poses[0].numpy().astype(np.float64))
K = ds.K()
for i in range(last):
    b = ds[i]
    mv.integrate(torch.floor(b["rgb"] * 255.0), b["depth"], K, b["c2w"].numpy(), None)
b = ds[last]
rgb = torch.floor(b["rgb"] * 255.0)
torch.cuda.synchronize()
for _ in range(5):
    mv.integrate(rgb, b["depth"], K, b["c2w"].numpy(), None)
torch.cuda.synchronize()
mv.filter_tsdf(0.0)
mv.copy_volume()
torch.cuda.synchronize()
print("n voxels", mv._n())
