import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remixfusion_amd.model.encodings import OneBlob
g = torch.Generator().manual_seed(8)
xw = torch.rand((20000, 3), generator=g) * 2.6 - 0.8
xw[:9, 0] = torch.tensor([0.0, 1.0, -1e-9, 0.0625, 0.5, 0.99999994, -0.5, 1.5, 1.4999999])
full32 = OneBlob(16, fp16=False)(xw.cuda())
full = full32.half().float()
fast = OneBlob(16, fp16=True)(xw.cuda())
bad = (fast != full).nonzero()
print("mismatches", bad.shape[0], "of", fast.numel())
for r, c in bad[:25].tolist():
    d, k = c // 16, c % 16
    print(f"row {r} dim {d} k {k} x {xw[r, d].item():.9g} fast {fast[r, c].item():.6g} full {full[r, c].item():.6g} full32 {full32[r, c].item():.9g}")
