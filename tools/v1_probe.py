"""dev helper: V1 (rfx_tsdf_integrate) on the bench frames: HIP-event time per call and, with a -DMV_STATS build,
the kernel's work counters.  usage: [RFX_LIB_PATH=build/variants/librfx_X.so] python tools/v1_probe.py [config]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd import _lib
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.Volume import moving_volume
class T: kfx = kfy = kfz = 0.0; first = 0
name = sys.argv[1] if len(sys.argv) > 1 else "office0"
cfg = synthetic_config(name)
ds = get_dataset(cfg, device="cuda", n_frames=64)
mv = moving_volume(cfg, T(), ds.poses[0].numpy().astype(np.float64))
frames = [ds[i] for i in range(0, 60, 3)]
K = ds.K()
rgb = [torch.floor(b["rgb"] * 255 + 0.5) for b in frames]
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
stats = hasattr(raw, "rfx_debug_mv_stats")
for b, c in zip(frames[:3], rgb): mv.integrate(c, b["depth"], K, b["c2w"].numpy(), None)
torch.cuda.synchronize()
if stats:
    buf = (C.c_ulonglong * 16)(); raw.rfx_debug_mv_stats(buf, 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(frames) + 1)]
ev[0].record()
for i, (b, c) in enumerate(zip(frames, rgb)):
    mv.integrate(c, b["depth"], K, b["c2w"].numpy(), None); ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(frames))]
print("%s lib=%s integrate call us: mean %.1f min %.1f max %.1f" % (name, os.path.basename(_lib.LIB_PATH), 1e3 * np.mean(ms), 1e3 * np.min(ms), 1e3 * np.max(ms)))
if stats:
    raw.rfx_debug_mv_stats(buf, 0)
    n = len(frames)
    names = ["waves", "waves_with_work", "chunk_items", "lanes_in_zrange", "lanes_in_image", "lanes_updated", "lanes_band", "risky_chunk_items",
             "items_untouched", "lanes_in_untouched_items", "items_with_store", "items_with_near", "-", "-", "-", "-"]
    print("per frame:", {k: int(v) // n for k, v in zip(names, buf)})
