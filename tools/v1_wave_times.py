"""dev helper (-DMV_TIMING build): start / end of every wave of mv_chunks_kernel on one frame of the office0 stream, as a
distribution -- where the kernel's wall time goes that the waves' mean lifetime does not explain (launch ramp, tail).
usage: RFX_LIB_PATH=build/variants/librfx_timing.so python tools/v1_wave_times.py [frame]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixfusion_amd import _lib
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.datasets import get_dataset
from remixfusion_amd.model.Volume import moving_volume
class T: kfx = kfy = kfz = 0.0; first = 0
last = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cfg = synthetic_config("office0")
ds = get_dataset(cfg, device="cuda", n_frames=last + 2)
mv = moving_volume(cfg, T(), ds.poses[0].numpy().astype(np.float64))
K = ds.K()
for i in range(last):
    b = ds[i]; mv.integrate(torch.floor(b["rgb"] * 255.0), b["depth"], K, b["c2w"].numpy(), None)
b = ds[last]; rgb = torch.floor(b["rgb"] * 255.0)
raw = C.CDLL(_lib.LIB_PATH)
NW = 8192
for rep in range(3):
    mv.integrate(rgb, b["depth"], K, b["c2w"].numpy(), None)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2 * NW))()
    assert raw.rfx_debug_mv_times(buf, NW) == 0
    t = np.array(buf, dtype=np.float64).reshape(NW, 2) / 100.0        # us
    t0 = t[:, 0].min()
    st, en = t[:, 0] - t0, t[:, 1] - t0
    d = en - st
    q = lambda a: " ".join("%.1f" % v for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
    print("rep %d: span %.1f us | start pct[0,10,50,90,99,100] %s | end %s | lifetime %s mean %.1f" % (rep, en.max(), q(st), q(en), q(d), d.mean()))
    xcd = (np.arange(NW) // 4) % 8
    print("   per blockIdx%8: end max " + " ".join("%.1f" % en[xcd == k].max() for k in range(8)) + " | lifetime mean " + " ".join("%.1f" % d[xcd == k].mean() for k in range(8)))
