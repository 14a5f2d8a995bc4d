import sys, os, time, copy
sys.path.insert(0, "/root/repo")
import torch
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0")
w = copy.deepcopy(cfg); w["mapping"]["first_iters"] = 4
wp = MappingPipeline(w, n_frames=20, seed=1000)
wf = wp.prefetch(list(range(12))); wp.start(wf[0])
for i in range(1, 12): wp.step(i, wf[i])
torch.cuda.synchronize(); del wp, wf
import gc; gc.collect()
pipe = MappingPipeline(cfg, n_frames=40)
frames = pipe.prefetch(list(range(26)))
pipe.start(frames[0])
for i in range(1, 6): pipe.step(i, frames[i])
torch.cuda.synchronize()
mp = pipe.mapper
def T(label, fn):
    t0 = time.perf_counter(); r = fn(); print(f"{label}: {1e6 * (time.perf_counter() - t0):.0f} us"); return r
T("model.to(device)", lambda: mp.model.to(mp.device))
d = T("_direct_iterations()", lambda: mp._direct_iterations())
dev = torch.device(mp.device)
n = d._n_rays()
B = T("_buffers(n, 0)", lambda: d._buffers(n, 0, dev))
T("_buffers(n, 0) again", lambda: d._buffers(n, 0, dev))
T("_buffers(n, K)", lambda: d._buffers(n, 3, dev))
T("_descriptor(False)", lambda: d._descriptor(B, False, dev))
T("_descriptor(True)", lambda: d._descriptor(B, True, dev))
T("_loss_weights", lambda: mp.model._loss_weights(dev))
T("dataset[5]", lambda: mp.dataset[5])
for f in (6, 11):
    t0 = time.perf_counter(); pipe.step(f, frames[f]); print(f"frame {f}: host {1e3 * (time.perf_counter() - t0):.3f} ms")
    for i in range(f + 1, f + 5): pipe.step(i, frames[i])
    torch.cuda.synchronize()
mp2 = pipe.mapper
for label, fn in (("map_optimizer.zero_grad()", lambda: mp2.map_optimizer.zero_grad()), ("rba_optimizer.zero_grad()", lambda: mp2.rba_optimizer.zero_grad()),
                  ("dataset[30]", lambda: mp2.dataset[30]), ("_current_rays", lambda: mp2._current_rays({k: (v[None, ...] if isinstance(v, torch.Tensor) else torch.tensor([v])) for k, v in mp2.dataset[30].items()})),
                  ("est clone", lambda: mp2.est_c2w_data[0:31:5].clone()), ("rba(last)", lambda: mp2.model.rba(mp2._camera_ids(7)[6:])),
                  ("integrate_kf", lambda: mp2.integrate_kf({k: (v[None, ...] if isinstance(v, torch.Tensor) else torch.tensor([v])) for k, v in mp2.dataset[30].items()}, mp2.est_c2w_data[30]))):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    print(f"{label}: {1e6 * (time.perf_counter() - t0) / 20:.1f} us per call")
