"""dev helper: rfx_field_backward_chain_weights + rfx_field_backward_weights at the bench's point count, many launches (for PMC)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from remixfusion_amd import _lib as L
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.model.scene_rep import JointEncoding
cfg = synthetic_config("office0")
m = JointEncoding(cfg, torch.from_numpy(np.array(cfg["mapping"]["bound"])), num_kf=8).cuda()
lib = L.load()
st = L.stream_ptr(torch.device("cuda"))
g = torch.Generator(device="cuda").manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 136093
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
x = torch.rand((n, 3), device="cuda", generator=g).clamp(0.001, 0.999).contiguous()
draw = torch.randn((n, 4), device="cuda", generator=g)
zf = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0          # fraction of rows without a gradient (real batches: 0.37)
if zf > 0:
    draw[torch.rand(n, device="cuda", generator=g) < zf] = 0.0
ws = torch.empty(int(lib.rfx_field_backward_workspace_bytes(n)) // 4 + 16, device="cuda")
desc = m._field_desc(False)
dw = torch.zeros(5312, device="cuda")
fns = (("chain_weights", lambda: lib.rfx_field_backward_chain_weights(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(ws), ws.numel() * 4, st)),
       ("dW", lambda: lib.rfx_field_backward_weights(n, L.ptr(draw), L.ptr(dw), L.ptr(dw) + 4 * 2592, L.ptr(dw) + 4 * 3104, L.ptr(dw) + 4 * 5216, L.ptr(ws), ws.numel() * 4, st)))
out = []
for name, fn in fns:
    for _ in range(3): assert fn() == 0
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = fn(); e1.record(); evs.append((e0, e1))
        assert rc == 0
    torch.cuda.synchronize()
    out.append(f"{name} {np.median([a.elapsed_time(b) for a, b in evs]) * 1e3:7.1f} us")
print(f"points {n} (zero rows {zf:.2f}): " + "  ".join(out))
