"""dev helper: forward / chain / chain_weights / dW times on ray-shaped points (coherent samples along rays) at the bench's size."""
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
S = 59
for n_rays in (int(a) for a in (sys.argv[1:] or ["2048", "2222", "2307"])):
    o = torch.rand((n_rays, 1, 3), device="cuda", generator=g) * 0.2 + 0.4
    d = torch.randn((n_rays, 1, 3), device="cuda", generator=g); d = d / d.norm(dim=-1, keepdim=True)
    t = torch.linspace(0.02, 0.45, S, device="cuda")[None, :, None]
    x = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3).contiguous()
    n = x.shape[0]
    raw = torch.empty((n, 4), device="cuda")
    draw = torch.randn((n, 4), device="cuda", generator=g)
    ws = torch.empty(int(lib.rfx_field_backward_workspace_bytes(n)) // 4 + 16, device="cuda")
    desc = m._field_desc(False)
    dw = torch.zeros(5312, device="cuda")
    res = []
    for name, fn in (("forward", lambda: lib.rfx_field_forward(C.byref(desc), L.ptr(x), n, L.ptr(raw), st)),
                     ("chain", lambda: lib.rfx_field_backward_chain(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(ws), ws.numel() * 4, st)),
                     ("chain_weights", lambda: lib.rfx_field_backward_chain_weights(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(ws), ws.numel() * 4, st)),
                     ("dW", lambda: lib.rfx_field_backward_weights(n, L.ptr(draw), L.ptr(dw), L.ptr(dw) + 4 * 2592, L.ptr(dw) + 4 * 3104, L.ptr(dw) + 4 * 5216, L.ptr(ws), ws.numel() * 4, st)),
                     ("chain_inputs", lambda: lib.rfx_field_backward_chain_inputs(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(ws), ws.numel() * 4, st)),
                     ("fwd_stash", lambda: lib.rfx_field_forward_stash(C.byref(desc), L.ptr(x), n, L.ptr(raw), L.ptr(ws), ws.numel() * 4, st)),
                     ("chain_w_st", lambda: lib.rfx_field_backward_chain_weights_stashed(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(ws), ws.numel() * 4, st)),
                     ("chain_in_st", lambda: lib.rfx_field_backward_chain_inputs_stashed(C.byref(desc), L.ptr(x), n, L.ptr(draw), L.ptr(ws), ws.numel() * 4, st))):
        for _ in range(3): fn()
        evs = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(); e1.record(); evs.append((e0, e1))
            assert rc == 0
        torch.cuda.synchronize()
        res.append(f"{name} {np.median([a.elapsed_time(b) for a, b in evs]) * 1e3:6.1f}")
    print(f"rays {n_rays:5d} points {n:7d} (us): " + "  ".join(res), flush=True)
