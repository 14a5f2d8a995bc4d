"""dev helper: how far the decoder weight gradients of a bench-size map iteration are from the float64 oracle, in units of
the oracle's own fp32 noise (the yardstick of tests/test_field_gpu.py::_grad_close), over several ray batches."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_timed_path_gpu as T
from test_field_gpu import _f64_params, _oracle_params
from remixfusion_amd import _lib as L
lib = L.load()
cfg, pipe, fr = T._pipeline("office0", 41)
mp, model, slam = pipe.mapper, pipe.model, pipe.slam
direct = mp._direct_iterations(); direct.stagewise_every = 0
tr, m = cfg["training"], cfg["mapping"]
S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
enc = model.embed_res_fn
nF, nL = enc.n_output_dims, int(enc.desc.n_levels)
last = 40
b = fr[last]
cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
n_kf = len(mp.keyframe.frame_ids)
n = int(m["sample"]) + max(int(m["sample"]) // n_kf, int(m["min_pixels_cur"]))
bbox = model.bounding_box.cpu()
dev = cur.device
params = [enc.params] + list(model.decoder_res.fused_weights())
poses = slam.est_c2w_data[0:last + 1:m["keyframe_every"]].clone().float().contiguous()
poses_all = torch.cat([poses, slam.est_c2w_data[last:last + 1].float()], 0)[:n_kf + 1].contiguous()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    random.seed(100 + rep); torch.manual_seed(100 + rep)
    for p_ in params: p_.grad = None
    fp = _oracle_params(cfg, model)
    fill = os.environ.get("WS_FILL")
    if fill:                                    # poison the workspace: a stage that reads what no stage of this iteration wrote shows up
        B0 = direct._buffers(n, 0, dev)
        if fill == "nan": B0.t.ws.fill_(float("nan"))
        elif fill == "big": B0.t.ws.fill_(3.0e30)
        else: B0.t.ws.uniform_(-1e3, 1e3)
        model._ws_buf = None
    direct.map_gradients(cur, poses_all)
    torch.cuda.synchronize()
    got = [p_.grad.detach().clone() for p_ in params]
    B = direct._buffers(n, 0, dev)
    f = {k: v.cpu() for k, v in T._ws_fields(lib, B, n, S, P, nF, nL).items()}
    for t in (fp.hash_table, fp.W1, fp.W2, fp.W3, fp.W4): t.requires_grad_(True)
    _, _, total = T._oracle_iteration(fp, cfg, bbox, f["o"], f["d"], f["z"], f["tgt"], f["td"], False, f["pts"])
    total.backward()
    fq = _f64_params(fp)
    _, _, total64 = T._oracle_iteration(fq, cfg, bbox, f["o"], f["d"], f["z"], f["tgt"], f["td"], False, f["pts"])
    total64.backward()
    line = []
    for g, a, q, nm in zip(got[1:], (fp.W1, fp.W2, fp.W3, fp.W4), (fq.W1, fq.W2, fq.W3, fq.W4), ("dW1", "dW2", "dW3", "dW4")):
        g64 = q.grad.double().reshape(-1); g32 = a.grad.double().reshape(-1); gg = g.cpu().double().reshape(-1)
        noise = float((g32 - g64).abs().max())
        err = (gg - g64).abs()
        lim = 1e-4 * g64.abs() + 4 * noise
        worst = float((err / lim).max())
        line.append(f"{nm}: noise {noise:.2e} max|g| {float(g64.abs().max()):.2e} max err {float(err.max()):.2e} (x{float(err.max())/noise:5.1f} noise) worst err/lim {worst:5.2f}")
    print(f"rep {rep}: " + " | ".join(line), flush=True)
