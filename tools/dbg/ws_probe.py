import sys, ctypes as C, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import test_timed_path_gpu as T
from remixfusion_amd import _lib as L
lib = L.load()
cfg, pipe, fr = T._pipeline("office0", 16)
mp, model, slam = pipe.mapper, pipe.model, pipe.slam
direct = mp._direct_iterations(); direct.stagewise_every = 0
tr, m = cfg["training"], cfg["mapping"]
S, P = int(tr["n_range_d"]) + int(tr["n_samples_d"]), int(tr["smooth_pts"]) - 1
enc = model.embed_res_fn
last = 15
b = fr[last]
cur = torch.cat([b["direction"], b["rgb"], b["depth"][..., None]], dim=-1).reshape(-1, 7).contiguous()
n_kf = len(mp.keyframe.frame_ids)
n = int(m["sample"]) + max(int(m["sample"]) // n_kf, int(m["min_pixels_cur"]))
poses_all = slam.est_c2w_data[0:last + 1:5].clone().float().contiguous()
print("n_kf", n_kf, "n", n, "poses", poses_all.shape, direct._n_rays())
lc = direct.map_gradients(cur, poses_all)
torch.cuda.synchronize()
B = direct._buffers(n, 0, poses_all.device)
print("B.n", B.n, B.cap_n, B.ws_bytes, B.p.ws - B.t.ws.data_ptr())
off = (C.c_size_t * 15)()
print(lib.rfx_ba_workspace_layout(n, S, P, enc.n_output_dims, int(enc.desc.n_levels), off, 15), list(off))
f = T._ws_fields(lib, B, n, S, P, enc.n_output_dims, int(enc.desc.n_levels))
for k, v in f.items():
    print(k, tuple(v.shape), v.reshape(-1)[:6].tolist(), float(v.float().abs().max()))
print("lc", lc.tolist())
