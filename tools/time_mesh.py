"""dev helper: where a per-keyframe mesh export goes (apartment sizes): each step of SLAM._save_mesh timed with a sync after it"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
from remixfusion_amd import mesh as M
name = sys.argv[1] if len(sys.argv) > 1 else "apartment"
cfg = synthetic_config(name); cfg["mapping"]["first_iters"] = 50; cfg["mesh"]["only_final"] = 1
pipe = MappingPipeline(cfg, n_frames=30)
fr = pipe.prefetch(list(range(21)))
pipe.start(fr[0])
for i in range(1, 21): pipe.step(i, fr[i])
torch.cuda.synchronize()
slam, model = pipe.slam, pipe.model
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print(f"  {label:34s} {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True); return r
for rep in range(2):
    print("rep", rep)
    bb, mcb, dev = slam.bounding_box, slam.marching_cube_bound, slam.bounding_box.device
    axes = T("get_voxels (cpu linspace)", lambda: M.get_voxels(mcb[0, 1], mcb[0, 0], mcb[1, 1], mcb[1, 0], mcb[2, 1], mcb[2, 0], 0.1, None))
    pts = T("meshgrid + to(device)", lambda: torch.stack(torch.meshgrid(*axes, indexing="ij"), -1).to(torch.float32).to(dev))
    sh = pts.shape; print("  grid", tuple(sh))
    flat = T("normalise", lambda: (pts.reshape(-1, 3) - bb[:, 0]) / (bb[:, 1] - bb[:, 0]))
    sdf = T("query_sdf_res", lambda: model.query_sdf_res(flat[:, None, :]).reshape(sh[:-1]).to(torch.float32))
    w = T("query_w_res", lambda: model.query_w_res(flat[:, None, :]).reshape(sh[:-1]))
    vf = T("marching_cubes", lambda: M.marching_cubes(sdf, 0.0, mask=w > 0))
    print("  verts", tuple(vf[0].shape), "faces", tuple(vf[1].shape))
    mesh = T("extract_mesh (whole)", lambda: M.extract_mesh(model.query_sdf_res, model.query_w_res, cfg, bb, color_func=model.query_color_residual, marching_cube_bound=mcb, voxel_size=0.1))
    T("write_ply", lambda: M.write_ply("/tmp/m.ply", mesh))
