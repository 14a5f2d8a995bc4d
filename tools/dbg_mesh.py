import sys, torch, numpy as np
sys.path.insert(0, '.')
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
from remixfusion_amd import mesh as M
cfg = synthetic_config("office0")
cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
cfg["mapping"].update({"first_iters": 50, "sample": 512})
cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
cfg["data"].update({"output": "/tmp/m", "exp_name": "t"})
pipe = MappingPipeline(cfg, n_frames=20, seed=1)
frames = pipe.prefetch(list(range(11)))
pipe.track_frame(0, frames[0])
pipe.mapper.init_mapvolume()
pipe.mapper.first_frame_mapping({k: v for k, v in frames[0].items() if k != "rgb255"}, cfg["mapping"]["first_iters"])
for i in range(1, 11):
    pipe.step(i, frames[i])
s = pipe.slam
print("bound", cfg["mapping"]["bound"], "mcb", cfg["mapping"]["marching_cubes_bound"], "sc", cfg["data"]["sc_factor"], cfg["data"]["translation"])
gbw = pipe.model.GBW.params; gbv = pipe.model.GBV.params.view(-1, 4)
print("GBW>0", int((gbw > 0).sum()), "of", gbw.numel(), "GBV tsdf<0", int((gbv[:, 0] < 0).sum()), "tsdf range", float(gbv[:,0].min()), float(gbv[:,0].max()))
mcb = s.marching_cube_bound
tx, ty, tz = M.get_voxels(mcb[0, 1], mcb[0, 0], mcb[1, 1], mcb[1, 0], mcb[2, 1], mcb[2, 0], 0.08, None)
pts = torch.stack(torch.meshgrid(tx, ty, tz, indexing="ij"), -1).float().cuda()
bb = s.bounding_box
flat = (pts.reshape(-1, 3) - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
print(flat.dtype, pts.shape)
sdf = pipe.model.query_sdf_res(flat[:, None, :]).reshape(pts.shape[:-1])
sdfx = pipe.model.query_sdf_ex(flat[:, None, :]).reshape(pts.shape[:-1])
w = pipe.model.query_w_res(flat[:, None, :]).reshape(pts.shape[:-1])
print("w>0", int((w > 0).sum()), "sdf<0", int((sdf < 0).sum()), "sdf_ex<0", int((sdfx < 0).sum()), "both", int(((sdf < 0) & (w > 0)).sum()))
print("sdf range", float(sdf.min()), float(sdf.max()), float(sdfx.min()), float(sdfx.max()))
for nm, f in (("res", sdf), ("ex", sdfx)):
    v, fc = M.marching_cubes(f, 0.0, mask=w > 0)
    v2, fc2 = M.marching_cubes(f, 0.0)
    print(nm, "faces masked", fc.shape[0], "unmasked", fc2.shape[0])
