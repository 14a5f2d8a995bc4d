"""dev helper: is a hipGraph replay of one mapping iteration cheaper than running it eagerly?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.optim as optim
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.pipeline import MappingPipeline
cfg = synthetic_config("office0"); cfg["mapping"]["first_iters"] = 20
pipe = MappingPipeline(cfg, n_frames=60)
frames = pipe.prefetch(list(range(40)))
pipe.start(frames[0])
for i in range(1, 31): pipe.step(i, frames[i])
mp, slam, model = pipe.mapper, pipe.slam, pipe.model
m = cfg["mapping"]
trainable = [{"params": model.decoder_res.parameters(), "weight_decay": 1e-6, "lr": m["lr_decoder"]},
             {"params": model.embed_res_fn.parameters(), "eps": 1e-15, "lr": m["lr_embed_res"]}]
opt = optim.Adam(trainable, betas=(0.9, 0.99), capturable=True)
ropt = optim.Adam([{"params": model.rba.parameters(), "weight_decay": 1e-6, "eps": 1e-15, "lr": m["lr_pose"]}], betas=(0.9, 0.99), capturable=True)
cur = 30
batch = {k: (v[None, ...] if isinstance(v, torch.Tensor) else torch.tensor([v])) for k, v in frames[cur].items() if k != "rgb255"}
current_rays = torch.cat([batch["direction"], batch["rgb"], batch["depth"][..., None]], dim=-1).reshape(-1, 7).cuda()
poses = mp.est_c2w_data[0:cur + 1:m["keyframe_every"]].clone()
all_index = torch.arange(0, poses.shape[0], device="cuda").unsqueeze(-1)

def it_map():
    rays, ids_all = mp._sample_rays(current_rays)
    ro, rd, ts, td = mp._world_rays(rays, ids_all, poses)
    ret = model.mapping(ro, rd, ts, td)
    loss = slam.get_loss_from_ret(ret, smooth=True)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)

def it_pose():
    poses_all = model.rba(all_index)
    rays, ids_all = mp._sample_rays(current_rays)
    ro, rd, ts, td = mp._world_rays(rays, ids_all, poses_all)
    ret = model.mapping(ro, rd, ts, td, clamp=True)
    loss = slam.get_loss_from_ret(ret, smooth=True)
    loss.backward()
    ropt.step()
    ropt.zero_grad(set_to_none=True); opt.zero_grad(set_to_none=True)

def timeit(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3

for name, f in (("map", it_map), ("pose", it_pose)):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): f()
    torch.cuda.current_stream().wait_stream(s)
    print(name, "eager cpu/wall ms", timeit(f))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(name, "capture+instantiate ms", (t1 - t0) * 1e3)
    print(name, "replay cpu/wall ms", timeit(g.replay))
    t0 = time.perf_counter()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        f()
    torch.cuda.synchronize(); print(name, "2nd capture ms", (time.perf_counter() - t0) * 1e3)
    del g, g2
