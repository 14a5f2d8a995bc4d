"""One-rank RCCL rehearsal (run by tests/test_rccl_one_rank_gpu.py in a process of its own, with RFX_DIST_FORCE_COLLECTIVES=1):
a 1-GPU box cannot hold two RCCL ranks, but a one-rank communicator executes the same torch.distributed calls on the same
device tensors (views of larger buffers, uneven split lists, MAX reductions, all_gather lists, stream ordering), so this runs the
one-scene pipeline's collectives through backend "nccl" for real and compares with the plain single-process pipeline.
Prints one JSON line."""
import json, os, sys, random, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
assert os.environ.get("RFX_DIST_FORCE_COLLECTIVES") == "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from remixfusion_amd.config import synthetic_config
from remixfusion_amd.dist import ShardedPipeline
from remixfusion_amd.pipeline import MappingPipeline

out = {"backend": dist.get_backend()}
N = 16
for mode in ("levels", "replicas"):
    losses = {}
    for sharded in (True, False):
        random.seed(0)
        cfg = synthetic_config("scene0000")
        cfg["synthetic"].update({"tracker": True, "depth_noise": 0.0, "dropout": 0.0, "clutter": 48})
        cfg["mapping"].update({"first_iters": 50, "shard_field": mode})
        cfg["mesh"]["async_export"] = False
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pipe = (ShardedPipeline(cfg, dist, 0, 1, device="cuda:0", n_frames=128, seed=0) if sharded
                    else MappingPipeline(cfg, device="cuda:0", n_frames=128, seed=0))      # (128: the camera path of the ATE runs)
        frames = pipe.prefetch(list(range(N)))
        pipe.start(frames[0])
        for i in range(1, N):
            pipe.step(i, frames[i])
        torch.cuda.synchronize()
        if sharded:
            pipe.mapper.sync_field()
            out[f"{mode}_iterations"] = type(pipe.mapper._direct_iterations()).__name__
            # cold readers of the sharded volume through the communicator
            t, w, c = pipe.mv.gather_whole()
            out[f"{mode}_volume_voxels"] = int(t.size)
            pts = torch.rand(2000, 3, device="cuda:0") * 2 - 1
            tri = pipe.mv.tri_interpolate(pts)
            out[f"{mode}_trilerp_rows"] = int(np.asarray(tri[0] if isinstance(tri, (tuple, list)) else tri).shape[0])
        losses[sharded] = [float(v) for v in pipe.mapper.last_losses.values()] if hasattr(pipe.mapper, "last_losses") else None
        pose = pipe.slam.est_c2w_data[N - 1].cpu().numpy()
        out[f"{mode}_{'sharded' if sharded else 'single'}_pose_t"] = [float(v) for v in pose[:3, 3]]
        # the tracker's own trajectory (what its search produced, before the mapper's pose refinement moves the keyframes)
        ro = pipe.slam.RO_c2w_data[1:N].cpu().numpy()
        out[f"{mode}_{'sharded' if sharded else 'single'}_ro_hex"] = ro.astype(np.float32).tobytes().hex()
        gt = frames[N - 1]["c2w"].numpy()
        out[f"{mode}_{'sharded' if sharded else 'single'}_pose_err_cm"] = float(np.linalg.norm(pose[:3, 3] - gt[:3, 3]) * 100)
        del pipe
        torch.cuda.empty_cache()
dist.barrier()
dist.destroy_process_group()
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "rccl_one_rank.json"), "w").write(json.dumps(out, indent=1))
print(json.dumps(out), flush=True)
