"""MappingPipeline: single-process driver of the mapping hot path on one GPU (or one volume shard
per rank): per frame the moving TSDF volume integrates the RGB-D image (V1); every ``map_every``
frames the Mapper integrates the keyframe into the global volume (G1) and runs ``iters`` map +
``BA_iters`` pose optimisation steps of the residual field.  Plays the role of run.py's
RemixFusion + the tracker's post_processing with ground-truth poses (BASELINE config 2:
"GT poses, tracker off")."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .datasets import get_dataset
from .model.scene_rep import JointEncoding
from .model.traj import Trajectory
from .model.Volume import moving_volume
from .mp_slam.mapper import Mapper
from .mp_slam.slam import SLAM


class MappingPipeline:
    def __init__(self, config: Dict, device: str = "cuda:0", n_frames: Optional[int] = None, seed: int = 0,
                 shard=None):
        self.config, self.device = config, torch.device(device)
        self.dataset = get_dataset(config, device=device, n_frames=n_frames)
        self.tsdf_only = bool(config["synthetic"].get("tsdf_only", False))
        self.traj = Trajectory()
        self.K = self.dataset.K()
        pose0 = self.dataset.poses[0].numpy().astype(np.float64)
        self.mv = self._make_volume(config, self.traj, pose0) if shard is None \
            else shard.make_volume(config, self.traj, pose0, self.device)
        self.shard = shard
        self.slam = self.mapper = self.model = None
        if not self.tsdf_only:
            SLAM.seed_everything(None, seed)       # before the decoder / pose-MLP initialisers draw: a run is a function of `seed`
            bb = torch.from_numpy(np.array(config["mapping"]["bound"])).to(self.device)
            num_kf = int(self.dataset.num_frames // config["mapping"]["keyframe_every"] + 1)
            self.model = JointEncoding(config, bb, num_kf).to(self.device)
            with torch.no_grad():
                for name, p in self.model.named_parameters():
                    if "rba" in name:
                        torch.nn.init.normal_(p, mean=0, std=0.0001)     # run.py:39-42
            self.slam = SLAM(config, self.dataset, self.model, self.device)
            self.slam.seed_everything(seed)
            self.mapper = Mapper(config, self.slam, self.model)
            self.mapper.shard = shard
        # synthetic.tracker = True: estimate poses with the ROTracker (BASELINE config 3) instead of GT poses;
        # the tracker then owns the moving volume, like in the reference (model/ROtracker.py:84)
        self.tracker = None
        if self.slam is not None and config["synthetic"].get("tracker", False):
            from .mp_slam.tracker import Tracker
            s = self.slam
            del self.mv
            self.tracker = Tracker(config, s, self.model, self.dataset, s.est_c2w_data, s.RO_c2w_data, s.est_c2w_data_rel,
                                   s.tracking_idx, s.mapping_idx, s.tracking_stop_flag, s.pose_gt, s.update_local_MV,
                                   s.keyframeDatabase.all_fuse_pose, self.device,
                                   volume_factory=(lambda c, t, p0: self._make_volume(c, t, p0)) if shard is None else None)
            self.mv = self.tracker.RO_Tracker.MV
        # The moving volume and the residual field touch disjoint memory (the reference runs them in two processes,
        # run.py:56-60): with ground-truth poses the per-frame V1 (and the volume moves) go on their own HIP stream and
        # overlap the mapper's iterations, which are latency- rather than bandwidth-bound.  pipeline.mv_stream: False
        # keeps everything on the current stream.
        self.mv_stream = None
        if self.tracker is None and self.slam is not None and self.device.type == "cuda" and \
                config.get("pipeline", {}).get("mv_stream", True):
            self.mv_stream = torch.cuda.Stream(device=self.device)
            self.mv_stream.wait_stream(torch.cuda.current_stream(self.device))      # the volume was initialised there
            self.mv.producer_stream = self.mv_stream     # readers on other streams (get_volume_all, ...) wait for it
        # With the tracker on, the reference runs tracker (+ moving volume) and mapper as two processes (run.py:56-60,
        # mp_slam/tracker.py:173-197): the tracker's pose search reads its results back every iteration (20 host
        # synchronisations per frame), and on ONE stream each of them also waits for whatever the mapper has queued (a 7 ms
        # mapper step at scene0000 sizes), which serialises the two.  The tracker and its volume therefore get a stream of their
        # own; the mapper's stream waits for it (an event, not the host) before it reads the poses the tracker wrote.
        self.track_stream = None
        if self.tracker is not None and self.device.type == "cuda" and config.get("pipeline", {}).get("track_stream", True):
            self.track_stream = torch.cuda.Stream(device=self.device)
            self.track_stream.wait_stream(torch.cuda.current_stream(self.device))
            self.mv.producer_stream = self.track_stream
        self.frames_done = 0

    def _make_volume(self, config, traj, pose0):
        return moving_volume(config, traj, pose0, device=self.device)

    # frames are rendered once and kept resident in HBM (bench: inputs resident before the timed region)
    def prefetch(self, ids: List[int]) -> Dict[int, Dict]:
        self.dataset.prefetch(ids)
        out = {}
        for i in ids:
            b = self.dataset[i]
            b["rgb255"] = torch.floor(b["rgb"] * 255.0)
            b["c2w_dev"] = b["c2w"].to(self.device)     # a pageable H2D copy in the frame loop would drain the stream
            out[i] = b
        if self.mv_stream is not None:                  # the frames above were produced on the current stream
            self.mv_stream.wait_stream(torch.cuda.current_stream(self.device))
        if self.track_stream is not None:
            self.track_stream.wait_stream(torch.cuda.current_stream(self.device))
        return out

    def start(self, batch0: Dict, first_iters: Optional[int] = None):
        """frame 0: MV integrate + first-frame mapping (run.py:57-60 / ROtracker.py:132)."""
        self.track_frame(0, batch0)
        if self.mapper is not None:
            n = self.config["mapping"]["first_iters"] if first_iters is None else first_iters
            self.mapper.first_frame_mapping({k: v for k, v in batch0.items() if k not in ("rgb255", "c2w_dev")}, n)
        # A full (generation-2) pass of Python's cyclic collector over a torch process's ~1e6 long-lived objects takes 60-110 ms
        # (measured: tools/tracker_times.py) -- a hundred frames' worth whenever it fires inside the frame loop.  Everything alive
        # now (modules, the model, the dataset, the buffers) stays for the run: `pipeline.gc_freeze: True` collects once and moves
        # it all to the permanent generation, so that later passes only walk what the loop itself creates.  Opt-in, because it is
        # process-wide and for the life of the process: a frozen pipeline that is dropped later is never collected (its cycles
        # hold its GPU buffers) -- right for a process that runs ONE stream, wrong for one that builds pipelines in a loop (tests).
        if self.config.get("pipeline", {}).get("gc_freeze", False):
            import gc
            gc.collect()
            gc.freeze()

    def track_frame(self, i: int, batch: Dict):
        """tracker side with GT pose: follow the camera with the volume, then integrate
        (ROtracker.post_processing, model/ROtracker.py:911-945)."""
        if self.tracker is not None:
            if i == 0:
                self.slam.est_c2w_data[0] = batch["c2w"].to(self.device)
                self.slam.RO_c2w_data[0] = batch["c2w"].to(self.device)
            else:
                b = {k: v for k, v in batch.items() if k not in ("rgb255", "c2w_dev")}
                if self.track_stream is None:
                    self.tracker.tracking(b, i)
                else:
                    if batch.get("rgb255") is None:     # not prefetched: the frame was just produced on the current stream
                        self.track_stream.wait_stream(torch.cuda.current_stream(self.device))
                        # ... and was allocated there: the tracker's stream reads these blocks (compute_vertex, integrate), so
                        # the caching allocator must not hand them to new current-stream work while that is still queued
                        for v in b.values():
                            if isinstance(v, torch.Tensor) and v.is_cuda:
                                v.record_stream(self.track_stream)
                    with torch.cuda.stream(self.track_stream):
                        self.tracker.tracking(b, i)
                self.slam.tracking_idx[0] = i
            return
        c2w = batch["c2w"]
        pose_np = c2w.numpy().astype(np.float64) if not c2w.is_cuda else c2w.cpu().numpy().astype(np.float64)
        rgb255 = batch.get("rgb255")
        if self.mv_stream is None:
            self._integrate(i, batch, rgb255, pose_np)
        else:
            if rgb255 is None:                          # not prefetched: the frame was just produced on the current stream
                self.mv_stream.wait_stream(torch.cuda.current_stream(self.device))
                # ... and was allocated there: tell the caching allocator that the volume's stream reads these blocks, or it
                # may hand them to new current-stream work while the queued integrate is still reading them
                for k in ("rgb", "depth"):
                    v = batch.get(k)
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(self.mv_stream)
            with torch.cuda.stream(self.mv_stream):
                self._integrate(i, batch, rgb255, pose_np)
        if self.slam is not None:
            c2w_dev = batch.get("c2w_dev")
            if c2w_dev is None:
                c2w_dev = c2w.to(self.device)
            ke = self.config["mapping"]["keyframe_every"]
            est, rel = self.slam.est_c2w_data, self.slam.est_c2w_data_rel
            if est.is_cuda and est.dtype == torch.float32 and est.is_contiguous() and rel.is_contiguous():
                # one launch: est[i] <- c2w, and for a non-keyframe rel[i] <- c2w @ inverse(est[newest keyframe]), the pose
                # relative to the last keyframe, like the tracker stores it
                c2w_dev = c2w_dev.to(torch.float32).contiguous()
                k = (i // ke) * ke
                _lib.check(_lib.load().rfx_frame_pose(_lib.ptr(c2w_dev), _lib.ptr(est[k]) if i % ke else None, _lib.ptr(est[i]),
                                                      _lib.ptr(rel[i]) if i % ke else None, _lib.stream_ptr(self.device)), "rfx_frame_pose")
            else:
                est[i] = c2w_dev
                if i % ke != 0:
                    # inv_ex: no host-side singularity check, i.e. no device sync in the frame loop
                    rel[i] = c2w_dev @ torch.linalg.inv_ex(est[(i // ke) * ke]).inverse
            self.slam.tracking_idx[0] = i

    def _integrate(self, i, batch, rgb255, pose_np):
        if i > 0:
            self.mv.check_move_volume_new(i, pose_np, self.traj, version=self.config["volume"]["version"])
        if rgb255 is None:
            rgb255 = torch.floor(batch["rgb"] * 255.0)
        self.mv.integrate(rgb255, batch["depth"], self.K, pose_np, self.mv.vol_bnds)

    def sync_volume(self):
        """make the current stream wait for the moving volume's stream (before reading the volume from it)"""
        if self.mv_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.mv_stream)
        if self.track_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.track_stream)

    def step(self, i: int, batch: Dict):
        """one frame of the stream: V1 always; mapper step when the reference's loop would fire."""
        self.track_frame(i, batch)
        if self.mapper is not None:
            m = self.config["mapping"]
            cur = int(self.slam.mapping_idx[0] + m["keyframe_every"])
            # the reference's mapper wakes when tracking_idx > mapping_idx + map_every (mapper.py:879)
            if i > int(self.slam.mapping_idx[0]) + m["map_every"] and cur < len(self.dataset):
                if self.track_stream is not None:       # the poses up to frame i are written on the tracker's stream
                    torch.cuda.current_stream(self.device).wait_stream(self.track_stream)
                self.mapper.step(cur)
        self.frames_done += 1
