"""Tracker: per-frame pose tracking loop (reference mp_slam/tracker.py:15-197): constant-velocity
prediction, ROTracker pose search against the moving TSDF volume, relative-pose bookkeeping, volume
update.  ``track_frame`` is the body of the reference's ``tracking()``; ``run()`` keeps the loop."""
from __future__ import annotations

import time

import numpy as np
import torch

from ..model.ROtracker import ROTracker


def orthogonalize_rotation(R: np.ndarray, epsilon: float = 1e-10) -> np.ndarray:
    """nearest orthogonal matrix by SVD, in the matrix's own precision, entries within epsilon of +-1 snapped: the reference's
    ``orthogonalize_rotation_matrix_tolerate`` (model/utils.py:63-70)."""
    U, _, V = np.linalg.svd(R)
    out = np.dot(U, V)
    out[np.abs(out - 1) < epsilon] = 1
    out[np.abs(out + 1) < epsilon] = -1
    return out


def constant_velocity(pp: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """pose after ``p`` when the motion pp -> p repeats (reference mp_slam/tracker.py:62-70): float32 on the host throughout,
    ``p @ inverse(pp)`` then ``@ p``, rotation re-orthogonalised.  pp, p: [4,4] float32 CPU tensors."""
    delta = p @ torch.inverse(pp).float()
    pred = delta @ p
    pred[:3, :3] = torch.from_numpy(orthogonalize_rotation(pred[:3, :3].numpy()))
    return pred


class Tracker:
    def __init__(self, config, SLAM, model, dataset, est_c2w_data, RO_c2w_data, est_c2w_data_rel, tracking_idx, mapping_idx,
                 tracking_stop_flag, pose_gt, update_local_MV, all_fuse_pose, device, volume_factory=None) -> None:
        self.config, self.slam, self.dataset = config, SLAM, dataset
        self.est_c2w_data, self.RO_c2w_data, self.est_c2w_data_rel = est_c2w_data, RO_c2w_data, est_c2w_data_rel
        self.tracking_idx, self.mapping_idx, self.tracking_stop_flag = tracking_idx, mapping_idx, tracking_stop_flag
        self.update_local_MV, self.all_fuse_pose, self.share_model, self.pose_gt = update_local_MV, all_fuse_pose, model, pose_gt
        self.frames_num = len(dataset)
        self.device = device
        self.RO_Tracker = ROTracker(config, dataset, device=device, volume_factory=volume_factory)
        self.all_poses = []
        self._key_inv, self._ro_key = (-1, None), (-1, None)
        self._ro_host = {}              # frame id -> the tracker's own result as it left the search (numpy, host): what the
                                        # constant-velocity prediction reads, without a device->host copy (and its wait) per frame

    def predict_current_pose(self, frame_id, constant_speed=True):
        """reference :55-72."""
        if frame_id == 1 or (not constant_speed):
            self.est_c2w_data[frame_id] = self.est_c2w_data[frame_id - 1]
        else:
            pp, p = (torch.from_numpy(self._ro_host[f]) if f in self._ro_host else self.RO_c2w_data[f].cpu().float()
                     for f in (frame_id - 2, frame_id - 1))
            self._pred_host = constant_velocity(pp, p).numpy()
            self.est_c2w_data[frame_id] = torch.from_numpy(self._pred_host).to(self.device)
            return self._pred_host                    # (the search starts from a host array: no read-back of what was just sent)
        return self.est_c2w_data[frame_id]

    def tracking(self, batch, frame_id):
        """reference :74-134."""
        cur_c2w = self.predict_current_pose(frame_id, self.config["tracking"]["const_speed"])
        RO_pose_np, rgb, depth = self.RO_Tracker.do_tracking(cur_c2w, None, batch, self.device)
        self.RO_Tracker.RO_pose.append(RO_pose_np)
        cur = torch.from_numpy(RO_pose_np).float().to(self.device)
        self.est_c2w_data[frame_id] = cur
        self.RO_c2w_data[frame_id] = cur
        self._ro_host[frame_id] = np.asarray(RO_pose_np, np.float32)
        self._ro_host.pop(frame_id - 3, None)
        self.all_poses.append(torch.from_numpy(RO_pose_np).float())
        ke = self.config["mapping"]["keyframe_every"]
        if frame_id % ke != 0:
            # pose relative to the frame's keyframe: the key pose is inverted in float32 on the HOST, as the reference does
            # (`torch.inverse(c2w_key.cpu())`, :112-116) -- once per keyframe here, its device copy re-used by the frames after it
            kf = (frame_id // ke) * ke
            if self._key_inv[0] != kf:
                key = torch.from_numpy(self._ro_key[1]) if self._ro_key[0] == kf else self.RO_c2w_data[kf].cpu().float()
                self._key_inv = (kf, torch.inverse(key).float().to(self.device))
            self.est_c2w_data_rel[frame_id] = cur @ self._key_inv[1]
        else:
            self._ro_key = (frame_id, np.asarray(RO_pose_np, np.float32).copy())
        self.RO_Tracker.post_processing(frame_id, RO_pose_np, rgb, depth, self.est_c2w_data)

    def run(self):
        """reference :173-197 (waits for the mapper when it lags)."""
        m = self.config["mapping"]
        for idx in range(self.frames_num):
            batch = self.dataset[idx]
            if idx == 0:
                self.all_poses.append(self.est_c2w_data[0].detach().cpu())
                self.RO_c2w_data[0] = self.est_c2w_data[0].detach().clone()
                continue
            while self.mapping_idx[0] < idx - m["map_every"] - m["map_every"] // 2:
                time.sleep(0.02)
            self.tracking(batch, idx)
            self.tracking_idx[0] = idx
        self.tracking_stop_flag[0] = 1
