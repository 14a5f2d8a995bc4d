"""moving_volume: the local TSDF volume that follows the camera, on MI355X.

Host-side mirror of the reference class ``model/Volume.py:19-1408`` (same constructor, method
names, argument meaning, attributes) with the PyCUDA kernels replaced by librfx (HIP).  State
lives in HBM as three flat fp32 torch tensors (+ three back buffers), z fastest.

Differences that are deliberate and documented in DESIGN.md:
  * ``integrate`` accepts numpy arrays (reference behaviour: synchronous upload) *or* CUDA
    tensors (no copy; the bench path).
  * no CPU mode: without a GPU / librfx the constructor raises (the reference crashes with an
    AttributeError in the same situation, model/Volume.py:613).
  * a volume whose dimensions would grow past the allocation raises instead of overflowing
    (latent bug in the reference, model/Volume.py:94-107,816).
Meshing / ply writers (reference :1280-1408) are CPU debug I/O and out of scope.
"""
from __future__ import annotations

import copy
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import _F3, _F6, _F9, _F16, check, farr, ptr, stream_ptr

_AXES = {"x": 0, "y": 1, "z": 2}


class moving_volume:
    """Moving volume of RGB-D images (reference: model/Volume.py:19)."""

    def __init__(self, cfg, traj, init_pose, gpu_mode=True, start=0, device: Optional[torch.device] = None):
        vol = cfg["volume"]
        self.config = cfg
        self.voxel_size = float(vol["voxel_size"])
        self.surface_trunc = cfg["training"]["trunc"]
        self.trunc_margin = vol["trunc"]
        self.first_len, self.second_len, self.third_len = vol["first_len"], vol["second_len"], vol["third_len"]
        self.more_angel_t = vol["more_angel_t"]
        self.fix_x, self.fix_y, self.fix_z = (vol[k]["fix"] for k in ("x_config", "y_config", "z_config"))
        self.x_len, self.y_len, self.z_len = (vol[k]["len"] for k in ("x_config", "y_config", "z_config"))
        self.x_range, self.y_range, self.z_range = (vol[k]["range"] for k in ("x_config", "y_config", "z_config"))
        self.version = vol["version"]
        self.t_treshold = vol["t_treshold"]
        self.cut = cfg.get("RO", {}).get("cut", 0)
        self.cut_dist = cfg.get("RO", {}).get("cut_dist", 8.0)
        self.weight_clamp = vol["weight_clamp"]
        self.index_decode = 0 if vol.get("index_decode", "reference") == "reference" else 1
        self.last_pcid = 0
        self.surface_pc = None
        self.start_id = 0
        self.frame_to_Vrange = {}
        self.color_const = 256 * 256

        # reference :57-71
        self.vol_bnds = np.asarray(self.initialize_vol_bnd(init_pose, traj, self.version))
        assert self.vol_bnds.shape == (3, 2), "[!] `vol_bnds` should be of shape (3, 2)."
        self._set_geometry(self.vol_bnds)

        if not gpu_mode:
            raise _lib.RfxError("moving_volume has no CPU mode (the reference has none either)")
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.RfxError("moving_volume needs a HIP device")
        self.gpu_mode = True
        self.device = torch.device(device if device is not None else "cuda:0")
        n = self._alloc_voxels()
        self._capacity = n
        dev = self.device
        # tsdf=1, weight=0, colour=0 (reference :85-107); sized for 288 GB HBM: front + back resident
        self.tsdf_vol_gpu = torch.ones(n, dtype=torch.float32, device=dev)
        self.weight_vol_gpu = torch.zeros(n, dtype=torch.float32, device=dev)
        self.color_vol_gpu = torch.zeros(n, dtype=torch.float32, device=dev)
        self.tsdf_vol_gpu_back = torch.ones(n, dtype=torch.float32, device=dev)
        self.weight_vol_gpu_back = torch.zeros(n, dtype=torch.float32, device=dev)
        self.color_vol_gpu_back = torch.zeros(n, dtype=torch.float32, device=dev)
        self._ws = None
        self._ws_hw = None

    # ------------------------------------------------------------------ geometry helpers
    def _set_geometry(self, bnds: np.ndarray) -> None:
        """reference :67-71 / :692-694 / :813-818 (dims = ceil(extent/voxel), snap upper bound)."""
        self.vol_bnds = bnds
        self.vol_dim = np.ceil((bnds[:, 1] - bnds[:, 0]) / self.voxel_size).copy(order="C").astype(int)
        self.vol_bnds[:, 1] = self.vol_bnds[:, 0] + self.vol_dim * self.voxel_size
        self.vol_origin = self.vol_bnds[:, 0].copy(order="C").astype(np.float32)
        if hasattr(self, "_capacity") and self._n() > self._capacity:
            raise _lib.RfxError("volume grew past its allocation (reference would overflow here)")

    def _can_shift_in_place_of_copy(self) -> bool:
        """copy_volume() followed at once by the re-gridding gather may be done as one gather between the two buffer sets
        (update_tsdf_swap_rot_trans(source="front")): for the plain single-GPU volume with both sets allocated."""
        return type(self).update_tsdf_swap_rot_trans is moving_volume.update_tsdf_swap_rot_trans and \
            "update_tsdf_swap_rot_trans" not in self.__dict__ and "copy_volume" not in self.__dict__ and \
            getattr(self, "tsdf_vol_gpu_back", None) is not None and getattr(self, "tsdf_vol_gpu", None) is not None and \
            self.tsdf_vol_gpu_back.numel() == self.tsdf_vol_gpu.numel()

    def _vols(self):
        return self.tsdf_vol_gpu, self.weight_vol_gpu, self.color_vol_gpu

    def _backs(self):
        return self.tsdf_vol_gpu_back, self.weight_vol_gpu_back, self.color_vol_gpu_back

    def _n(self) -> int:
        return int(np.prod(self.vol_dim))

    def _alloc_voxels(self) -> int:
        """voxels per array to allocate (dist.sharded_volume: one x-slab)"""
        return int(np.prod(self.vol_dim))

    def _slab(self) -> Tuple[int, int]:
        """x-planes this object holds: all of them (dist.sharded_volume: one slab)"""
        return 0, int(self.vol_dim[0])

    def _workspace(self, H: int, W: int) -> torch.Tensor:
        d = tuple(int(v) for v in self.vol_dim)
        x0, x1 = self._slab()
        key = (H, W, x1 - x0) + d[1:]
        if self._ws_hw != key:
            nbytes = _lib.load().rfx_tsdf_integrate_workspace_bytes(x1 - x0, d[1], d[2], H, W)
            self._ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            self._ws_hw = key
        return self._ws

    def _dev(self, a, dtype=torch.float32) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    # ------------------------------------------------------------------ kernels
    def integrate(self, color_im, depth_im, cam_intr, cam_pose, old_bnd, obs_weight=1., reintegrate_flag=0.0,
                  color_packed: Optional[torch.Tensor] = None):
        """Integrate an RGB-D frame (reference :713-757, kernel :196-336).

        color_im (H,W,3) 0..255, depth_im (H,W) metres, cam_intr (3,3), cam_pose (4,4) c2w,
        old_bnd (3,2) or 6 values (only read when reintegrate_flag==1).  Fire-and-forget.
        """
        lib = _lib.load()
        H, W = depth_im.shape[-2:]
        ws = self._workspace(H, W)
        st = stream_ptr(self.device)
        depth = self._dev(depth_im).reshape(-1)
        K = np.asarray(cam_intr.detach().cpu() if isinstance(cam_intr, torch.Tensor) else cam_intr, np.float32).reshape(-1)
        c2w = np.asarray(cam_pose.detach().cpu() if isinstance(cam_pose, torch.Tensor) else cam_pose, np.float32).reshape(-1)
        ob = np.zeros(6, np.float32) if old_bnd is None else np.asarray(old_bnd, np.float32).reshape(-1)
        d = self.vol_dim
        x0, x1 = self._slab()
        # the colour packing of the reference's host code (:725-728) happens inside the call (rfx_tsdf_integrate_rgb),
        # unless the caller brings an already packed image
        if color_packed is None:
            img, fn, what = self._dev(color_im).reshape(-1, 3), lib.rfx_tsdf_integrate_rgb, "rfx_tsdf_integrate_rgb"
        else:
            img, fn, what = color_packed, lib.rfx_tsdf_integrate_slab, "rfx_tsdf_integrate_slab"
        check(fn(ptr(self.tsdf_vol_gpu), ptr(self.weight_vol_gpu), ptr(self.color_vol_gpu),
                 int(d[0]), int(d[1]), int(d[2]), x0, x1, farr(_F3, self.vol_origin), self.voxel_size,
                 farr(_F9, K), farr(_F16, c2w), ptr(img), ptr(depth), H, W,
                 float(self.trunc_margin), float(obs_weight), int(self.weight_clamp == 1.0),
                 int(reintegrate_flag == 1.0), farr(_F6, ob), self.index_decode,
                 ptr(ws), ws.numel() * 4, st), what)

    def clean_volume(self):
        """reference :656-677 (kernel :561-583)."""
        t, w, c = self._vols()
        check(_lib.load().rfx_tsdf_fill(ptr(t), ptr(w), ptr(c), self._n(), stream_ptr(self.device)), "rfx_tsdf_fill")

    def update_tsdf_swap_clean(self, vol_bnds, old_bnds):
        """reference :679-711: adopt new bounds, then clean."""
        self._set_geometry(vol_bnds)
        self.clean_volume()

    def copy_volume(self):
        """front -> back buffers (reference :883-908, kernel :585-610)."""
        t, w, c = self._vols()
        tb, wb, cb = self._backs()
        check(_lib.load().rfx_tsdf_copy(ptr(t), ptr(w), ptr(c), ptr(tb), ptr(wb), ptr(cb), self._n(),
                                        stream_ptr(self.device)), "rfx_tsdf_copy")

    def update_tsdf_swap_rot_trans(self, vol_bnds, old_bnds, source="back"):
        """Re-grid the volume into new bounds, gathering from the back copy
        (reference :796-855, kernel :128-194).

        source="front": the state copy_volume() + this call leave -- front = re-gridded volume, back = the volume as it
        was -- without the copy: the gather reads the front buffers, writes the back ones (every voxel of the new grid is
        written), and the two sets trade places.  Saves one 2 x 12 B/voxel sweep per move (1.9 ms at 800x800x600)."""
        self._set_geometry(vol_bnds)
        old_origin = old_bnds[:, 0].copy(order="C").astype(np.float32)
        old_dim = np.ceil((old_bnds[:, 1] - old_bnds[:, 0]) / self.voxel_size).copy(order="C").astype(int)
        if source == "front":
            (self.tsdf_vol_gpu, self.weight_vol_gpu, self.color_vol_gpu,
             self.tsdf_vol_gpu_back, self.weight_vol_gpu_back, self.color_vol_gpu_back) = (*self._backs(), *self._vols())
        t, w, c = self._vols()
        tb, wb, cb = self._backs()
        d = self.vol_dim
        check(_lib.load().rfx_tsdf_shift(ptr(t), ptr(w), ptr(c), int(d[0]), int(d[1]), int(d[2]),
                                         farr(_F3, self.vol_origin), ptr(tb), ptr(wb), ptr(cb),
                                         int(old_dim[0]), int(old_dim[1]), int(old_dim[2]), farr(_F3, old_origin),
                                         self.voxel_size, self.index_decode, stream_ptr(self.device)), "rfx_tsdf_shift")

    # A driver may run integrate / the volume moves on a stream of their own (MappingPipeline does, to overlap the mapper):
    # it names that stream here, and the methods that READ the volume on the current stream wait for it first.
    producer_stream = None

    def _wait_for_producer(self):
        if self.producer_stream is not None and self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            if cur != self.producer_stream:
                cur.wait_stream(self.producer_stream)

    def tri_interpolate(self, query_pc):
        """Trilinear tsdf/rgb at world points (reference :760-794, kernel :337-458).
        Returns (result [N,5], mask [N]) as numpy, like the reference."""
        self._wait_for_producer()
        pts = self._dev(query_pc).reshape(-1, 3)
        out = torch.empty((pts.shape[0], 5), dtype=torch.float32, device=self.device)
        t, w, c = self._vols()
        d = self.vol_dim
        check(_lib.load().rfx_tsdf_trilerp(ptr(t), ptr(w), ptr(c), int(d[0]), int(d[1]), int(d[2]),
                                           farr(_F3, self.vol_origin), self.voxel_size, ptr(pts), pts.shape[0],
                                           ptr(out), stream_ptr(self.device)), "rfx_tsdf_trilerp")
        result = out.cpu().numpy()
        notvalid = (result[:, 0] == 10.0) & (result[:, 1] == 0.0) & (result[:, 2] == 0.0) & (result[:, 3] == 0.0)
        return result, ~notvalid

    def filter_tsdf(self, weight_threshold):
        """reference :857-881 (kernel :462-487)."""
        t, w, c = self._vols()
        check(_lib.load().rfx_tsdf_filter(ptr(t), ptr(w), ptr(c), self._n(), float(weight_threshold),
                                          stream_ptr(self.device)), "rfx_tsdf_filter")

    def get_truncated_pc(self, pc_num=5000000, trunc_tsdf=0.5):
        """Near-surface voxels as a point cloud (reference :622-653, kernel :489-559)."""
        self._wait_for_producer()
        pc = torch.zeros((pc_num, 7), dtype=torch.float32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        d = self.vol_dim
        check(_lib.load().rfx_tsdf_truncated_pc(ptr(self.tsdf_vol_gpu), ptr(self.color_vol_gpu), int(d[0]), int(d[1]),
                                                int(d[2]), farr(_F3, self.vol_origin), self.voxel_size,
                                                float(self.trunc_margin), int(pc_num), float(trunc_tsdf), ptr(pc),
                                                cnt.data_ptr(), self.index_decode, stream_ptr(self.device)),
              "rfx_tsdf_truncated_pc")
        truncated_pc = pc.cpu().numpy()
        valid = (truncated_pc[:, 0] != 0.0) & (truncated_pc[:, 1] != 0.0) & (truncated_pc[:, 2] != 0.0)
        return truncated_pc[valid, :]

    def track_evaluate(self, vertex4, normal3, R, T, cand, search_size, n_cand, K9, H, W, level, level_index, value, count):
        """nearest-voxel TSDF residuals of the tracker's pose candidates against THIS volume (reference kernel
        model/ROtracker.py:144-270): value (sum of |tsdf - target| in 2^-30 units) / count dev int64 [n_cand].  A sharded
        volume overrides it (its slab + an exact integer sum over ranks)."""
        d = self.vol_dim
        check(_lib.load().rfx_track_evaluate(ptr(self.tsdf_vol_gpu), int(d[0]), int(d[1]), int(d[2]), farr(_F3, self.vol_origin),
                                             float(self.voxel_size), ptr(vertex4), ptr(normal3), farr(_F9, R), farr(_F3, T), ptr(cand),
                                             farr(_F6, search_size), int(n_cand), farr(_F9, K9), int(H), int(W), int(level),
                                             int(level_index), ptr(value), ptr(count), stream_ptr(self.device)), "rfx_track_evaluate")

    # the device-side search (rfx_track_search_*): what it reads of this volume; a sharded volume overrides both
    track_search_reduce = None                         # callable(sums int64 [2, rows]) adding the evaluation's sums over ranks, or None

    def track_search_volume(self):
        d = self.vol_dim
        return {"tsdf": self.tsdf_vol_gpu, "dim": (int(d[0]), int(d[1]), int(d[2])), "slab": (0, int(d[0])),
                "origin": self.vol_origin, "voxel": float(self.voxel_size)}

    def get_volume_all(self):
        """D2H copy of the three volumes, flat, z fastest (reference :1265-1277)."""
        self._wait_for_producer()
        n = self._n()
        self.tsdf_vol_cpu = self.tsdf_vol_gpu[:n].cpu().numpy()
        self.weight_vol_cpu = self.weight_vol_gpu[:n].cpu().numpy()
        self.color_vol_cpu = self.color_vol_gpu[:n].cpu().numpy()
        return self.tsdf_vol_cpu, self.weight_vol_cpu, self.color_vol_cpu

    # ------------------------------------------------------------------ bound logic (host, float64)
    def initialize_vol_bnd(self, cam_pose_iter, traj, version):
        """reference :910-925."""
        if version == "center":
            return self.center_volbnd(None, cam_pose_iter, traj)
        return self.more_volbnd(None, cam_pose_iter, traj)

    @staticmethod
    def _anchor(traj, pose) -> None:
        traj.kfx, traj.kfy, traj.kfz = pose[0, 3], pose[1, 3], pose[2, 3]

    def center_volbnd(self, vol_bnds, cam_pose_iter, tsdf_cam):
        """Box of half-lengths (x_len,y_len,z_len) around the camera rounded to whole metres
        (reference :1133-1149)."""
        self._anchor(tsdf_cam, cam_pose_iter)
        c = np.round(cam_pose_iter[:3, 3], 0)
        half = np.array([self.x_len, self.y_len, self.z_len], dtype=np.float64)
        out = np.zeros((3, 2))
        out[:, 0] = c - half
        out[:, 1] = c + half
        return out

    def _axis_order(self, cam_pose_iter, fixed):
        """sort world axes by angle to the camera forward direction (reference :1175-1194)."""
        fwd = np.matmul(cam_pose_iter[:3, :3], np.asarray([[0], [0], [1]], np.float32)).squeeze()
        units = np.eye(3, dtype=np.float32)
        if fixed is None:
            res = [self.require_angle_projection(fwd, units[a]) for a in range(3)]
        else:
            res = [self.require_angle_projection(fwd, units[a], fixed=fixed) for a in range(3)]
        angles = [r[0] for r in res]
        flags = [r[1] for r in res]
        s = sorted(angles)
        order = [angles.index(s[0]), angles.index(s[1]), angles.index(s[2])]
        return order, [flags[o] for o in order], s

    def more_volbnd(self, vol_bnds, cam_pose_iter, tsdf_cam):
        """reference :1151-1202."""
        out = np.zeros((3, 2))
        self._anchor(tsdf_cam, cam_pose_iter)
        center_cam = np.round(cam_pose_iter[:3, 3], 0)
        self.fixed_axis = None
        for name, fix, rng in (("x", self.fix_x, self.x_range), ("y", self.fix_y, self.y_range),
                               ("z", self.fix_z, self.z_range)):
            if fix:
                self.fixed_axis, self.fixed_range = name, rng
        order, oflags, _ = self._axis_order(cam_pose_iter, self.fixed_axis)
        tsdf_cam.first = order[0]
        out = self.more_calculations(out, order, oflags, center_cam)
        if self.fixed_axis is not None:
            out[_AXES[self.fixed_axis], :] = self.fixed_range[0], self.fixed_range[1]
        return out

    def more_calculations(self, vol_bnds, axis_priority, axis_flag, center_cam):
        """reference :1110-1131: long box ahead of the camera along the dominant axis."""
        first, second, third = axis_priority
        fl = self.first_len
        ahead = np.ceil(fl / 2) + fl
        behind = np.floor(fl / 2)
        f0 = axis_flag[0]
        vol_bnds[first, 0] = center_cam[first] - behind * f0 - ahead * (not f0)
        vol_bnds[first, 1] = center_cam[first] + ahead * f0 + behind * (not f0)
        vol_bnds[second, 0] = center_cam[second] - self.second_len
        vol_bnds[second, 1] = center_cam[second] + self.second_len
        vol_bnds[third, 0] = center_cam[third] - self.third_len
        vol_bnds[third, 1] = center_cam[third] + self.third_len
        return vol_bnds

    def require_angle(self, x, y, absolute=False):
        """angle (deg) between x and y, folded to <=90 with a sign flag (reference :1204-1232)."""
        cos_theta = x.dot(y) / (np.sqrt(x.dot(x)) * np.sqrt(y.dot(y)) + 1e-3)
        angle_value = np.arccos(cos_theta) * 180 / np.pi
        if absolute:
            return angle_value
        if angle_value > 90:
            return 180 - angle_value, -1
        return angle_value, 1

    def require_angle_projection(self, x, y, absolute=False, fixed="z"):
        """reference :1235-1251: drop the fixed axis, then require_angle."""
        keep = {"x": slice(1, None), "y": slice(0, None, 2), "z": slice(None, 2)}[fixed]
        return self.require_angle(x[keep], y[keep], absolute)

    def frameid_to_Vrange(self, value):
        """reference :1084-1105."""
        for (start, end), rng in self.frame_to_Vrange.items():
            if start <= value <= end:
                return rng
        return self.vol_bnds

    def check_move_volume_new(self, cur_id, cam_pose_iter, traj, version="center", larger_flag=False, get_pc=False,
                              gap=100) -> Tuple[bool, np.ndarray]:
        """Move the volume when the camera leaves the t_treshold box (and, for version 'more',
        when the dominant viewing axis changes).  reference :930-1082."""
        flag = False
        old_bnds = copy.deepcopy(self.vol_bnds)
        tmp = copy.deepcopy(self.vol_bnds)
        anchors = [traj.kfx, traj.kfy, traj.kfz]
        fixed = [self.fix_x, self.fix_y, self.fix_z]
        moved = False
        for a in range(3):
            delta = cam_pose_iter[a, 3] - anchors[a]
            if np.abs(delta) > self.t_treshold and not fixed[a]:
                tmp[a, :] += delta
                setattr(traj, ("kfx", "kfy", "kfz")[a], cam_pose_iter[a, 3])
                moved = True
        if moved:
            for a in range(3):
                tmp[a, 0] = round(tmp[a, 0], 0)
                tmp[a, 1] = round(tmp[a, 1], 0)
            if not (tmp == old_bnds).all():
                flag = True
                if self._can_shift_in_place_of_copy():
                    self.update_tsdf_swap_rot_trans(tmp, old_bnds, source="front")
                else:
                    self.copy_volume()
                    self.update_tsdf_swap_rot_trans(tmp, old_bnds)

        if version == "more":
            tmp = copy.deepcopy(self.vol_bnds)
            center_cam = np.round(cam_pose_iter[:3, 3], 0)
            order, oflags, sorted_angles = self._axis_order(cam_pose_iter, None)
            thr = self.more_angel_t * (2 if larger_flag else 1)
            if order[0] != traj.first and sorted_angles[0] < thr:
                self._anchor(traj, cam_pose_iter)
                vb = self.more_calculations(tmp, order, oflags, center_cam)
                if self.fixed_axis is not None:
                    vb[_AXES[self.fixed_axis], :] = self.fixed_range[0], self.fixed_range[1]
                if not (vb == old_bnds).all():
                    if get_pc and (cur_id - self.last_pcid) > gap:
                        self.last_pcid = cur_id
                        self.surface_pc = self.get_truncated_pc()
                    # NOTE: like the reference (:1078) no copy_volume() precedes this swap.
                    self.update_tsdf_swap_rot_trans(vb, old_bnds)
                    traj.first = order[0]
                    flag = True
        return flag, old_bnds
