"""ROTracker: gradient-free pose tracking by randomised (particle-swarm) optimisation against the
moving TSDF volume.  Host-side mirror of the reference ``model/ROtracker.py:33-971`` with its three
PyCUDA kernels replaced by librfx (``rfx_track_vertex / _normal / _evaluate``).

Differences, on purpose:
  * the pre-sampled particle templates ("PST", 60 float32 TIFFs under PFO/fps_uniform_sphere in the reference,
    read there with cv2) are read from ``RO.PST_path`` by ``model/pst.py`` (own baseline-TIFF reader) into the
    same ``ALL_PST[class][index]`` container (or from an ``.npz`` archive of the same 60 arrays: the test suite's fixture);
    where neither ``RFX_PST_PATH`` nor ``RO.PST_path`` names templates: seeded templates of the same structure, with a
    warning (``RO.PST_fallback: "generated"``, the default -- the package ships no template data);
  * ``cal_transform``'s python loop over up to 10 240 candidates is vectorised with numpy (same
    selection: the first ``count_search`` candidates that beat candidate 0, same weights);
  * compute_vertex's cuRAND jitter is replaced by a counter-based hash (exactly zero anyway for
    RO.sample_range = 0, which every reference config uses);
  * mesh dumps (RO.save_volume) are CPU debug I/O and not built;
  * the 20 iterations of ``random_optimization`` run on the device by default (``RO.device_search``, librfx
    ``rfx_track_search_*``): the pose, the search box and the loop's flags stay in device memory and the host reads them
    once per frame instead of copying 2 x P sums back and selecting candidates in numpy 20 times.  The host loop below is the
    restatement of the reference's and stays (``RO.device_search: False``); tests compare the two.
"""
from __future__ import annotations

import math
import random
from typing import Dict, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import _F3, _F6, _F9, check, farr, ptr, stream_ptr
from .traj import Trajectory
from .Volume import moving_volume


from .pst import generated_pst, load_pst, make_pst, pst_slot  # noqa: E402,F401  (make_pst re-exported)


class ROTracker(object):
    def __init__(self, cfg, data_stream, device=None, volume_factory=None) -> None:
        self.cfg = cfg
        ro = cfg["RO"]
        self.data_stream = data_stream
        self.device = torch.device(device if device is not None else "cuda:0")
        self.estimate_pose = []
        self.larger_flag = False
        self.init_size = ro["init_size"]
        self.scaling_coefficient = ro["scaling_coefficient"]
        self.particle_iter_lens = ro["particle_iter_lens"]
        self.PST_size = ro["PST_size"]
        self.fix_level_index = ro["fix_level_index"]
        self.count_search = ro["count_search"]
        self.filter_weight = ro["filter_weight"]
        self.cut, self.cut_dist = ro["cut"], ro["cut_dist"]
        self.truncation = cfg["volume"]["trunc"]
        self.sample_range = ro["sample_range"]
        self.iterative_scale = ro["iterative_scale"]
        self.device_search = bool(ro.get("device_search", True))
        self.get_pc = cfg["training"]["surface_weight"] > 0
        self.traj = Trajectory("./results/")
        self.start_frame, self.end_frame = 0, len(self.data_stream)

        init_batch = self.data_stream[0]
        init_pose = init_batch["c2w"].squeeze().cpu().numpy()
        self.RO_pose = []
        # volume_factory(cfg, traj, pose0): a one-scene-on-N-GPUs run hands in its x-slab volume (dist.sharded_volume)
        self.MV = (volume_factory(cfg, self.traj, init_pose.astype(np.float64)) if volume_factory is not None else
                   moving_volume(cfg, self.traj, init_pose.astype(np.float64), start=0, device=self.device))
        self.im_h, self.im_w = self.data_stream.H, self.data_stream.W
        d = self.data_stream
        self.K = np.array([[d.fx, 0.0, d.cx], [0.0, d.fy, d.cy], [0.0, 0.0, 1.0]])
        n = self.im_h * self.im_w
        # reference :111-117: buffers start as ones (border normals are never written and stay non-zero)
        self.depth_vertex_gpu = torch.ones(n * 4, dtype=torch.float32, device=self.device)
        self.normal_vertex_gpu = torch.ones(n * 3, dtype=torch.float32, device=self.device)
        self.move_frameid = 0
        self.initialize_search_size = np.zeros((6))
        self.previous_frame_success = False
        self.tiff_index = [0, 1 + 20, 2 + 40, 3, 4 + 20, 5 + 40, 6 + 0, 7 + 20, 8 + 40, 9 + 0, 10 + 20, 11 + 40, 12 + 0,
                           13 + 20, 14 + 40, 15 + 0, 16 + 20, 17 + 40, 18 + 0, 19 + 20]
        self.depth_level = [32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16]
        self.PST_path = ro.get("PST_path", "PFO/fps_uniform_sphere")        # reference configs/*/*.yaml RO.PST_path
        self.readpst(self.PST_path, self.PST_size)
        self.current_global_R = np.zeros((3, 3), dtype=np.float32)
        self.current_global_T = np.zeros((3), dtype=np.float32)
        rgb = torch.floor(init_batch["rgb"].squeeze() * 255.0)
        self.MV.integrate(rgb, init_batch["depth"].squeeze(), self.K, init_pose, self.MV.vol_bnds, obs_weight=1.)

    # ------------------------------------------------------------------ particle templates
    def readpst(self, PST_path, PST_size):
        """reference :834-866: ``ALL_PST[class][index]`` <- ``PST_path/pst_{size}_{num}.tiff`` ([P,6] float32), plus
        device copies.  ``RFX_PST_PATH`` in the environment overrides the configured directory (and must exist).  Where neither
        names templates: generated ones, with a warning (``RO.PST_fallback: "generated"``, the default -- the package ships no
        template data, model/pst.py), or an error (any other value of ``RO.PST_fallback``)."""
        import warnings
        from .pst import resolve_pst_source
        ro = self.cfg["RO"]
        path = resolve_pst_source(PST_path)
        if path is not None:
            self.ALL_PST = load_pst(path, PST_size, self.tiff_index)
            self.PST_source = path
        elif ro.get("PST_fallback", "generated") == "generated":
            warnings.warn(f"ROTracker: no PST templates at {PST_path!r} (and no RFX_PST_PATH); searching with GENERATED "
                          "templates (RO.PST_fallback='generated'): poses will differ from the reference's", stacklevel=2)
            self.ALL_PST = generated_pst(ro.get("PST_seed", 20251205), PST_size, self.tiff_index)
            self.PST_source = "generated"
        else:
            self.ALL_PST = load_pst(PST_path or "", PST_size, self.tiff_index)      # raises with the explanation
        self.ALL_PST_dev = {c: torch.from_numpy(a).to(self.device) for c, a in self.ALL_PST.items()}

    def get_PST(self, tiff_index):
        cls, _, slot = pst_slot(tiff_index)
        return self.ALL_PST[cls][slot, ...]

    def _get_PST_dev(self, tiff_index):
        cls, _, slot = pst_slot(tiff_index)
        return self.ALL_PST_dev[cls][slot]

    # ------------------------------------------------------------------ kernels
    def init_searchsize(self):
        self.iter_trans_vector = np.zeros((6), dtype=np.float32)
        self.search_size = np.zeros((6), dtype=np.float32)
        self.previous_search_size = np.zeros((6), dtype=np.float32)
        self.search_size[...] = self.init_size

    def init_depth_vertex(self, depth_im, cam_intr):
        """reference :426-451."""
        depth = depth_im if isinstance(depth_im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(depth_im, np.float32))
        self.depth_map_gpu = depth.to(self.device, torch.float32).reshape(-1).contiguous()
        seed_num = random.randint(1, 1000000)
        check(_lib.load().rfx_track_vertex(ptr(self.depth_map_gpu), ptr(self.depth_vertex_gpu), farr(_F9, np.asarray(cam_intr).reshape(-1)),
                                           self.im_h, self.im_w, float(self.cut_dist), float(self.truncation),
                                           float(self.sample_range), seed_num, None, stream_ptr(self.device)), "rfx_track_vertex")

    def init_normal(self):
        """reference :453-468."""
        check(_lib.load().rfx_track_normal(ptr(self.depth_vertex_gpu), ptr(self.normal_vertex_gpu), self.im_h, self.im_w,
                                           stream_ptr(self.device)), "rfx_track_normal")

    def evaluate_tsdf(self, cur_id, level, node_size, cam_intr, level_index):
        """mean |tsdf - target| per candidate (reference :536-604).  Returns (mean, sum, count) as numpy."""
        P = int(node_size // 1024) * 1024            # grid = int(node_size/(32*32)) blocks of 1024 candidates
        sums = torch.empty((2, P), dtype=torch.int64, device=self.device)      # fixed-point sums (2^-30 units), hit counts
        val, cnt = sums[0], sums[1]
        self.MV.track_evaluate(self.depth_vertex_gpu, self.normal_vertex_gpu, self.current_global_R.reshape(-1), self.current_global_T,
                               self._cand_dev, self.search_size, P, np.asarray(cam_intr).reshape(-1), self.im_h, self.im_w, level,
                               level_index, val, cnt)
        n_all = self.transform_candidate.shape[0]
        sv = np.zeros(n_all, np.float32)
        sc = np.zeros(n_all, np.float32)
        both = sums.cpu().numpy()
        sv[:P] = (both[0].astype(np.float64) * 2.0 ** -30).astype(np.float32)      # rounded to float32 once (rfx_track_search_update: the same)
        sc[:P] = both[1].astype(np.float32)
        return sv / (sc + 1e-6), sv, sc

    # ------------------------------------------------------------------ the search on the device
    def _search_desc(self, cam_intr):
        """rfx_track_search for the volume as it is NOW (it moves and re-grids between frames) + the scratch it owns"""
        if getattr(self, "_search_state", None) is None:
            rows = max(int(a.shape[1]) for a in self.ALL_PST_dev.values())
            self._search_state = torch.zeros(_lib.RFX_TRACK_STATE_WORDS, dtype=torch.float32, device=self.device)
            self._search_sums = torch.zeros((2, rows), dtype=torch.int64, device=self.device)      # value_q30, count
        s = _lib.TrackSearch()
        vol = self.MV.track_search_volume()
        s.tsdf = ptr(vol["tsdf"])
        s.dx, s.dy, s.dz = (int(v) for v in vol["dim"])
        s.x0, s.x1 = (int(v) for v in vol["slab"])
        for i in range(3):
            s.origin[i] = float(np.float32(vol["origin"][i]))
        s.voxel = float(vol["voxel"])
        s.vertex4, s.normal3 = ptr(self.depth_vertex_gpu), ptr(self.normal_vertex_gpu)
        for k in range(_lib.RFX_TRACK_STEPS):
            t = self._get_PST_dev(self.tiff_index[k])
            s.templates[k] = t.data_ptr()
            s.template_rows[k] = int(t.shape[0])
            s.n_eval[k] = int(self.PST_size[k % 3] // 1024) * 1024
            s.level[k] = int(self.depth_level[k])
        for i, v in enumerate(np.asarray(cam_intr, np.float32).reshape(-1)):
            s.K[i] = float(v)
        s.H, s.W = int(self.im_h), int(self.im_w)
        s.count_search, s.fix_level_index = int(self.count_search), int(bool(self.fix_level_index))
        s.iterative_scale = int(bool(self.iterative_scale))
        s.scaling_coefficient = float(self.scaling_coefficient)
        s.state, s.value_q30, s.count = ptr(self._search_state), self._search_sums[0].data_ptr(), self._search_sums[1].data_ptr()
        return s

    def random_optimization_device(self, cur_id, cam_pose, rgb_im, depth_im, cam_intr, beta=0.9, inherit=False):
        """``random_optimization`` with the loop on the device: 2 launches per iteration, ONE device->host copy per frame
        (the 64-word state: pose, search box, flags).  Same arithmetic as the host loop below."""
        import ctypes as C
        lib = _lib.load()
        R0 = np.asarray(cam_pose[:3, :3], np.float32).copy()
        T0 = np.asarray(cam_pose[:3, 3], np.float32).copy()
        if inherit is True and self.previous_frame_success:
            self.search_size = self.initialize_search_size
        else:
            self.init_searchsize()
        self.init_depth_vertex(depth_im, cam_intr)
        self.init_normal()
        s = self._search_desc(cam_intr)
        s.beta = float(beta)
        st = stream_ptr(self.device)
        args = (farr(_F9, R0.reshape(-1)), farr(_F3, T0), farr(_F6, self.search_size))
        if self.MV.track_search_reduce is None:
            check(lib.rfx_track_search_run(C.byref(s), *args, int(self.particle_iter_lens), st), "rfx_track_search_run")
        else:                                        # a slab of a sharded volume: the sums are added over the ranks in between
            check(lib.rfx_track_search_begin(C.byref(s), *args, st), "rfx_track_search_begin")
            for i in range(self.particle_iter_lens):
                check(lib.rfx_track_search_evaluate(C.byref(s), st), "rfx_track_search_evaluate")
                self.MV.track_search_reduce(self._search_sums)
                check(lib.rfx_track_search_update(C.byref(s), i, st), "rfx_track_search_update")
        state = self._search_state.cpu().numpy()                 # the frame's one synchronisation
        flags = state.view(np.int32)
        if flags[37]:
            raise ValueError("invalid quaternion in the particle template (reference exits here, :662-669)")
        self.current_global_R = state[0:9].reshape(3, 3).copy()
        self.current_global_T = state[9:12].copy()
        self.search_size = state[12:18].copy()
        self.previous_search_size = state[18:24].copy()
        self.search_successes = int(flags[38])
        if flags[36]:                                            # iteration 0 succeeded (reference :816-821)
            self.initialize_search_size = self.search_size
            self.previous_frame_success = True
        else:
            self.previous_frame_success = False
        cam_pose_iter = np.eye(4, dtype=np.float32)
        cam_pose_iter[:3, :3] = self.current_global_R
        cam_pose_iter[:3, 3] = self.current_global_T
        return cam_pose_iter

    # ------------------------------------------------------------------ host logic
    def update_PST(self, tsdf, mean_transform, min_scale=1e-3, scale=0.09):
        """anisotropic search-size update (reference :493-534): float64 throughout (numpy 1.21 promotes the reference's
        float32-scalar (op) Python-float expressions to float64), stored as float32."""
        s = [abs(float(mean_transform[k])) + min_scale for k in (0, 1, 2, 4, 5, 6)]
        norm = math.sqrt(s[0] ** 2 + s[1] ** 2 + s[2] ** 2 + s[3] ** 2 + s[4] ** 2 + s[5] ** 2)
        t = float(tsdf)
        for k in range(6):
            self.search_size[k] = scale * t * (s[k] / norm) + min_scale

    def cal_transform(self, search_value):
        """fitness-weighted mean of the first ``count_search`` candidates that beat the null candidate
        (reference :606-709).  Returns (success, min_tsdf, [tx,ty,tz,qw,qx,qy,qz]).  The reference's scalar loop, vectorised:
        float32 weights and products, float64 running sums in candidate order, as the reference's numpy (1.21.6) computes them
        (oracle/tracker_host_oracle.py spells the types out; tests/test_oracle_tracker_host.py compares bit for bit)."""
        mean_transform = np.zeros((7), dtype=np.float32)
        sv = np.asarray(search_value, np.float32)
        origin_tsdf = sv[0]
        better = np.flatnonzero(sv[1:] < origin_tsdf) + 1
        if better.size == 0:
            return False, origin_tsdf, mean_transform
        sel = better[:self.count_search]
        cand = np.asarray(self.transform_candidate, np.float32)[sel]
        fit = sv[sel]
        w = origin_tsdf - fit                                            # float32
        ss = np.asarray(self.search_size, np.float32)
        q = cand[:, 3:6] * ss[3:6]                                       # float32
        q2 = (q * q).astype(np.float64)
        rad = ((1.0 - q2[:, 0]) - q2[:, 1]) - q2[:, 2]
        if (rad < 0).any():
            raise ValueError("invalid quaternion in the particle template (reference exits here, :662-669)")
        cols = np.empty((sel.size, 9), np.float64)
        cols[:, 0:6] = cand * w[:, None]                                 # float32 products
        cols[:, 6] = np.sqrt(rad) * w.astype(np.float64)
        cols[:, 7] = w
        cols[:, 8] = fit * w                                             # float32 product
        sums = np.cumsum(cols, axis=0)[-1]                               # sequential, like the loop's `+=`
        sw = float(sums[7])
        mean_tsdf = float(sums[8]) / sw
        for k in range(3):
            mean_transform[k] = (float(sums[k]) / sw) * float(ss[k])
        qww = float(sums[6]) / sw
        qxx, qyy, qzz = ((float(sums[3 + k]) / sw) * float(ss[3 + k]) for k in range(3))
        lens = 1 / math.sqrt(qww * qww + qxx * qxx + qyy * qyy + qzz * qzz)
        mean_transform[3:7] = (qww * lens, qxx * lens, qyy * lens, qzz * lens)
        return True, mean_tsdf, mean_transform

    def _search_step(self, i, st, search_value, beta):
        """what one iteration does with the candidates' fitness (reference :745-826): move to the weighted mean, pick the next
        template and pixel offset, rescale and smooth the search box.  ``st``: the loop's success / previous_success /
        count_particle / level_index.  (The device-side search does the same in rfx_track_search_update.)"""
        success, min_tsdf, mean_transform = self.cal_transform(search_value)
        st["success"] = success
        count_particle = st["count_particle"]
        qw, qx, qy, qz = mean_transform[3:7]
        if success:
            if count_particle < 19:
                count_particle += 1
            Rinc = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                             [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                             [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]], dtype=np.float32)
            self.current_global_T += mean_transform[:3]
            self.current_global_R = np.matmul(Rinc, self.current_global_R)
        st["count_particle"] = count_particle
        level_index = 1 if self.fix_level_index else st["level_index"] + 5
        st["level_index"] = level_index % (self.depth_level[count_particle])
        st["min_tsdf"] = min_tsdf
        self.update_PST(min_tsdf, mean_transform, scale=self.scaling_coefficient)
        if st["previous_success"] and success:
            for k in range(6):                                       # float64 (Python float * float32 scalar), stored as float32
                self.search_size[k] = beta * float(self.search_size[k]) + (1 - beta) * float(self.previous_search_size[k])
        elif success:
            if self.iterative_scale:
                st["previous_success"] = True
            self.previous_search_size[:] = self.search_size
        if not success:
            st["previous_success"] = False
        if i == 0:
            if success:
                self.initialize_search_size = self.search_size
                self.previous_frame_success = True
            else:
                self.previous_frame_success = False

    def random_optimization(self, cur_id, cam_pose, rgb_im, depth_im, cam_intr, beta=0.9, inherit=False):
        """20 iterations of: evaluate the particle set around the current pose, move to the fitness-weighted
        mean, rescale the search box (reference :713-831)."""
        self.current_global_R = np.asarray(cam_pose[:3, :3], np.float32).copy()
        self.current_global_T = np.asarray(cam_pose[:3, 3], np.float32).copy()
        if inherit is True and self.previous_frame_success:
            self.search_size = self.initialize_search_size
        else:
            self.init_searchsize()
        self.init_depth_vertex(depth_im, cam_intr)
        self.init_normal()
        st = {"previous_success": False, "success": False, "count_particle": 0, "level_index": 5}
        for i in range(self.particle_iter_lens):
            if not st["success"]:
                st["count_particle"] = 0
            count_particle = st["count_particle"]
            PST_class = count_particle % 3
            self.transform_candidate = self.get_PST(self.tiff_index[count_particle])
            self._cand_dev = self._get_PST_dev(self.tiff_index[count_particle])
            level = self.depth_level[count_particle]
            search_value, sv, sc = self.evaluate_tsdf(cur_id, level, self.PST_size[PST_class], cam_intr, st["level_index"])
            self._search_step(i, st, search_value, beta)
        cam_pose_iter = np.eye(4, dtype=np.float32)
        cam_pose_iter[:3, :3] = self.current_global_R
        cam_pose_iter[:3, 3] = self.current_global_T
        return cam_pose_iter

    def do_tracking(self, init_pose, decoder, batch, device):
        """reference :869-907.  Returns (pose [4,4] numpy, rgb 0..255, depth) -- images stay on the device."""
        if isinstance(init_pose, torch.Tensor):
            init_pose = init_pose.detach().cpu().numpy()
        depth = batch["depth"].squeeze()
        rgb = torch.floor(batch["rgb"].squeeze() * 255.0)
        self.gt_pose = batch["c2w"].squeeze().cpu().numpy()
        search = self.random_optimization_device if self.device_search else self.random_optimization
        cam_pose_iter = search(batch["frame_id"], init_pose, rgb, depth, self.K)
        return cam_pose_iter, rgb, depth

    def post_processing(self, cur_id, cam_pose_iter, rgb, depth, est_c2w_data):
        """follow the camera with the volume, then integrate the frame (reference :911-945)."""
        move_flag, old_volbnd = self.MV.check_move_volume_new(cur_id, np.asarray(cam_pose_iter, np.float64), self.traj,
                                                              version=self.MV.version, larger_flag=self.larger_flag,
                                                              get_pc=self.get_pc)
        if move_flag:
            start = 0 if self.MV.start_id == 0 else self.MV.start_id
            self.MV.start_id = cur_id
            self.MV.frame_to_Vrange[(start, cur_id - 1)] = old_volbnd
            self.larger_flag = False
            self.move_frameid = cur_id
        self.MV.integrate(rgb, depth, self.K, cam_pose_iter, old_volbnd, obs_weight=1.)

    def cal_ape_error(self, gt, our_t):
        gt = gt.cpu().numpy() if isinstance(gt, torch.Tensor) else np.asarray(gt)
        our_t = our_t.cpu().numpy() if isinstance(our_t, torch.Tensor) else np.asarray(our_t)
        return float(np.average(np.abs(gt[:3, 3] - our_t)))
