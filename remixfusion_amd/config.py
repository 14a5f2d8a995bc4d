"""Config handling: the reference's YAML schema (config.py:4-51 there) plus built-in dicts for
the synthetic streams BASELINE.json names.

``load_config(path)`` reads a reference-format YAML with recursive ``inherit_from`` and a deep
merge, so existing RemixFusion config files keep working.  ``synthetic_config(name)`` returns
the same schema filled with the values of the corresponding reference base file (SURVEY.md
Appendix C) but sized for the synthetic 640x480 / 320x240 streams.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Optional

import yaml


def deep_update(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    """Recursive dict merge: values of ``src`` win; nested dicts are merged key by key."""
    for key, val in src.items():
        if isinstance(val, dict):
            node = dst.get(key)
            if not isinstance(node, dict):
                node = {}
                dst[key] = node
            deep_update(node, val)
        else:
            dst[key] = val
    return dst


def load_config(path: str, default_path: Optional[str] = None, cull_mesh: bool = False) -> Dict[str, Any]:
    with open(path, "r") as fh:
        special = yaml.full_load(fh) or {}
    parent = special.get("inherit_from")
    if parent is not None and cull_mesh:
        parent = os.path.join("../", parent)
    if parent is not None:
        cfg = load_config(parent, default_path)
    elif default_path is not None:
        with open(default_path, "r") as fh:
            cfg = yaml.full_load(fh) or {}
    else:
        cfg = {}
    return deep_update(cfg, special)


def _cam(H: int, W: int, near: float, far: float, depth_trunc: float) -> Dict[str, Any]:
    # fx = fy = 0.9 W, principal point at the image centre
    return {"H": H, "W": W, "fx": 0.9 * W, "fy": 0.9 * W, "cx": (W - 1) / 2.0, "cy": (H - 1) / 2.0,
            "png_depth_scale": 1000.0, "crop_edge": 0, "near": near, "far": far, "depth_trunc": depth_trunc}


def _axis(length: float) -> Dict[str, Any]:
    return {"fix": 0, "len": length, "range": [0, 1]}


_BASE: Dict[str, Any] = {   # values of configs/Replica/replica.yaml
    "dataset": "synthetic",
    "data": {"downsample": 1, "sc_factor": 1, "translation": 0, "num_workers": 0, "exp_name": "synthetic",
             "output": "output/synthetic", "datadir": "synthetic/room", "trainskip": 1},
    "globalV": {"use": 1, "base_resolution": 200, "n_levels": 1, "per_level_scale": 1, "n_features_per_level": 4},
    "mapping": {"sample": 2048, "first_mesh": False, "iters": 5, "BA_iters": 5, "lr_embed": 0.01,
                "lr_embed_res": 0.01, "lr_decoder": 0.01, "lr_rot": 0.0005, "lr_trans": 0.0005, "lr_pose": 0.0005,
                "keyframe_every": 5, "map_every": 5, "n_pixels": 0.05, "first_iters": 200, "optim_cur": False,
                "min_pixels_cur": 100, "map_accum_step": 1, "pose_accum_step": 1, "map_wait_step": 0,
                "filter_depth": False, "opt_pose": True, "clamp": 1.0, "pose_scale": 0.01, "save_ckpt": False,
                "device_sampling": True, "direct_iterations": True, "unused_gradients": False,
                "bound": [[-3, 3], [-4, 2.5], [-2, 2.5]],
                "marching_cubes_bound": [[-2.2, 2.6], [-3.4, 2.1], [-1.4, 2.0]]},
    "grid": {"enc": "HashGrid", "tcnn_encoding": True, "hash_size": 16, "voxel_color": 0.08, "voxel_sdf": 0.02},
    "pos": {"enc": "OneBlob", "n_bins": 16},
    "decoder": {"geo_feat_dim": 15, "hidden_dim": 32, "num_layers": 2, "num_layers_color": 2,
                "hidden_dim_color": 32, "tcnn_network": False},
    "cam": _cam(480, 640, 0.1, 5.0, 100.0),
    "training": {"rgb_weight": 5.0, "depth_weight": 0.1, "sdf_weight": 1000, "fs_weight": 10, "surface_weight": 0,
                 "eikonal_weight": 0, "smooth_weight": 0.000001, "smooth_pts": 32, "smooth_vox": 0.1,
                 "smooth_margin": 0.05, "n_samples_d": 11, "range_d": 0.15, "n_range_d": 48, "n_importance": 0,
                 "perturb": 1, "white_bkgd": False, "c_trunc": 0.1, "trunc": 0.05, "rot_rep": "axis_angle",
                 "rgb_missing": 0.05},
    "mesh": {"resolution": 512, "vis": 1000, "voxel_eval": 0.06, "voxel_final": 0.02, "visualisation": False,
             "mesh_bound_scale": 1.02, "only_final": 1, "render_img": 0},
    "volume": {"voxel_size": 0.01, "version": "center", "trunc": 0.05, "weight_threshold": 2.0, "weight_clamp": 1.0,
               "t_treshold": 1, "x_config": _axis(4), "y_config": _axis(4), "z_config": _axis(3),
               "first_len": 4, "second_len": 4, "third_len": 3, "more_angel_t": 20},
    "RO": {"init_size": 0.02, "scaling_coefficient": 0.09, "particle_iter_lens": 20, "PST_size": [10240, 3072, 1024],
           "PST_path": "PFO/fps_uniform_sphere",     # the reference's directory (relative to ITS checkout); RFX_PST_PATH overrides
           "PST_fallback": "generated",              # nothing found: seeded templates + a warning (the package ships no template data)
           "PST_seed": 20251205, "count_search": 200, "fix_level_index": 0, "filter_weight": 2, "rgb_rose": 0,
           "save_volume": 0, "save_freq": 1000, "cut": 0, "cut_dist": 8.0, "sample_range": 0.0, "iterative_scale": True},
    "tracking": {"ignore_edge_W": 20, "ignore_edge_H": 20, "const_speed": True},
    "video": {"save": False, "save_freq": 20},
    "synthetic": {"room": [[-2.8, 2.8], [-3.6, 2.1], [-1.5, 1.5]], "n_frames": 600, "seed": 20251205,
                  "depth_noise": 0.002, "dropout": 0.05, "pose_opt": True},
}

_OVERRIDES: Dict[str, Dict[str, Any]] = {
    # cfg 1: Replica room0, 320x240, TSDF-only, 4 cm MV so a CPU finishes
    "room0_tsdf": {"cam": _cam(240, 320, 0.1, 5.0, 100.0),
                   "mapping": {"bound": [[-1, 7], [-1.3, 3.7], [-1.7, 1.4]],
                               "marching_cubes_bound": [[-1, 7], [-1.3, 3.7], [-1.7, 1.4]]},
                   "volume": {"voxel_size": 0.04},
                   "synthetic": {"room": [[-0.8, 6.8], [-1.1, 3.5], [-1.5, 1.2]], "tsdf_only": True}},
    # cfg 2: Replica office0, 640x480, full mapping, GT poses
    "office0": {},
    # cfg 3: ScanNet scene0000 (configs/ScanNet/scannet.yaml + scene0000.yaml values)
    "scene0000": {"cam": _cam(460, 620, 0.0, 6.0, 5.0),
                  "mapping": {"bound": [[-0.2, 8.6], [-0.2, 8.9], [-0.2, 3.4]], "first_iters": 500, "min_pixels_cur": 20,
                              "filter_depth": True, "clamp": 1.5, "lr_pose": 0.002,
                              "marching_cubes_bound": [[-0.2, 8.6], [-0.2, 8.9], [-0.2, 3.4]]},
                  "grid": {"hash_size": 19},
                  "training": {"smooth_weight": 0.001, "smooth_pts": 64, "n_samples_d": 96, "range_d": 0.25,
                               "n_range_d": 21, "c_trunc": 0.25, "trunc": 0.06, "rgb_missing": 0.0},
                  "volume": {"voxel_size": 0.04, "trunc": 0.15, "x_config": _axis(5), "y_config": _axis(5),
                             "z_config": _axis(3)},
                  "synthetic": {"room": [[0.0, 8.4], [0.0, 8.7], [0.0, 3.2]]}},
    # cfg 4: BS3D cafeteria, 2 cm voxels, 1280x720
    "cafeteria": {"cam": _cam(720, 1280, 0.0, 8.0, 100.0),
                  "mapping": {"bound": [[-16, 15], [-20, 15], [-3, 7]], "first_iters": 500, "filter_depth": True,
                              "clamp": 2.0, "pose_scale": 1.0,
                              "marching_cubes_bound": [[-16, 15], [-20, 15], [-3, 7]]},
                  "grid": {"hash_size": 21},
                  "training": {"smooth_weight": 0.001, "smooth_pts": 64, "range_d": 0.5, "c_trunc": 0.25, "trunc": 0.06,
                               "rgb_missing": 0.0},
                  "volume": {"voxel_size": 0.02, "trunc": 0.06, "x_config": _axis(7), "y_config": _axis(7),
                             "z_config": _axis(3)},
                  "synthetic": {"room": [[-12, 12], [-14, 12], [-1.5, 4.5]]}},
    # cfg 5: uHumans2 apartment at 1 cm (BASELINE asks 1 cm), 720x480
    "apartment": {"cam": _cam(480, 720, 0.0, 20.0, 100.0),
                  "mapping": {"bound": [[-13, 13], [-7, 7], [-1, 7]], "iters": 10, "first_iters": 500,
                              "optim_cur": True, "filter_depth": True, "lr_pose": 0.00005,
                              "marching_cubes_bound": [[-13, 13], [-7, 7], [-1, 7]]},
                  "grid": {"hash_size": 21},
                  "training": {"smooth_weight": 0.001, "smooth_pts": 64, "n_samples_d": 96, "range_d": 0.5,
                               "n_range_d": 21, "c_trunc": 0.25, "trunc": 0.06, "rgb_missing": 0.0},
                  "volume": {"voxel_size": 0.01, "trunc": 0.06, "x_config": _axis(8), "y_config": _axis(8),
                             "z_config": _axis(3)},
                  # BASELINE config 5: "online marching-cubes mesh extract per keyframe" = the in-loop export of the reference
                  # (mp_slam/mapper.py:912-918) switched on at the keyframe rate
                  "mesh": {"vis": 5, "only_final": 0},
                  "synthetic": {"room": [[-10, 10], [-5, 5], [0, 3]]}},
    # north-star target workload (SURVEY 8d "stress"): a (10 m)^3 scene at 1 cm on one GPU -- 1000^3 voxels, 12 GB + back
    # buffers; field and schedule as cfg 2
    "stress10m": {"cam": _cam(480, 640, 0.1, 8.0, 100.0),
                  "mapping": {"bound": [[-5, 5], [-5, 5], [-5, 5]], "marching_cubes_bound": [[-5, 5], [-5, 5], [-5, 5]]},
                  "volume": {"x_config": _axis(5), "y_config": _axis(5), "z_config": _axis(5)},
                  "synthetic": {"room": [[-4.7, 4.7], [-4.6, 4.6], [-4.5, 4.5]]}},
}


def synthetic_config(name: str = "office0") -> Dict[str, Any]:
    if name not in _OVERRIDES:
        raise KeyError(f"unknown synthetic config {name!r}; have {sorted(_OVERRIDES)}")
    cfg = copy.deepcopy(_BASE)
    deep_update(cfg, copy.deepcopy(_OVERRIDES[name]))
    cfg["data"]["exp_name"] = name
    return cfg
