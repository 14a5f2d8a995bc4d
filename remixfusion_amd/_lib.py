"""ctypes binding of librfx.so (include/rfx.h).  There is no CPU fallback: if the library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RFX_LIB_PATH", os.path.join(_HERE, "librfx.so"))   # override: A/B builds of the library
RFX_MAX_LEVELS = 16


class RfxError(RuntimeError):
    pass


class GridDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_feat", C.c_int32),
                ("scale", C.c_float * RFX_MAX_LEVELS), ("res", C.c_uint32 * RFX_MAX_LEVELS),
                ("size", C.c_uint32 * RFX_MAX_LEVELS), ("offset", C.c_uint32 * RFX_MAX_LEVELS),
                ("hashed", C.c_uint32 * RFX_MAX_LEVELS)]


class FieldDesc(C.Structure):
    _fields_ = [("hash", GridDesc), ("hash_table", C.c_void_p), ("gbv", C.c_void_p), ("gbv_res", C.c_int32),
                ("w1", C.c_void_p), ("w2", C.c_void_p), ("w3", C.c_void_p), ("w4", C.c_void_p),
                ("tsdf_scale", C.c_float), ("c_trunc", C.c_float), ("trunc", C.c_float),
                ("clamp_hi", C.c_float), ("clamp_mode", C.c_int32), ("pos_fp16", C.c_int32), ("staged", C.c_void_p)]


class SamplerDesc(C.Structure):
    _fields_ = [("near", C.c_float), ("far", C.c_float), ("range_d", C.c_float),
                ("n_range_d", C.c_int32), ("n_samples_d", C.c_int32), ("perturb", C.c_float)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("beta1", C.c_float), ("beta2", C.c_float), ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float),
                ("eps", C.c_float), ("weight_decay", C.c_float), ("neg_step_size", C.c_float), ("bias_correction2_sqrt", C.c_float)]


ADAM_MAX_TENSORS = 16
LOSS_WS_DOUBLES = 2048      # RFX_LOSS_WS_DOUBLES
# rfx_ba_desc.stage_events (RFX_BA_EV_*): entry i is recorded behind the last launch of stage i
BA_STAGE_EVENTS = 10
BA_EV = {"start": 0, "prologue": 1, "forward": 2, "loss": 3, "chain": 4, "weights": 5, "scatter": 6, "dx_table": 7, "dx": 8, "pose": 9}


class BaDesc(C.Structure):
    _fields_ = [("field", FieldDesc), ("sampler", SamplerDesc), ("bbox", C.c_double * 6), ("bbox_f64", C.c_int32),
                ("sc_factor", C.c_float), ("depth_trunc", C.c_float), ("trunc", C.c_float), ("rgb_missing_on", C.c_int32),
                ("loss_w_dev", C.c_void_p), ("tv_P", C.c_int32), ("tv_voxel", C.c_float), ("tv_margin", C.c_float),
                ("tv_scale", C.c_float), ("tv_normalise", C.c_int32), ("kf_rays", C.c_void_p), ("rays_per_kf", C.c_int64),
                ("num_kf", C.c_int64), ("kf_frame_ids", C.c_void_p), ("keyframe_every", C.c_int32), ("cur_rays", C.c_void_p),
                ("cur_population", C.c_int64), ("n_kf_samples", C.c_int64), ("n_cur", C.c_int64), ("seed_kf", C.c_uint64),
                ("seed_cur", C.c_uint64), ("poses16", C.c_void_p), ("K", C.c_int32), ("u_z", C.c_void_p), ("u6", C.c_void_p),
                ("seed_u", C.c_uint64), ("hash_entries", C.c_int64), ("d_hash", C.c_void_p), ("d_w", C.c_void_p), ("d_poses16", C.c_void_p),
                ("losses8", C.c_void_p), ("tv_sum", C.c_void_p), ("rba", C.c_void_p), ("rba_acts", C.c_void_p),
                ("rba_scale", C.c_float), ("rba_grads", C.c_void_p), ("rba_ws", C.c_void_p),
                ("stage_events", C.c_void_p)]


class LevelRows(C.Structure):
    _fields_ = [("rows", C.c_void_p * RFX_MAX_LEVELS), ("ld", C.c_int32 * RFX_MAX_LEVELS), ("col", C.c_int32 * RFX_MAX_LEVELS)]


class BaShard(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("level_start", C.c_int32 * (RFX_MAX_LEVELS + 1)),
                ("ray_start", C.c_int64 * (RFX_MAX_LEVELS + 1)), ("feat_send", C.c_void_p), ("feat_recv", C.c_void_p),
                ("demb_send", C.c_void_p), ("demb_recv", C.c_void_p), ("loss_sums8", C.c_void_p), ("dx_send", C.c_void_p),
                ("dx_recv", C.c_void_p)]


RFX_TRACK_STEPS = 20
RFX_TRACK_STATE_WORDS = 64
RFX_TRACK_MAX_COUNT_SEARCH = 512


class TrackSearch(C.Structure):
    _fields_ = [("tsdf", C.c_void_p), ("dx", C.c_int32), ("dy", C.c_int32), ("dz", C.c_int32), ("x0", C.c_int32), ("x1", C.c_int32),
                ("origin", C.c_float * 3), ("voxel", C.c_float), ("vertex4", C.c_void_p), ("normal3", C.c_void_p),
                ("templates", C.c_void_p * RFX_TRACK_STEPS), ("template_rows", C.c_int32 * RFX_TRACK_STEPS),
                ("n_eval", C.c_int32 * RFX_TRACK_STEPS), ("level", C.c_int32 * RFX_TRACK_STEPS), ("K", C.c_float * 9),
                ("H", C.c_int32), ("W", C.c_int32), ("count_search", C.c_int32), ("fix_level_index", C.c_int32),
                ("iterative_scale", C.c_int32), ("reserved", C.c_int32), ("scaling_coefficient", C.c_double), ("beta", C.c_double),
                ("state", C.c_void_p), ("value_q30", C.c_void_p), ("count", C.c_void_p)]


class RbaParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w0", "b0", "w1", "b1", "w2", "b2", "w3", "b3")] + [("hidden", C.c_int32)]


class RbaGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w0", "b0", "w1", "b1", "w2", "b2", "w3", "b3")]


_P = C.c_void_p
_F3 = C.c_float * 3
_F6 = C.c_float * 6
_F9 = C.c_float * 9
_F16 = C.c_float * 16
_D6 = C.c_double * 6
_i, _f, _l, _sz = C.c_int, C.c_float, C.c_int64, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/rfx.h declares (tests check this).
PROTOTYPES = {
    "rfx_abi_version": (_i, []),
    "rfx_last_hip_error": (_i, []),
    "rfx_event_create": (_i, [C.POINTER(C.c_void_p)]),
    "rfx_event_destroy": (_i, [_P]),
    "rfx_event_elapsed_ms": (_i, [_P, _P, C.POINTER(C.c_float)]),
    "rfx_tsdf_integrate_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "rfx_tsdf_integrate": (_i, [_P, _P, _P, _i, _i, _i, _F3, _f, _F9, _F16, _P, _P, _i, _i, _f, _f, _i, _i, _F6,
                                _i, _P, _sz, _P]),
    "rfx_tsdf_integrate_slab": (_i, [_P, _P, _P, _i, _i, _i, _i, _i, _F3, _f, _F9, _F16, _P, _P, _i, _i, _f, _f, _i, _i, _F6,
                                     _i, _P, _sz, _P]),
    "rfx_pack_color": (_i, [_P, _P, _l, _P]),
    "rfx_tsdf_fill": (_i, [_P, _P, _P, _l, _P]),
    "rfx_tsdf_copy": (_i, [_P, _P, _P, _P, _P, _P, _l, _P]),
    "rfx_tsdf_shift": (_i, [_P, _P, _P, _i, _i, _i, _F3, _P, _P, _P, _i, _i, _i, _F3, _f, _i, _P]),
    "rfx_tsdf_shift_slab": (_i, [_P, _P, _P, _i, _i, _i, _i, _i, _F3, _P, _P, _P, _i, _i, _i, _i, _i, _F3, _f, _i, _P]),
    "rfx_tsdf_shift_source_planes": (_i, [_i, _i, _F3, _i, _F3, _f, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rfx_tsdf_trilerp": (_i, [_P, _P, _P, _i, _i, _i, _F3, _f, _P, _l, _P, _P]),
    "rfx_tsdf_integrate_rgb": (_i, [_P, _P, _P, _i, _i, _i, _i, _i, _F3, _f, _F9, _F16, _P, _P, _i, _i, _f, _f, _i, _i, _F6,
                                     _i, _P, _sz, _P]),
    "rfx_tsdf_trilerp_slab": (_i, [_P, _P, _i, _i, _i, _i, _i, _P, _P, _F3, _f, _P, _l, _P, _P, _P]),
    "rfx_tsdf_filter": (_i, [_P, _P, _P, _l, _f, _P]),
    "rfx_tsdf_truncated_pc": (_i, [_P, _P, _i, _i, _i, _F3, _f, _f, _i, _f, _P, _P, _i, _P]),
    "rfx_tsdf_truncated_pc_slab": (_i, [_P, _P, _i, _i, _i, _i, _i, _F3, _f, _f, _i, _f, _P, _P, _P, _i, _P]),
    "rfx_track_evaluate_slab": (_i, [_P, _i, _i, _i, _i, _i, _F3, _f, _P, _P, _F9, _F3, _P, _F6, _i, _F9, _i, _i, _i, _i, _P, _P, _P]),
    "rfx_gbv_integrate": (_i, [_P, _P, _i, _F6, _F9, _P, _P, _P, _i, _i, _f, _f, _P]),
    "rfx_gbv_clear": (_i, [_P, _l, _P]),
    "rfx_grid_encode_forward": (_i, [C.POINTER(GridDesc), _P, _P, _l, _P, _P]),
    "rfx_grid_encode_backward_workspace_bytes": (C.c_size_t, [_l, _i]),
    "rfx_grid_encode_backward": (_i, [C.POINTER(GridDesc), _P, _P, _l, _P, _P, _P, _P, C.c_size_t, _P]),
    "rfx_oneblob_forward": (_i, [_P, _l, _i, _i, _P, _P]),
    "rfx_field_forward": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P]),
    "rfx_field_backward_workspace_bytes": (_sz, [_l]),
    "rfx_field_backward": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "rfx_field_backward_chain": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_backward_chain_inputs": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_backward_chain_weights": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_forward_stash": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_backward_chain_stashed": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_backward_chain_inputs_stashed": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_backward_chain_weights_stashed": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_backward_weights": (_i, [_l, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "rfx_field_backward_scatter": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _P, _sz, _P]),
    "rfx_field_backward_dx": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _P, _sz, _P]),
    "rfx_field_query_sdf": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P]),
    "rfx_field_query_color": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P]),
    "rfx_sample_z": (_i, [C.POINTER(SamplerDesc), _P, _P, _l, _P, _P]),
    "rfx_ray_points": (_i, [_P, _P, _P, _l, _i, _D6, _i, _P, _P]),
    "rfx_composite_forward": (_i, [_P, _P, _l, _i, _f, _f, _P, _P, _P, _P]),
    "rfx_composite_backward": (_i, [_P, _P, _l, _i, _f, _f, _P, _P, _P, _P]),
    "rfx_mapping_loss_forward": (_i, [_P, _P, _P, _P, _P, _P, _l, _i, _f, _f, _i, _P, _P, _P, _P]),
    "rfx_mapping_loss_sums": (_i, [_P, _P, _P, _P, _P, _P, _l, _i, _f, _f, _i, _P, _P, _P]),
    "rfx_mapping_loss_finalize": (_i, [_P, _l, _i, _P, _P, _P]),
    "rfx_mapping_loss_backward": (_i, [_P, _P, _P, _P, _P, _P, _l, _i, _f, _f, _f, _f, _i, _P, _P, _P, _P, _P, _P]),
    "rfx_tv_forward": (_i, [_P, _i, _i, _P, _P]),
    "rfx_tv_backward": (_i, [_P, _i, _i, _f, _P, _P, _P]),
    "rfx_random_subset": (_i, [C.c_uint64, _l, _l, _P, _P]),
    "rfx_uniform_draws": (_i, [C.c_uint64, _i, _l, _P, _P]),
    "rfx_track_search_bytes": (_sz, []),
    "rfx_track_search_begin": (_i, [_P, _F9, _F3, _F6, _P]),
    "rfx_track_search_evaluate": (_i, [_P, _P]),
    "rfx_track_search_update": (_i, [_P, _i, _P]),
    "rfx_track_search_run": (_i, [_P, _F9, _F3, _F6, _i, _P]),
    "rfx_random_subset_dev": (_i, [C.c_uint64, _P, _l, _l, _P, _P, _P]),
    "rfx_grid_encode_backward_workspace_bytes_for": (C.c_size_t, [_P, _l]),
    "rfx_ba_workspace_bytes_for": (C.c_size_t, [_l, _i, _i, _P]),
    "rfx_track_vertex": (_i, [_P, _P, _F9, _i, _i, _f, _f, _f, C.c_uint32, _P, _P]),
    "rfx_track_normal": (_i, [_P, _P, _i, _i, _P]),
    "rfx_track_evaluate": (_i, [_P, _i, _i, _i, _F3, _f, _P, _P, _F9, _F3, _P, _F6, _i, _F9, _i, _i, _i, _i, _P, _P, _P]),
    "rfx_field_backward_scatter_merged": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _l, _P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "rfx_field_staged_floats": (C.c_size_t, []),
    "rfx_field_stage_weights": (_i, [C.POINTER(FieldDesc), _P, _P]),
    "rfx_tv_lattice": (_i, [_P, _i, _f, _f, _D6, _i, _i, _P, _P]),
    "rfx_gather_rays": (_i, [_P, _l, _l, _P, _i, _P, _l, _l, _l, C.c_uint64, C.c_uint64, _P, _i, _P, _P, _P, _P, _P, _P, _P]),
    "rfx_pose_grad": (_i, [_P, _P, _P, _P, _l, _i, _P, _P]),
    "rfx_ba_desc_bytes": (C.c_size_t, []),
    "rfx_adam_tensor_bytes": (C.c_size_t, []),
    "rfx_adam_step": (_i, [C.POINTER(AdamTensor), C.c_int, _P]),
    "rfx_ba_workspace_bytes": (C.c_size_t, [_l, _i, _i, _i, _i]),
    "rfx_ba_forward_backward": (_i, [C.POINTER(BaDesc), _P, C.c_size_t, _P]),
    "rfx_ba_workspace_layout": (_i, [_l, _i, _i, _i, _i, C.POINTER(C.c_size_t), _i]),
    "rfx_field_stash_put": (_i, [C.POINTER(LevelRows), _l, _P, _sz, _P]),
    "rfx_field_forward_stashed": (_i, [C.POINTER(FieldDesc), _P, _l, _P, _P, _sz, _P]),
    "rfx_field_backward_demb_rows": (_i, [_l, C.POINTER(LevelRows), _P, _i, _P, _P, _sz, _P]),
    "rfx_grid_encode_backward_merged": (_i, [C.POINTER(GridDesc), _P, _P, _l, _P, _P, _l, _P, _P, _P, _sz, _P]),
    "rfx_ba_shard_bytes": (C.c_size_t, []),
    "rfx_ba_shard_lookup": (_i, [C.POINTER(BaDesc), C.POINTER(BaShard), _P, _sz, _P]),
    "rfx_ba_shard_lookup_rays": (_i, [C.POINTER(BaDesc), C.POINTER(BaShard), _P, _sz, _P]),
    "rfx_ba_shard_lookup_tv": (_i, [C.POINTER(BaDesc), C.POINTER(BaShard), _P, _sz, _P]),
    "rfx_ba_shard_render": (_i, [C.POINTER(BaDesc), C.POINTER(BaShard), _P, _sz, _P]),
    "rfx_ba_shard_scatter": (_i, [C.POINTER(BaDesc), C.POINTER(BaShard), _P, _sz, _P]),
    "rfx_ba_shard_pose": (_i, [C.POINTER(BaDesc), C.POINTER(BaShard), _P, _sz, _P]),
    "rfx_rba_acts_floats": (C.c_size_t, [_l]),
    "rfx_rba_grads_floats": (C.c_size_t, [_l]),
    "rfx_frame_pose": (_i, [_P, _P, _P, _P, _P]),
    "rfx_rba_set_init_pose": (_i, [_P, _i, _i, _P, _P, _P, _P]),
    "rfx_rba_forward": (_i, [C.POINTER(RbaParams), _P, _P, _P, _l, _i, _f, _P, _P, _P]),
    "rfx_rba_backward": (_i, [C.POINTER(RbaParams), _P, _l, _P, _f, C.POINTER(RbaGrads), _P, _P]),
    "rfx_mc_count": (_i, [_P, _P, _i, _i, _i, _f, _P, _P, _P]),
    "rfx_mc_emit": (_i, [_P, _P, _i, _i, _i, _f, _P, _i, _P, _P, _P, _P, _P]),
    "rfx_render_rays": (_i, [C.POINTER(FieldDesc), C.POINTER(SamplerDesc), _P, _P, _P, _P, _l, _D6, _i, _f, _P, _P, _P]),
}

_ERR = {-1: "RFX_ERR_ARG", -2: "RFX_ERR_HIP", -3: "RFX_ERR_UNSUPPORTED", -4: "RFX_ERR_WORKSPACE"}
_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen librfx.so and bind every prototype.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RfxError(f"{LIB_PATH} not found: build it with `python -m remixfusion_amd.build` "
                       "(hipcc --offload-arch=gfx950). remixfusion_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.rfx_abi_version() != 10:
        raise RfxError("librfx.so ABI version mismatch")
    if lib.rfx_adam_tensor_bytes() != C.sizeof(AdamTensor):
        raise RfxError("rfx_adam_tensor layout mismatch between librfx.so and _lib.AdamTensor")
    if lib.rfx_ba_desc_bytes() != C.sizeof(BaDesc):
        raise RfxError("rfx_ba_desc layout mismatch between include/rfx.h and remixfusion_amd/_lib.py")
    if lib.rfx_ba_shard_bytes() != C.sizeof(BaShard):
        raise RfxError("rfx_ba_shard layout mismatch between include/rfx.h and remixfusion_amd/_lib.py")
    if lib.rfx_track_search_bytes() != C.sizeof(TrackSearch):
        raise RfxError("rfx_track_search layout mismatch between include/rfx.h and remixfusion_amd/_lib.py")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        extra = f" (hipError {load().rfx_last_hip_error()})" if status == -2 else ""
        raise RfxError(f"{what} failed: {_ERR.get(status, status)}{extra}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """device pointer of a contiguous fp32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RfxError("librfx kernels need device (cuda/HIP) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise RfxError("tensor must be contiguous")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None) -> int:
    """the HIP stream torch is currently issuing to on `device` (a torch.device, an index, or None = current device)"""
    if _raw_stream is not None:          # same value as current_stream().cuda_stream without building a Stream object
        if isinstance(device, torch.device):
            idx = device.index
        else:
            idx = device
        if idx is None:
            idx = torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def farr(ctype, values: Sequence[float]):
    return ctype(*[float(v) for v in values])


def random_subset(population: int, k: int, device) -> torch.Tensor:
    """k distinct indices of range(population) as a device int64 tensor; the key comes from python's
    ``random`` so that ``random.seed`` / SLAM.seed_everything make runs repeatable."""
    import random as _random
    out = torch.empty(int(k), dtype=torch.int64, device=device)
    check(load().rfx_random_subset(_random.getrandbits(64), int(population), int(k), out.data_ptr(), stream_ptr(device)),
          "rfx_random_subset")
    return out
