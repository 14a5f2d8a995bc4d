// rfx_common.h -- shared host/device helpers for librfx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rfx.h"

#define RFX_WAVE 64

namespace rfx {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) { g_last_hip_error = (int)e; return RFX_ERR_HIP; }

#define RFX_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return ::rfx::hip_fail(_e);   \
    } while (0)

#define RFX_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return ::rfx::hip_fail(_e);   \
    } while (0)

inline hipStream_t as_stream(rfx_stream s) { return reinterpret_cast<hipStream_t>(s); }

// The whole library is compiled with -ffp-contract=off: a*b+c fuses only where fmaf() is
// written.  madd() marks the places where the reference's nvcc (-fmad=true) contracts.
__device__ __forceinline__ float madd(float a, float b, float c) { return fmaf(a, b, c); }

// CUDA __float2int_rn: round-half-even.
__device__ __forceinline__ int f2i_rn(float v) { return (int)rintf(v); }

// fused forms of public entry points, used by rfx_ba_forward_backward (defined in rfx_render.hip)
int ba_prologue(const float* kf_rays, int64_t rays_per_kf, int64_t num_kf, const int64_t* kf_frame_ids, int keyframe_every,
                const float* cur_rays, int64_t cur_population, int64_t n_kf_samples, int64_t n_cur, uint64_t seed_kf,
                uint64_t seed_cur, const float* poses16, int K, const rfx_sampler_desc* sampler, const float* u01, uint64_t seed_u,
                const double bbox[6], int bbox_f64, float* rays_o, float* rays_d, float* target_rgb, float* target_d,
                float* d_cam, int32_t* pose_idx, float* z_vals, float* x01, const rfx_field_desc* field, const float* u6, int tv_P,
                float tv_voxel, float tv_margin, int tv_normalise, float* tv_pts, float* tv_feat, float* zero, int64_t zero_floats,
                rfx_stream stream);
int field_backward_weights_overwrite(int64_t n, const float* draw4, float* dw1, float* dw2, float* dw3, float* dw4,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream);   // rfx_field.hip
int pose_chain_backward(const float* dx01, const float* z_vals, const float* d_cam, const int32_t* pose_idx, int64_t n, int S,
                        const double bbox[6], int K, float* dposes16, const rfx_rba_params* prm, const float* acts, float scale,
                        const rfx_rba_grads* gr, float* workspace, rfx_stream stream, int* done);      // rfx_pose.hip
int field_backward_weights_scatter(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, float* dw1, float* dw2,
                                   float* dw3, float* dw4, const float* extra_x01, const float* extra_dfeat, int64_t extra_n,
                                   float* d_hash, void* workspace, size_t workspace_bytes, void* scatter_ws, size_t scatter_bytes,
                                   rfx_stream stream);                                              // rfx_field.hip
int field_backward_chain_stashed_counted(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, void* workspace,
                                         size_t workspace_bytes, int variant, const int* ray_counts, int S, rfx_stream stream);   // rfx_field.hip
int composite_loss_forward(const float* raw4, const float* z_vals, const float* target_rgb, const float* target_d, int64_t n_rays,
                           int S, float trunc, float sc_factor, float trunc_loss, float depth_trunc, int rgb_missing_on,
                           float* rgb_map, float* depth_map, double* sums, int* n_partials, const float* tv_feat, int tv_P, int tv_C,
                           float tv_scale, float* tv_dfeat, rfx_stream stream);     // + rfx_tv_backward beside it (tv_dfeat given)
int loss_backward_from_partials(const float* raw4, const float* z_vals, const float* rgb_map, const float* depth_map,
                                const float* target_rgb, const float* target_d, int64_t n_rays, int S, float trunc, float sc_factor,
                                float trunc_loss, float depth_trunc, int rgb_missing_on, const double* sums, int n_partials,
                                const float* gout4, float* lc8, float* d_raw4, int32_t* ray_counts, rfx_stream stream);
                                // ray_counts (optional): [n_rays] samples of each ray whose d_raw4 row is not all zero

}  // namespace rfx
