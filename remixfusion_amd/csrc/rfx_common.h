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

// lc[0..3] = losses (rgb, depth, sdf, fs), lc[4..7] = coef: d loss_i / d (its squared-error sum), from the per-block
// partial sums; for a 256-thread block, lc in LDS, valid for all threads on return.
__device__ __forceinline__ void loss_finalize(const double* __restrict__ partial, int n_partials, int64_t n_rays, int S, float* lc) {
    __shared__ double part[32][8], sums[8];
    {   // thread = (value v, slice q): partials q, q + 32, ... in order, then the 32 slices in order
        const int v = threadIdx.x & 7, q = threadIdx.x >> 3;
        double a = 0.0;
        for (int k = q; k < n_partials; k += 32) a += partial[k * 8 + v];
        part[q][v] = a;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double a = 0.0;
        for (int q = 0; q < 32; ++q) a += part[q][threadIdx.x];
        sums[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ns = (double)n_rays * (double)S;
        const double tot = sums[5] + sums[6];
        const float fs_w = (float)(1.0 - sums[5] / tot), sdf_w = (float)(1.0 - sums[6] / tot);
        const float c_rgb = (float)(1.0 / (3.0 * (double)n_rays)), c_dep = (float)(1.0 / sums[2]);
        const float c_sdf = (float)(1.0 / ns) * sdf_w, c_fs = (float)(1.0 / ns) * fs_w;
        lc[0] = (float)sums[0] * c_rgb; lc[1] = (float)sums[1] * c_dep;
        lc[2] = (float)sums[4] * c_sdf; lc[3] = (float)sums[3] * c_fs;
        lc[4] = c_rgb; lc[5] = c_dep; lc[6] = c_sdf; lc[7] = c_fs;
    }
    __syncthreads();
}


// fused forms of public entry points, used by rfx_ba_forward_backward (defined in rfx_render.hip)
int ba_prologue(const float* kf_rays, int64_t rays_per_kf, int64_t num_kf, const int64_t* kf_frame_ids, int keyframe_every,
                const float* cur_rays, int64_t cur_population, int64_t n_kf_samples, int64_t n_cur, uint64_t seed_kf,
                uint64_t seed_cur, const float* poses16, int K, const rfx_sampler_desc* sampler, const float* u01, uint64_t seed_u,
                const double bbox[6], int bbox_f64, float* rays_o, float* rays_d, float* target_rgb, float* target_d,
                float* d_cam, int32_t* pose_idx, float* z_vals, float* x01, const rfx_field_desc* field, const float* u6, int tv_P,
                float tv_voxel, float tv_margin, int tv_normalise, float* tv_pts, float* tv_feat, float* zero, int64_t zero_floats,
                float trunc_loss, float depth_trunc, double* count_partials, int* n_count_partials, rfx_stream stream,
                const rfx_grid_desc* tv_grid = nullptr,      // (tv_grid: the lattice's lookups use it instead of field->hash)
                int parts = 3);                              // 1: the ray batch (+ weight staging); 2: TV lattice + zero-fill; 3: both
int ba_count_partials(int64_t n_rays);                       // count_partials triples ba_prologue writes for a batch of n_rays
int composite_loss_grad(const float* raw4, const float* z_vals, const float* target_rgb, const float* target_d, int64_t n_rays, int S,
                        float trunc, float sc_factor, float trunc_loss, float depth_trunc, int rgb_missing_on, float* rgb_map,
                        float* depth_map, double* sums, int* n_partials, const double* count_partials, int n_count_partials,
                        const float* gout4, float* d_raw4, int32_t* ray_counts, const float* tv_feat, int tv_P, int tv_C, float tv_scale,
                        float* tv_dfeat, rfx_stream stream, int64_t n_rays_total = 0);   // (> n_rays: a share of a larger batch)
int loss_finalize_launch(const double* sums, int n_partials, int64_t n_rays, int S, float* lc8, rfx_stream stream);
int field_backward_weights_overwrite(int64_t n, const float* draw4, float* dw1, float* dw2, float* dw3, float* dw4,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream);   // rfx_field.hip
int pose_chain_backward(const float* dx01, const float* z_vals, const float* d_cam, const int32_t* pose_idx, int64_t n, int S,
                        const double bbox[6], int K, float* dposes16, const rfx_rba_params* prm, const float* acts, float scale,
                        const rfx_rba_grads* gr, float* workspace, rfx_stream stream, int* done);      // rfx_pose.hip
int field_backward_weights_scatter(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, float* dw1, float* dw2,
                                   float* dw3, float* dw4, const float* extra_x01, const float* extra_dfeat, int64_t extra_n,
                                   float* d_hash, void* workspace, size_t workspace_bytes, void* scatter_ws, size_t scatter_bytes,
                                   rfx_stream stream, int overwrite_from_level = RFX_MAX_LEVELS + 1,
                                   void* weights_done_event = nullptr);   // rfx_field.hip
// the scatter WRITES (not adds) the gradient of the levels from the returned one on: the caller skips their zero-fill (rfx_field.hip)
int scatter_overwrite_from_level(const rfx_grid_desc& g, int64_t n_all, bool have_scratch, const float* dtable);
int grid_encode_backward_merged_from(const rfx_grid_desc* g, const float* table, const float* x01_a, int64_t n_a,
                                     const float* dfeat_a, const float* x01_b, int64_t n_b, const float* dfeat_b, float* dtable,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream, int overwrite_from_level);
int field_backward_chain_stashed_counted(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, void* workspace,
                                         size_t workspace_bytes, int variant, const int* ray_counts, int S, const double* loss_partials,
                                         int n_loss_partials, float* lc8, int* finalized, rfx_stream stream);   // rfx_field.hip

}  // namespace rfx
