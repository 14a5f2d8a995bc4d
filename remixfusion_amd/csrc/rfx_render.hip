// rfx_render.hip -- ray sampling, point normalisation, SDF volume rendering (forward/backward) and the
// fused eval renderer for gfx950.
//
// Replaces the ATen op chains of JointEncoding.render_rays / sdf2weights / raw2outputs
// (model/scene_rep.py:107-127,156-179,407-456).  Layout: one wave per ray, lanes along the samples
// (S = n_range_d + n_samples_d = 59 or 117 in the reference configs; S <= 128 supported), so every
// [n_rays, S, .] access is a coalesced row and all per-ray reductions are wave reductions.
// The fused renderer never materialises points, encodings or raw values: rays in, rgb/depth out.
#include "rfx_field_mlp.h"
#include <algorithm>

namespace rfx {

constexpr int MAX_S = 128;

// torch.linspace(start, end, steps)[i] in fp32 (ATen: symmetric evaluation around the middle)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

struct SamplerK {
    float near, far, range_d, perturb;
    int n_range_d, n_samples_d;
};

// Sorted merge by rank: element A[i] (uniform near..far) lands at i + #{B < A[i]}, element B[k]
// (around the target depth) at k + #{A <= B[k]} -- the value sequence torch.sort would produce.
// zs: per-wave LDS scratch of MAX_S floats.  Lane j then owns samples j, j+64.
__device__ __forceinline__ void sample_ray(const SamplerK& s, float target_d, float* zs, int lane) {
    const int nA = s.n_samples_d, nB = s.n_range_d, S = nA + nB;
    const bool has_d = target_d > 0.0f;
    for (int j = lane; j < S; j += 64) {
        float v;
        int rank;
        if (j < nA) {
            v = linspace_at(s.near, s.far, nA, j);
            int c = 0;
            for (int k = 0; k < nB; ++k) {
                const float b = has_d ? linspace_at(-s.range_d, s.range_d, nB, k) + target_d
                                      : linspace_at(s.near, s.far, nB, k);
                c += (b < v) ? 1 : 0;
            }
            rank = j + c;
        } else {
            const int k = j - nA;
            v = has_d ? linspace_at(-s.range_d, s.range_d, nB, k) + target_d : linspace_at(s.near, s.far, nB, k);
            int c = 0;
            for (int i = 0; i < nA; ++i) c += (linspace_at(s.near, s.far, nA, i) <= v) ? 1 : 0;
            rank = k + c;
        }
        zs[rank] = v;
    }
}

// stratified jitter of scene_rep.py:437-441 for sample j given its sorted neighbours
__device__ __forceinline__ float jitter(const float* zs, int j, int S, float u) {
    const float z = zs[j];
    const float lower = j == 0 ? z : 0.5f * (z + zs[j - 1]);
    const float upper = j == S - 1 ? z : 0.5f * (zs[j + 1] + z);
    return lower + (upper - lower) * u;
}

__global__ __launch_bounds__(256) void sample_z_kernel(SamplerK s, const float* __restrict__ target_d,
                                                       const float* __restrict__ u01, int64_t n_rays,
                                                       float* __restrict__ z_vals) {
    __shared__ float zsh[4][MAX_S];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = s.n_range_d + s.n_samples_d;
    float* zs = zsh[wv];
    for (int64_t ray = (int64_t)blockIdx.x * 4 + wv; ray < n_rays; ray += (int64_t)gridDim.x * 4) {
        sample_ray(s, target_d[ray], zs, lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave are visible to it
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < S; j += 64) {
            float z = zs[j];
            if (s.perturb > 0.0f && u01) z = jitter(zs, j, S, u01[ray * S + j]);
            z_vals[ray * S + j] = z;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// x01 = ((o + d*z) - bb_min) / (bb_max - bb_min); the reference evaluates the normalisation in
// float64 when mapping.bound holds a non-integer (torch type promotion, scene_rep.py:388).
struct BoxK { double lo[3], hi[3], inv[3]; float lo32[3], ext32[3]; int f64; };      // lo32 / ext32: the fp32 form's operands (host)

// float64 form (a bound with a non-integer entry, scene_rep.py:388): the quotient (p - lo) / (hi - lo) is formed as a product
// with the extent's reciprocal (host, float64): one rounding of 1.1e-16 relative more before the result is rounded to fp32,
// which changes that fp32 number for about one value in 5e8 (by one ulp) -- and costs two float64 operations instead of the
// ~35 instructions of a float64 division, three times per sample.
__device__ __forceinline__ float normalise(const BoxK& b, int d, float p) {
    if (b.f64) return (float)(((double)p - b.lo[d]) * b.inv[d]);
    return (p - b.lo32[d]) / b.ext32[d];
}

__global__ __launch_bounds__(256) void ray_points_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                         const float* __restrict__ z, int64_t n_rays, int S, BoxK box,
                                                         float* __restrict__ x01) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * S) return;
    const int64_t r = i / S;
    const float zz = z[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) x01[i * 3 + k] = normalise(box, k, o[r * 3 + k] + d[r * 3 + k] * zz);
}

// ------------------------------------------------------------------ R1: wave-level compositing
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// per-lane state for up to NS = 2 samples (j = lane, lane+64)
struct RayW {
    float w[2];       // normalised weights
    float wsum;       // sum of masked un-normalised weights + 1e-8
};

// s[c], z[c]: sdf / depth of this lane's samples (c = 0,1); valid[c] = sample exists.
__device__ __forceinline__ RayW ray_weights(const float s[2], const float z[2], const bool valid[2], int S, int lane,
                                            float trunc, float sc_factor) {
    // first j with s[j]*s[j+1] < 0  (argmax of the 0/1 mask; 0 when there is none)
    int first = S;   // sentinel
    const float s64 = __shfl(s[1], 0);                    // sample 64 lives in lane 0, chunk 1
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int j = lane + 64 * c;
        const float up = __shfl_down(s[c], 1);            // s[j+1] for lanes < 63
        const float nxt = lane < 63 ? up : (c == 0 ? s64 : 0.0f);
        if (j + 1 < S && s[c] * nxt < 0.0f) first = min(first, j);
    }
    first = wave_min(first);
    if (first == S) first = 0;
    const float z_first = first < 64 ? __shfl(z[0], first) : __shfl(z[1], first - 64);
    const float lim = z_first + sc_factor * trunc;
    RayW r;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float w = 0.f;
        if (valid[c]) {
            w = sigmoidf(s[c] / trunc) * sigmoidf(-s[c] / trunc);
            w = (z[c] < lim) ? w : 0.0f;
        }
        r.w[c] = w;
        sum += w;
    }
    r.wsum = wave_sum(sum) + 1e-8f;
#pragma unroll
    for (int c = 0; c < 2; ++c) r.w[c] = r.w[c] / r.wsum;
    return r;
}

__global__ __launch_bounds__(256) void composite_forward_kernel(const float4* __restrict__ raw, const float* __restrict__ zv,
                                                                int64_t n_rays, int S, float trunc, float sc,
                                                                float* __restrict__ rgb, float* __restrict__ depth,
                                                                float* __restrict__ weights) {
    const int lane = threadIdx.x & 63;
    for (int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += (int64_t)gridDim.x * 4) {
        float s[2], z[2];
        float4 rv[2];
        bool valid[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            valid[c] = j < S;
            rv[c] = valid[c] ? raw[ray * S + j] : make_float4(0.f, 0.f, 0.f, 0.f);
            z[c] = valid[c] ? zv[ray * S + j] : 0.f;
            s[c] = rv[c].w;
        }
        const RayW rw = ray_weights(s, z, valid, S, lane, trunc, sc);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            a0 += rw.w[c] * rv[c].x; a1 += rw.w[c] * rv[c].y; a2 += rw.w[c] * rv[c].z; ad += rw.w[c] * z[c];
            if (weights && valid[c]) weights[ray * S + lane + 64 * c] = rw.w[c];
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); ad = wave_sum(ad);
        if (lane == 0) { rgb[ray * 3] = a0; rgb[ray * 3 + 1] = a1; rgb[ray * 3 + 2] = a2; depth[ray] = ad; }
    }
}

__global__ __launch_bounds__(256) void composite_backward_kernel(const float4* __restrict__ raw, const float* __restrict__ zv,
                                                                 int64_t n_rays, int S, float trunc, float sc,
                                                                 const float* __restrict__ d_rgb,
                                                                 const float* __restrict__ d_depth,
                                                                 float4* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    for (int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += (int64_t)gridDim.x * 4) {
        float s[2], z[2];
        float4 rv[2];
        bool valid[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            valid[c] = j < S;
            rv[c] = valid[c] ? raw[ray * S + j] : make_float4(0.f, 0.f, 0.f, 0.f);
            z[c] = valid[c] ? zv[ray * S + j] : 0.f;
            s[c] = rv[c].w;
        }
        const RayW rw = ray_weights(s, z, valid, S, lane, trunc, sc);
        const float g0 = d_rgb[ray * 3], g1 = d_rgb[ray * 3 + 1], g2 = d_rgb[ray * 3 + 2], gd = d_depth[ray];
        // g_j = dL/d wn_j ; dL/dw_j = (g_j - sum_k wn_k g_k) / W
        float gj[2], dot = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            gj[c] = g0 * rv[c].x + g1 * rv[c].y + g2 * rv[c].z + gd * z[c];
            dot += rw.w[c] * gj[c];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!valid[c]) continue;
            float ds = 0.f;
            if (rw.w[c] != 0.0f) {    // mask passed (w is exactly 0 where masked)
                const float a = s[c] / trunc;
                const float sp = sigmoidf(a), sm = sigmoidf(-a);
                ds = (gj[c] - dot) / rw.wsum * (sp * sm * (sm - sp) / trunc);
            }
            d_raw[ray * S + lane + 64 * c] = make_float4(rw.w[c] * g0, rw.w[c] * g1, rw.w[c] * g2, ds);
        }
    }
}

// ------------------------------------------------------------------ fused eval renderer
// one wave per ray; lanes = samples; S1 + points + Q1 (encode + MFMA MLP) + R1, all in registers.
// NCHUNK = ceil(S / 64) lane passes per ray (1 for the 59-sample configurations, 2 for the 117-sample ones): with one pass
// nothing is carried between passes and the kernel fits its 256 registers without scratch (the generic two-pass form wrote
// 366 MiB of spills per 640x480 frame).
template <bool POS16, int NCHUNK>
__global__ __launch_bounds__(256, RENDER_WAVES) void render_rays_kernel(FieldK f, SamplerK s, BoxK box,
                                                                    const float* __restrict__ rays_o,
                                                                    const float* __restrict__ rays_d,
                                                                    const float* __restrict__ target_d,
                                                                    const float* __restrict__ u01, int64_t n_rays,
                                                                    float sc, float* __restrict__ rgb,
                                                                    float* __restrict__ depth) {
    __shared__ __attribute__((aligned(16))) float wl[FWD_SLOTS * 64];
    __shared__ float zsh[4][MAX_S];
    stage_weights(f, wl, FWD_SLOTS);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = s.n_range_d + s.n_samples_d;
    float* zs = zsh[wv];
    // XCD-contiguous ray order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one; speed only,
    // never correctness) and each XCD has its own L2.  Neighbouring rays of an image gather the same cells of the 128 MB
    // global volume and of the fine hash levels: with rays dealt out block by block every XCD's L2 fetched every line
    // (1.8 GB of fabric reads per 640x480 frame against a 128 MB table).  Here XCD slot x = b % 8 walks its own contiguous
    // eighth of the ray list, so a line is fetched by one L2 and re-used by the rays next to it and the rows below.
#ifndef RENDER_XCD_ORDER
#define RENDER_XCD_ORDER 1
#endif
    const bool by_xcd = RENDER_XCD_ORDER && gridDim.x >= 64;          // small ray lists: plain order
    const int64_t n_slots = by_xcd ? 8 : 1;
    const int64_t slot = by_xcd ? (blockIdx.x & 7) : 0, blk = by_xcd ? (blockIdx.x >> 3) : blockIdx.x;
    const int64_t blocks_per_slot = by_xcd ? ((int64_t)gridDim.x + 7 - slot) / 8 : gridDim.x;
    const int64_t r0 = n_rays * slot / n_slots, r1 = n_rays * (slot + 1) / n_slots;
    for (int64_t ray = r0 + blk * 4 + wv; ray < r1; ray += blocks_per_slot * 4) {
        sample_ray(s, target_d[ray], zs, lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
        const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
        float sdf[2] = {0.f, 0.f}, z[2] = {0.f, 0.f};
        float4 rv[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        bool valid[2] = {false, false};
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            const int j = lane + 64 * c;
            valid[c] = j < S;
            if (64 * c >= S) continue;           // wave-uniform: second chunk only for S > 64
            const int jj = valid[c] ? j : S - 1;
            float zz = zs[jj];
            if (s.perturb > 0.0f && u01) zz = jitter(zs, jj, S, u01[ray * S + jj]);
            z[c] = zz;
            float x[3];
            x[0] = normalise(box, 0, ox + dx * zz);
            x[1] = normalise(box, 1, oy + dy * zz);
            x[2] = normalise(box, 2, oz + dz * zz);
            Enc e;
            encode_point(f, x, e);
            Mlp m;
            mlp_forward_123<0, false, POS16>(f, x, wl, lane, e, m);
            float raw[4];
            mlp_forward_4(wl, lane, e, m, raw);
            rv[c] = make_float4(raw[0], raw[1], raw[2], raw[3]);
            sdf[c] = raw[3];
            __builtin_amdgcn_sched_barrier(0);       // S > 64: do not interleave the two chunks (register pressure)
        }
        __builtin_amdgcn_wave_barrier();
        const RayW rw = ray_weights(sdf, z, valid, S, lane, f.trunc, sc);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ad = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            a0 += rw.w[c] * rv[c].x; a1 += rw.w[c] * rv[c].y; a2 += rw.w[c] * rv[c].z; ad += rw.w[c] * z[c];
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); ad = wave_sum(ad);
        if (lane == 0) { rgb[ray * 3] = a0; rgb[ray * 3 + 1] = a1; rgb[ray * 3 + 2] = a2; depth[ray] = ad; }
    }
}

// ------------------------------------------------------------------ L1: fused mapping losses
// sums (double, caller zero-fills): 0 rgb sq-err, 1 depth sq-err (valid rays), 2 #valid rays,
// 3 free-space sq-err, 4 sdf sq-err, 5 #front samples, 6 #sdf samples (both counted before the
// valid-depth mask, like get_masks; model/utils.py:170-198, scene_rep.py:493-517).
struct LossK {
    float trunc_loss;     // training.trunc * data.sc_factor
    float depth_trunc;    // cam.depth_trunc
    int   rgb_missing_on; // training.rgb_missing > 0 (bool cast in the reference, see scene_rep.py:495-498)
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum `v` over the block's waves (fixed order); result valid in every thread.  red: one double of LDS per wave.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += red[w];
    __syncthreads();
    return r;
}

constexpr int LOSS_BLOCKS = 256, LOSS_THREADS = 256;      // 256 x 8 partial sums = RFX_LOSS_WS_DOUBLES; a ray costs a
                                                          // wave two dependent memory latencies, so many short waves
static_assert(LOSS_BLOCKS * 8 == RFX_LOSS_WS_DOUBLES, "loss workspace");

// per-block partial sums, plain stores: no atomics (one address sustains ~90 of them per us), no zero-fill before,
// and the finalize kernel adds the partials in a fixed order
__global__ __launch_bounds__(LOSS_THREADS) void mapping_loss_forward_kernel(LossK L, const float4* __restrict__ raw,
                                                                   const float* __restrict__ zv, const float* __restrict__ rgb_map,
                                                                   const float* __restrict__ depth_map,
                                                                   const float* __restrict__ tgt_rgb, const float* __restrict__ tgt_d,
                                                                   int64_t n_rays, int S, double* __restrict__ sums) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    double a_rgb = 0, a_dep = 0, a_val = 0, a_fs = 0, a_sdf = 0, a_nfs = 0, a_nsdf = 0;
    for (int64_t ray = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); ray < n_rays; ray += (int64_t)gridDim.x * wpb) {
        const float d = tgt_d[ray];
        const bool valid = (d > 0.0f) && (d < L.depth_trunc);
        if (lane == 0) {
            const float rw = (valid || L.rgb_missing_on) ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e = rgb_map[ray * 3 + c] * rw - tgt_rgb[ray * 3 + c] * rw;
                a_rgb += (double)(e * e);
            }
            if (valid) { const float e = depth_map[ray] - d; a_dep += (double)(e * e); a_val += 1.0; }
        }
        for (int j = lane; j < S; j += 64) {
            const float z = zv[ray * S + j], s = raw[ray * S + j].w;
            const float front = (z < (d - L.trunc_loss)) ? 1.0f : 0.0f;
            const float back = (z > (d + L.trunc_loss)) ? 1.0f : 0.0f;
            const float sm = (1.0f - front) * (1.0f - back) * (d > 0.0f ? 1.0f : 0.0f);
            a_nfs += front; a_nsdf += sm;
            if (valid) {
                const float ef = s * front - front;
                const float es = (z + s * L.trunc_loss) * sm - d * sm;
                a_fs += (double)(ef * ef); a_sdf += (double)(es * es);
            }
        }
    }
    __shared__ double red[LOSS_THREADS / 64];
    a_rgb = block_sum_d(a_rgb, red); a_dep = block_sum_d(a_dep, red); a_val = block_sum_d(a_val, red);
    a_fs = block_sum_d(a_fs, red); a_sdf = block_sum_d(a_sdf, red); a_nfs = block_sum_d(a_nfs, red);
    a_nsdf = block_sum_d(a_nsdf, red);
    if (threadIdx.x == 0) {
        double* o = sums + (size_t)blockIdx.x * 8;
        o[0] = a_rgb; o[1] = a_dep; o[2] = a_val; o[3] = a_fs; o[4] = a_sdf; o[5] = a_nfs; o[6] = a_nsdf; o[7] = 0.0;
    }
}

// (loss_finalize: rfx_common.h)
__global__ __launch_bounds__(256) void mapping_loss_finalize_kernel(const double* __restrict__ partial, int n_partials, int64_t n_rays,
                                                                    int S, float* __restrict__ losses, float* __restrict__ coef) {
    __shared__ float lc[8];
    loss_finalize(partial, n_partials, n_rays, S, lc);
    if (threadIdx.x < 4) { losses[threadIdx.x] = lc[threadIdx.x]; coef[threadIdx.x] = lc[4 + threadIdx.x]; }
}

// d_raw4 = d(sum_i gout_i * loss_i)/d raw  (+ optional external grads on the maps), compositing included.
// PARTIALS: the coefficients are not read from `coef` but derived (by every block, with the finalize kernel's own
// arithmetic) from the forward's per-block partial sums, and block 0 writes losses + coefficients to lc_out[8]: the forward
// then needs no finalize launch of its own.
template <bool PARTIALS>
__global__ __launch_bounds__(256) void mapping_loss_backward_kernel(LossK L, const float4* __restrict__ raw,
                                                                    const float* __restrict__ zv, const float* __restrict__ rgb_map,
                                                                    const float* __restrict__ depth_map,
                                                                    const float* __restrict__ tgt_rgb, const float* __restrict__ tgt_d,
                                                                    int64_t n_rays, int S, float trunc, float sc,
                                                                    const float* __restrict__ coef, const float* __restrict__ gout,
                                                                    const float* __restrict__ g_rgb_map,
                                                                    const float* __restrict__ g_depth_map, float4* __restrict__ d_raw,
                                                                    const double* __restrict__ partial, int n_partials,
                                                                    float* __restrict__ lc_out, int* __restrict__ ray_counts) {
    const int lane = threadIdx.x & 63;
    __shared__ float lc[8];
    if (PARTIALS) {
        loss_finalize(partial, n_partials, n_rays, S, lc);
        if (blockIdx.x == 0 && threadIdx.x < 8) lc_out[threadIdx.x] = lc[threadIdx.x];
        coef = lc + 4;
    }
    const float k_rgb = gout[0] * coef[0], k_dep = gout[1] * coef[1], k_sdf = gout[2] * coef[2], k_fs = gout[3] * coef[3];
    for (int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += (int64_t)gridDim.x * 4) {
        float s[2], z[2];
        float4 rv[2];
        bool ok[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            ok[c] = j < S;
            rv[c] = ok[c] ? raw[ray * S + j] : make_float4(0.f, 0.f, 0.f, 0.f);
            z[c] = ok[c] ? zv[ray * S + j] : 0.f;
            s[c] = rv[c].w;
        }
        const RayW rw = ray_weights(s, z, ok, S, lane, trunc, sc);
        const float d = tgt_d[ray];
        const bool valid = (d > 0.0f) && (d < L.depth_trunc);
        const float w2 = (valid || L.rgb_missing_on) ? 1.0f : 0.0f;
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g[c] = k_rgb * 2.0f * w2 * (rgb_map[ray * 3 + c] * w2 - tgt_rgb[ray * 3 + c] * w2);
            if (g_rgb_map) g[c] += g_rgb_map[ray * 3 + c];
        }
        float gd = valid ? k_dep * 2.0f * (depth_map[ray] - d) : 0.0f;
        if (g_depth_map) gd += g_depth_map[ray];
        float gj[2], dot = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            gj[c] = g[0] * rv[c].x + g[1] * rv[c].y + g[2] * rv[c].z + gd * z[c];
            dot += rw.w[c] * gj[c];
        }
        dot = wave_sum(dot);
        int nz = 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!ok[c]) continue;
            float ds = 0.f;
            if (rw.w[c] != 0.0f) {
                const float a = s[c] / trunc;
                const float sp = sigmoidf(a), sm = sigmoidf(-a);
                ds = (gj[c] - dot) / rw.wsum * (sp * sm * (sm - sp) / trunc);
            }
            if (valid) {
                const float front = (z[c] < (d - L.trunc_loss)) ? 1.0f : 0.0f;
                const float back = (z[c] > (d + L.trunc_loss)) ? 1.0f : 0.0f;
                const float smk = (1.0f - front) * (1.0f - back) * (d > 0.0f ? 1.0f : 0.0f);
                ds += k_fs * 2.0f * (s[c] * front - front) * front;
                ds += k_sdf * 2.0f * ((z[c] + s[c] * L.trunc_loss) * smk - d * smk) * (L.trunc_loss * smk);
            }
            const float4 o = make_float4(rw.w[c] * g[0], rw.w[c] * g[1], rw.w[c] * g[2], ds);
            d_raw[ray * S + lane + 64 * c] = o;
            nz += (o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) ? 1 : 0;
        }
        if (ray_counts) {                      // rows of this ray with a gradient (the selection of the field backward counts them)
            nz = (int)wave_sum((float)nz);
            if (lane == 0) ray_counts[ray] = nz;
        }
    }
}

// ------------------------------------------------------------------ TV1: total variation of lattice features
// feat [P,P,P,C]; sum over the three axes of squared forward differences (mp_slam/slam.py:211-215).
__global__ __launch_bounds__(256) void tv_forward_kernel(const float* __restrict__ feat, int P, int Cn, double* __restrict__ sum) {
    const int64_t total = (int64_t)P * P * P * Cn;
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = i / Cn;
        const int z = (int)(cell % P), y = (int)((cell / P) % P), x = (int)(cell / ((int64_t)P * P));
        const float v = feat[i];
        if (x + 1 < P) { const float e = feat[i + (int64_t)P * P * Cn] - v; acc += (double)(e * e); }
        if (y + 1 < P) { const float e = feat[i + (int64_t)P * Cn] - v; acc += (double)(e * e); }
        if (z + 1 < P) { const float e = feat[i + Cn] - v; acc += (double)(e * e); }
    }
    __shared__ double red[4];
    acc = block_sum_d(acc, red);
    if (threadIdx.x == 0) atomicAdd(sum, acc);
}

// dfeat = (*gscale) * scale * d(sum)/d feat
__device__ __forceinline__ void tv_backward_body(const float* __restrict__ feat, int P, int Cn, float scale,
                                                 const float* __restrict__ gscale, float* __restrict__ dfeat, int block, int n_blocks) {
    const int64_t total = (int64_t)P * P * P * Cn;
    const float k = 2.0f * scale * (gscale ? gscale[0] : 1.0f);
    for (int64_t i = (int64_t)block * blockDim.x + threadIdx.x; i < total; i += (int64_t)n_blocks * blockDim.x) {
        const int64_t cell = i / Cn;
        const int z = (int)(cell % P), y = (int)((cell / P) % P), x = (int)(cell / ((int64_t)P * P));
        const float v = feat[i];
        const int64_t sx = (int64_t)P * P * Cn, sy = (int64_t)P * Cn, sz = Cn;
        float g = 0.f;
        if (x + 1 < P) g -= feat[i + sx] - v;
        if (x > 0) g += v - feat[i - sx];
        if (y + 1 < P) g -= feat[i + sy] - v;
        if (y > 0) g += v - feat[i - sy];
        if (z + 1 < P) g -= feat[i + sz] - v;
        if (z > 0) g += v - feat[i - sz];
        dfeat[i] = k * g;
    }
}

__global__ __launch_bounds__(256) void tv_backward_kernel(const float* __restrict__ feat, int P, int Cn, float scale,
                                                          const float* __restrict__ gscale, float* __restrict__ dfeat) {
    tv_backward_body(feat, P, Cn, scale, gscale, dfeat, blockIdx.x, gridDim.x);
}

// the TV backward as a block role of another launch (composite_loss_grad_tv_kernel): the gradient of the TV term w.r.t. the
// lattice features the prologue looked up only feeds the backward, like the loss gradient made beside it
struct TvBackK { const float* feat; int P, Cn; float scale; float* dfeat; };

// R1 + L1 forward AND L1 backward of a BA iteration's rays in one launch.  The backward needs the whole batch's loss
// coefficients -- 1 / #valid rays, and the free-space / sdf weights from the two sample counts (model/utils.py:170-198) -- which
// is why forward and backward are two kernels everywhere else.  But those three counts depend on the target depths and the
// sample depths only, not on the field's output: the prologue that made the samples counts them (cnt_partials, one triple per
// block of rays), every block here adds the triples up (integers in double: exact in any order) and knows the coefficients
// before it has rendered anything.  Per ray: composite (ray_weights) -> maps -> squared errors (partial sums, for the reported
// losses only) -> d_raw with the maps still in registers.  Expression for expression composite_forward_kernel +
// mapping_loss_forward_kernel + mapping_loss_backward_kernel: the same d_raw bit for bit.  The remaining blocks: the TV backward (tv_backward_body).
constexpr int LOSS_GRAD_BLOCKS = 1024;            // partial sums: LOSS_GRAD_BLOCKS x 8 doubles

__global__ __launch_bounds__(LOSS_THREADS) void composite_loss_grad_tv_kernel(LossK L, const float4* __restrict__ raw,
                                                                              const float* __restrict__ zv,
                                                                              const float* __restrict__ tgt_rgb,
                                                                              const float* __restrict__ tgt_d, int64_t n_rays, int S,
                                                                              float trunc, float sc, float* __restrict__ rgb_map,
                                                                              float* __restrict__ depth_map, double* __restrict__ sums,
                                                                              const double* __restrict__ cnt_partials, int n_cnt,
                                                                              const float* __restrict__ gout, float4* __restrict__ d_raw,
                                                                              int* __restrict__ ray_counts, int nb_loss, TvBackK tv,
                                                                              int64_t n_total) {
    __shared__ double red[LOSS_THREADS / 64];
    __shared__ double cpart[64][4], ctot[4];
    __shared__ float coef[4];
    if ((int)blockIdx.x >= nb_loss) {
        tv_backward_body(tv.feat, tv.P, tv.Cn, tv.scale, nullptr, tv.dfeat, (int)blockIdx.x - nb_loss, (int)gridDim.x - nb_loss);
        return;
    }
    {   // the batch's counts: 0 valid rays, 1 front samples, 2 sdf samples (thread = (value, slice), slices in order)
        const int v = threadIdx.x & 3, q = threadIdx.x >> 2;
        double a = 0.0;
        for (int k = q; k < n_cnt; k += 64) a += cnt_partials[k * 4 + v];
        cpart[q][v] = a;
        __syncthreads();
        if (threadIdx.x < 4) {
            double t = 0.0;
            for (int qq = 0; qq < 64; ++qq) t += cpart[qq][threadIdx.x];
            ctot[threadIdx.x] = t;
        }
        __syncthreads();
        if (threadIdx.x == 0) {             // loss_finalize's coefficients (rfx_common.h), from the same three numbers; n_total:
            // the rays of the whole batch (the counts are the whole batch's), of which this launch renders n_rays
            const double ns = (double)n_total * (double)S;
            const double tot = ctot[1] + ctot[2];
            const float fs_w = (float)(1.0 - ctot[1] / tot), sdf_w = (float)(1.0 - ctot[2] / tot);
            coef[0] = (float)(1.0 / (3.0 * (double)n_total)); coef[1] = (float)(1.0 / ctot[0]);
            coef[2] = (float)(1.0 / ns) * sdf_w; coef[3] = (float)(1.0 / ns) * fs_w;
        }
        __syncthreads();
    }
    const float k_rgb = gout[0] * coef[0], k_dep = gout[1] * coef[1], k_sdf = gout[2] * coef[2], k_fs = gout[3] * coef[3];
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    double a_rgb = 0, a_dep = 0, a_val = 0, a_fs = 0, a_sdf = 0, a_nfs = 0, a_nsdf = 0;
    for (int64_t ray = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); ray < n_rays; ray += (int64_t)nb_loss * wpb) {
        float s[2], z[2];
        float4 rv[2];
        bool vs[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            vs[c] = j < S;
            rv[c] = vs[c] ? raw[ray * S + j] : make_float4(0.f, 0.f, 0.f, 0.f);
            z[c] = vs[c] ? zv[ray * S + j] : 0.f;
            s[c] = rv[c].w;
        }
        const RayW rw = ray_weights(s, z, vs, S, lane, trunc, sc);
        // ---- forward: maps and squared errors (composite_forward_kernel, mapping_loss_forward_kernel)
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, md = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            m0 += rw.w[c] * rv[c].x; m1 += rw.w[c] * rv[c].y; m2 += rw.w[c] * rv[c].z; md += rw.w[c] * z[c];
        }
        m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2); md = wave_sum(md);
        const float d = tgt_d[ray];
        const bool valid = (d > 0.0f) && (d < L.depth_trunc);
        const float mm[3] = {m0, m1, m2};
        if (lane == 0) {
            rgb_map[ray * 3] = m0; rgb_map[ray * 3 + 1] = m1; rgb_map[ray * 3 + 2] = m2; depth_map[ray] = md;
            const float w = (valid || L.rgb_missing_on) ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e = mm[c] * w - tgt_rgb[ray * 3 + c] * w;
                a_rgb += (double)(e * e);
            }
            if (valid) { const float e = md - d; a_dep += (double)(e * e); a_val += 1.0; }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!vs[c]) continue;
            const float zz = z[c], ss = s[c];
            const float front = (zz < (d - L.trunc_loss)) ? 1.0f : 0.0f;
            const float back = (zz > (d + L.trunc_loss)) ? 1.0f : 0.0f;
            const float sm = (1.0f - front) * (1.0f - back) * (d > 0.0f ? 1.0f : 0.0f);
            a_nfs += front; a_nsdf += sm;
            if (valid) {
                const float ef = ss * front - front;
                const float es = (zz + ss * L.trunc_loss) * sm - d * sm;
                a_fs += (double)(ef * ef); a_sdf += (double)(es * es);
            }
        }
        // ---- backward (mapping_loss_backward_kernel)
        const float w2 = (valid || L.rgb_missing_on) ? 1.0f : 0.0f;
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = k_rgb * 2.0f * w2 * (mm[c] * w2 - tgt_rgb[ray * 3 + c] * w2);
        const float gd = valid ? k_dep * 2.0f * (md - d) : 0.0f;
        float gj[2], dot = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            gj[c] = g[0] * rv[c].x + g[1] * rv[c].y + g[2] * rv[c].z + gd * z[c];
            dot += rw.w[c] * gj[c];
        }
        dot = wave_sum(dot);
        int nz = 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!vs[c]) continue;
            float ds = 0.f;
            if (rw.w[c] != 0.0f) {
                const float a = s[c] / trunc;
                const float sp = sigmoidf(a), sm = sigmoidf(-a);
                ds = (gj[c] - dot) / rw.wsum * (sp * sm * (sm - sp) / trunc);
            }
            if (valid) {
                const float front = (z[c] < (d - L.trunc_loss)) ? 1.0f : 0.0f;
                const float back = (z[c] > (d + L.trunc_loss)) ? 1.0f : 0.0f;
                const float smk = (1.0f - front) * (1.0f - back) * (d > 0.0f ? 1.0f : 0.0f);
                ds += k_fs * 2.0f * (s[c] * front - front) * front;
                ds += k_sdf * 2.0f * ((z[c] + s[c] * L.trunc_loss) * smk - d * smk) * (L.trunc_loss * smk);
            }
            const float4 o = make_float4(rw.w[c] * g[0], rw.w[c] * g[1], rw.w[c] * g[2], ds);
            d_raw[ray * S + lane + 64 * c] = o;
            nz += (o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) ? 1 : 0;
        }
        if (ray_counts) {
            nz = (int)wave_sum((float)nz);
            if (lane == 0) ray_counts[ray] = nz;
        }
    }
    a_rgb = block_sum_d(a_rgb, red); a_dep = block_sum_d(a_dep, red); a_val = block_sum_d(a_val, red);
    a_fs = block_sum_d(a_fs, red); a_sdf = block_sum_d(a_sdf, red); a_nfs = block_sum_d(a_nfs, red);
    a_nsdf = block_sum_d(a_nsdf, red);
    if (threadIdx.x == 0) {
        double* o = sums + (size_t)blockIdx.x * 8;
        o[0] = a_rgb; o[1] = a_dep; o[2] = a_val; o[3] = a_fs; o[4] = a_sdf; o[5] = a_nfs; o[6] = a_nsdf; o[7] = 0.0;
    }
}

// ------------------------------------------------------------------ counter-based uniforms
// The reference draws its jitter (torch.rand((n, S)), scene_rep.py:437) and the TV lattice offset (torch.rand(6),
// slam.py:198-203) from torch's generator: two launches per iteration that produce nothing but random numbers.  A BA
// iteration can draw them itself instead (rfx_ba_desc.seed_u): element e of stream `stream` is word 0 of
// Philox4x32-10(counter = (e, stream), key = seed), mapped to [0, 1) with 24 bits like torch's float uniform.
// rfx_uniform_draws fills a buffer with the same numbers (the stage-by-stage issue and the tests use it).
__device__ __forceinline__ uint32_t philox_word0(uint64_t seed, uint32_t stream, uint64_t e) {
    uint32_t c0 = (uint32_t)e, c1 = (uint32_t)(e >> 32), c2 = stream, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ float draw_uniform(uint64_t seed, uint32_t stream, uint64_t e) {
    return (float)(philox_word0(seed, stream, e) >> 8) * 5.9604644775390625e-08f;      // 2^-24
}
constexpr uint32_t DRAW_STREAM_JITTER = 0u, DRAW_STREAM_LATTICE = 1u;

__global__ __launch_bounds__(256) void uniform_draws_kernel(uint64_t seed, uint32_t stream, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = draw_uniform(seed, stream, (uint64_t)i);
}

// TV lattice points (mp_slam/slam.py:198-207): P^3 points at spacing `voxel`, anchored at
// bound_lo + offset with offset = u[0:3] * offset_max + margin, plus a sub-voxel jitter u[3:6], then
// normalised by the bound.  torch's type promotion is reproduced: a float64 bound -> everything in
// double with the jitter; an integer bound -> the jitter's cast to int64 truncates it to 0 and the rest runs in fp32.
struct LatticeK {
    double lo[3], hi[3];
    int f64, normalise, P;
    float voxel, margin;
};

__device__ __forceinline__ void tv_lattice_point(const LatticeK& L, const float u6[6], int64_t i, float xyz[3]) {
    const int c[3] = {(int)(i / ((int64_t)L.P * L.P)), (int)((i / L.P) % L.P), (int)(i % L.P)};
    const double grid = (double)(L.P) * (double)L.voxel;          // (sample_points - 1) * voxel_size
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float out;
        if (L.f64) {
            const double ext = L.hi[d] - L.lo[d];
            const double off = (double)u6[d] * (ext - grid - 2.0 * (double)L.margin) + (double)L.margin;
            const double p = ((double)c[d] + (double)u6[3 + d]) * (double)L.voxel + L.lo[d] + off;
            out = (float)(L.normalise ? (p - L.lo[d]) / ext : p);
        } else {
            const float ext = (float)(L.hi[d] - L.lo[d]);
            const float off = u6[d] * ((ext - (float)grid) - (float)(2.0 * (double)L.margin)) + L.margin;
            const float p = ((float)c[d] * L.voxel + (float)L.lo[d]) + off;
            out = L.normalise ? (p - (float)L.lo[d]) / ext : p;
        }
        xyz[d] = out;
    }
}

__global__ __launch_bounds__(256) void tv_lattice_kernel(LatticeK L, const float* __restrict__ u6, float* __restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)L.P * L.P * L.P;
    if (i >= n) return;
    const float u[6] = {u6[0], u6[1], u6[2], u6[3], u6[4], u6[5]};
    float x[3];
    tv_lattice_point(L, u, i, x);
    pts[i * 3] = x[0]; pts[i * 3 + 1] = x[1]; pts[i * 3 + 2] = x[2];
}

// ------------------------------------------------------------------ distinct random indices
// Replaces python's random.sample(range(N), k) of the ray samplers (model/keyframe.py:33,89;
// mp_slam/mapper.py:396): out[i] = pi(i), i < k, where pi is a keyed pseudo-random permutation of
// [0, N) -- a 4-round Feistel network on the next even power of two, cycle-walked back into range.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int64_t feistel_index(uint64_t seed, int64_t i, int64_t N, int half_bits) {
    const uint32_t mask = (half_bits >= 32) ? 0xffffffffu : ((1u << half_bits) - 1u);
    uint64_t x = (uint64_t)i;
    do {
        uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint32_t key = (uint32_t)(seed >> (16 * round)) ^ (0x9e3779b9u * (round + 1)) ^ (uint32_t)(seed >> 32);
            const uint32_t f = mix32(r ^ key) & mask;
            const uint32_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = ((uint64_t)l << half_bits) | r;
    } while (x >= (uint64_t)N);       // cycle walking: a permutation of [0, 2^(2 half_bits)) restricted to [0, N)
    return (int64_t)x;
}

static int feistel_half_bits(int64_t population) {
    int bits = 1;
    while (bits < 63 && (1LL << bits) < population) ++bits;
    return (bits + 1) / 2;
}

__global__ __launch_bounds__(256) void random_subset_kernel(uint64_t seed, int64_t N, int64_t k, int half_bits,
                                                            int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    out[i] = feistel_index(seed, i, N, half_bits);
}

// the same with the population read on the device: N = *population_dev if that exceeds k, else fallback_population (and
// *used_fallback = 1).  For a caller that must not wait for a count it has just computed on the device: the keyframe store's
// "k rays among those with a valid depth, or among all rays when too few are valid" (model/keyframe.py:37-47).
__global__ __launch_bounds__(256) void random_subset_dev_kernel(uint64_t seed, const int64_t* __restrict__ population_dev,
                                                                int64_t fallback, int64_t k, int64_t* __restrict__ out,
                                                                int32_t* __restrict__ used_fallback) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cnt = *population_dev;
    const bool fb = !(cnt > k);
    const int64_t N = fb ? fallback : cnt;
    if (i == 0 && used_fallback) *used_fallback = fb ? 1 : 0;
    if (i >= k) return;
    int bits = 1;
    while (bits < 63 && (1LL << bits) < N) ++bits;
    out[i] = feistel_index(seed, i, N, (bits + 1) / 2);
}

// ------------------------------------------------------------------ M1: ray batch of one BA iteration
// Replaces the per-iteration host glue of mp_slam/mapper.py:394-409 (sample_global_rays + random.sample of
// current-frame pixels + cat + poses_all[ids] + rays_o / rays_d): one launch draws both index sets,
// gathers the 7-float rays (cam dir 3 | rgb 3 | depth 1), and rotates them by the ray's pose.
struct GatherK {
    const float* kf_rays; int64_t rays_per_kf, kf_population; const int64_t* kf_frame_ids; int keyframe_every;
    const float* cur_rays; int64_t cur_population;
    int64_t n_kf, n_cur; uint64_t seed_kf, seed_cur; int hb_kf, hb_cur;
    const float* poses; int K;
};

__global__ __launch_bounds__(256) void gather_rays_kernel(GatherK g, float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                          float* __restrict__ tgt_rgb, float* __restrict__ tgt_d,
                                                          float* __restrict__ d_cam, int* __restrict__ pose_idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_kf + g.n_cur) return;
    const float* ray;
    int k;
    if (i < g.n_kf) {
        const int64_t idx = feistel_index(g.seed_kf, i, g.kf_population, g.hb_kf);
        ray = g.kf_rays + idx * 7;
        k = (int)(g.kf_frame_ids[idx / g.rays_per_kf] / g.keyframe_every);
        k = ((k % g.K) + g.K) % g.K;
    } else {
        const int64_t idx = feistel_index(g.seed_cur, i - g.n_kf, g.cur_population, g.hb_cur);
        ray = g.cur_rays + idx * 7;
        k = g.K - 1;                                   // the current frame rides on the last pose
    }
    const float dc[3] = {ray[0], ray[1], ray[2]};
    const float* __restrict__ P = g.poses + (size_t)k * 16;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rays_d[i * 3 + r] = P[4 * r] * dc[0] + P[4 * r + 1] * dc[1] + P[4 * r + 2] * dc[2];
        rays_o[i * 3 + r] = P[4 * r + 3];
        tgt_rgb[i * 3 + r] = ray[3 + r];
        d_cam[i * 3 + r] = dc[r];
    }
    tgt_d[i] = ray[6];
    pose_idx[i] = k;
}

// gather_rays_kernel + sample_z_kernel + ray_points_kernel in one launch, wave = ray (the three stages of a BA
// iteration's ray batch are each a few microseconds of launch-bound work).  Same expressions, so the outputs are
// bit-identical to the three separate launches.  block / n_blocks: the block's place among the blocks that run this
// body (the prologue kernel below gives the rest of its grid to other work).  seed_u != 0 (and no u01): the jitter
// uniforms are drawn here (draw_uniform, stream DRAW_STREAM_JITTER, element ray * S + sample).
struct RayOut {
    float *rays_o, *rays_d, *tgt_rgb, *tgt_d, *d_cam; int* pose_idx; float *z_vals, *x01;
};

// cnt (optional, this lane's share): [0] rays with a valid target depth, [1] front samples, [2] sdf samples -- the three
// counts the loss coefficients are made of (get_masks, model/utils.py:170-198; counted as mapping_loss_forward_kernel does)
struct LossCountK { float trunc_loss, depth_trunc; double* partials; };

__device__ __forceinline__ void ray_setup_body(const GatherK& g, const SamplerK& s, const BoxK& box, const float* __restrict__ u01,
                                               uint64_t seed_u, const RayOut& out, int block, int n_blocks, float (*zsh)[MAX_S],
                                               const LossCountK* lcnt = nullptr, double* cnt = nullptr) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = s.n_range_d + s.n_samples_d;
    float* zs = zsh[wv];
    const int64_t n = g.n_kf + g.n_cur;
    for (int64_t i = (int64_t)block * 4 + wv; i < n; i += (int64_t)n_blocks * 4) {
        const float* ray;
        int k;
        if (i < g.n_kf) {
            const int64_t idx = feistel_index(g.seed_kf, i, g.kf_population, g.hb_kf);
            ray = g.kf_rays + idx * 7;
            k = (int)(g.kf_frame_ids[idx / g.rays_per_kf] / g.keyframe_every);
            k = ((k % g.K) + g.K) % g.K;
        } else {
            const int64_t idx = feistel_index(g.seed_cur, i - g.n_kf, g.cur_population, g.hb_cur);
            ray = g.cur_rays + idx * 7;
            k = g.K - 1;
        }
        const float dc[3] = {ray[0], ray[1], ray[2]};
        const float td = ray[6];
        const float* __restrict__ P = g.poses + (size_t)k * 16;
        float o[3], d[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            d[r] = P[4 * r] * dc[0] + P[4 * r + 1] * dc[1] + P[4 * r + 2] * dc[2];
            o[r] = P[4 * r + 3];
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                out.rays_d[i * 3 + r] = d[r]; out.rays_o[i * 3 + r] = o[r]; out.tgt_rgb[i * 3 + r] = ray[3 + r]; out.d_cam[i * 3 + r] = dc[r];
            }
            out.tgt_d[i] = td;
            out.pose_idx[i] = k;
        }
        sample_ray(s, td, zs, lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave are visible to it
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < S; j += 64) {
            float z = zs[j];
            if (s.perturb > 0.0f) {
                if (u01) z = jitter(zs, j, S, u01[i * S + j]);
                else if (seed_u) z = jitter(zs, j, S, draw_uniform(seed_u, DRAW_STREAM_JITTER, (uint64_t)(i * S + j)));
            }
            out.z_vals[i * S + j] = z;
#pragma unroll
            for (int c = 0; c < 3; ++c) out.x01[(i * S + j) * 3 + c] = normalise(box, c, o[c] + d[c] * z);
            if (cnt) {
                const float front = (z < (td - lcnt->trunc_loss)) ? 1.0f : 0.0f;
                const float back = (z > (td + lcnt->trunc_loss)) ? 1.0f : 0.0f;
                const float sm = (1.0f - front) * (1.0f - back) * (td > 0.0f ? 1.0f : 0.0f);
                cnt[1] += front; cnt[2] += sm;
            }
        }
        if (cnt && lane == 0 && (td > 0.0f) && (td < lcnt->depth_trunc)) cnt[0] += 1.0;
        __builtin_amdgcn_wave_barrier();
    }
}

// Everything at the head of a BA iteration that depends on nothing computed in it, in ONE launch (a dependent kernel
// boundary costs 3-5 us on this part, more than most of these stages' work): blocks [0, nb_rays) build the ray batch
// (ray_setup_body), the next nb_stage blocks refresh the decoder's MFMA operand image from the weights the optimizer just
// stepped (stage_weights_kernel's work), the rest lay out the TV lattice and look its features up, thread = (lattice
// point, level) like grid_encode_forward_lp_kernel (tv_lattice_kernel's + that kernel's work; lp_shift = 4, F = 2), and the
// last blocks zero-fill a buffer (the hash gradient, which the end of the iteration accumulates into).
constexpr int PROLOGUE_ZERO = 8192;
struct TvEncK {
    LatticeK L; rfx_grid_desc g; const float* table; const float* u6; float* pts; float* feat;
};

__global__ __launch_bounds__(256) void ba_prologue_kernel(GatherK g, SamplerK s, BoxK box, const float* __restrict__ u01,
                                                          uint64_t seed_u, RayOut out, int nb_rays, FieldK f,
                                                          float* __restrict__ staged, int nb_stage, TvEncK tv, int nb_tv,
                                                          float* __restrict__ zero, int64_t zero_floats, LossCountK lcnt) {
    __shared__ float zsh[4][MAX_S];
    __shared__ double cred[4];
    // dispatch order = block index: the ray batch first (a chain of dependent latencies per ray: the launch's critical
    // path; with the lattice's lookups ahead of it the launch took 25 us instead of 20)
    const int b = blockIdx.x;
    if (b < nb_rays) {
        if (lcnt.partials) {
            double cnt[3] = {0.0, 0.0, 0.0};
            ray_setup_body(g, s, box, u01, seed_u, out, b, nb_rays, zsh, &lcnt, cnt);
            const double c0 = block_sum_d(cnt[0], cred), c1 = block_sum_d(cnt[1], cred), c2 = block_sum_d(cnt[2], cred);
            if (threadIdx.x == 0) { double* o = lcnt.partials + (size_t)b * 4; o[0] = c0; o[1] = c1; o[2] = c2; o[3] = 0.0; }
        } else {
            ray_setup_body(g, s, box, u01, seed_u, out, b, nb_rays, zsh);
        }
    } else if (b < nb_rays + nb_stage) {
        const int i = (b - nb_rays) * 256 + threadIdx.x;
        if (i < ALL_SLOTS * 64) staged[i] = staged_weight(f, i >> 6, i & 63);
    } else if (b >= nb_rays + nb_stage + nb_tv) {
        // zero-fill (the hash-gradient buffer the iteration's last kernel accumulates into): PROLOGUE_ZERO floats per block
        const int64_t base = (int64_t)(b - nb_rays - nb_stage - nb_tv) * PROLOGUE_ZERO;
        const int64_t end = min(zero_floats, base + PROLOGUE_ZERO);
        if ((((uintptr_t)zero) & 15) == 0) {
            for (int64_t i = base + (int64_t)threadIdx.x * 4; i + 3 < end; i += 256 * 4) *reinterpret_cast<float4*>(zero + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t i = base + ((end - base) & ~(int64_t)3) + threadIdx.x; i < end; i += 256) zero[i] = 0.f;
        } else {
            for (int64_t i = base + threadIdx.x; i < end; i += 256) zero[i] = 0.f;
        }
    } else {
        float* us = zsh[0];
        if (threadIdx.x < 6) us[threadIdx.x] = tv.u6 ? tv.u6[threadIdx.x] : draw_uniform(seed_u, DRAW_STREAM_LATTICE, (uint64_t)threadIdx.x);
        __syncthreads();
        const int64_t gid = (int64_t)(b - nb_rays - nb_stage) * 256 + threadIdx.x;
        const int64_t p = gid >> 4;
        const int l = (int)(gid & 15);
        if (p >= (int64_t)tv.L.P * tv.L.P * tv.L.P || l >= tv.g.n_levels) return;
        const float u[6] = {us[0], us[1], us[2], us[3], us[4], us[5]};
        float x[3];
        tv_lattice_point(tv.L, u, p, x);
        if (l == 0) { tv.pts[p * 3] = x[0]; tv.pts[p * 3 + 1] = x[1]; tv.pts[p * 3 + 2] = x[2]; }
        reinterpret_cast<float2*>(tv.feat + p * (int64_t)(tv.g.n_levels * 2))[l] = lookup2(tv.table, get_level(tv.g, l), x);
    }
}

// dL/dposes[k] (rows 0..2): rotation part += g_d (x) d_cam, translation column += g_o; block = pose,
// fixed-order block reduction (deterministic).
__global__ __launch_bounds__(256) void pose_grad_kernel(const float* __restrict__ g_o, const float* __restrict__ g_d,
                                                        const float* __restrict__ d_cam, const int* __restrict__ pose_idx,
                                                        int64_t n, float* __restrict__ dposes) {
    __shared__ float red[256][13];
    const int k = blockIdx.x;
    float a[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) a[q] = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        if (pose_idx[i] != k) continue;
        const float dc[3] = {d_cam[i * 3], d_cam[i * 3 + 1], d_cam[i * 3 + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float gd = g_d ? g_d[i * 3 + r] : 0.f;
            a[4 * r] += gd * dc[0]; a[4 * r + 1] += gd * dc[1]; a[4 * r + 2] += gd * dc[2];
            a[4 * r + 3] += g_o ? g_o[i * 3 + r] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) red[threadIdx.x][q] = a[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int q = 0; q < 12; ++q) red[threadIdx.x][q] += red[threadIdx.x + s][q];
        __syncthreads();
    }
    if (threadIdx.x < 16) dposes[(size_t)k * 16 + threadIdx.x] = threadIdx.x < 12 ? red[0][threadIdx.x] : 0.f;
}

static int make_sampler(const rfx_sampler_desc* d, SamplerK* k) {
    if (!d) return RFX_ERR_ARG;
    if (d->n_range_d < 1 || d->n_samples_d < 0) return RFX_ERR_ARG;
    if (d->n_range_d + d->n_samples_d > MAX_S) return RFX_ERR_UNSUPPORTED;
    k->near = d->near; k->far = d->far; k->range_d = d->range_d; k->perturb = d->perturb;
    k->n_range_d = d->n_range_d; k->n_samples_d = d->n_samples_d;
    return RFX_OK;
}

static BoxK make_box(const double bbox[6], int f64) {
    BoxK b;
    for (int i = 0; i < 3; ++i) { b.lo[i] = bbox[2 * i]; b.hi[i] = bbox[2 * i + 1]; b.inv[i] = 1.0 / (bbox[2 * i + 1] - bbox[2 * i]);
                                  b.lo32[i] = (float)bbox[2 * i]; b.ext32[i] = (float)bbox[2 * i + 1] - (float)bbox[2 * i]; }
    b.f64 = f64 ? 1 : 0;
    return b;
}

static inline int ray_grid(int64_t n_rays) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n_rays + 3) / 4, 256 * 8));
}

}  // namespace rfx

using namespace rfx;

extern "C" {

int rfx_sample_z(const rfx_sampler_desc* s, const float* target_d, const float* u01, int64_t n_rays, float* z_vals,
                 rfx_stream stream) {
    if (n_rays == 0) return RFX_OK;
    SamplerK k;
    int rc = make_sampler(s, &k);
    if (rc) return rc;
    if (!target_d || !z_vals || n_rays < 0) return RFX_ERR_ARG;
    if (n_rays == 0) return RFX_OK;
    hipLaunchKernelGGL(sample_z_kernel, dim3(ray_grid(n_rays)), dim3(256), 0, as_stream(stream), k, target_d, u01, n_rays, z_vals);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays, int S,
                   const double bbox[6], int bbox_f64, float* x01, rfx_stream stream) {
    if (n_rays == 0) return RFX_OK;
    if (!rays_o || !rays_d || !z_vals || !bbox || !x01 || n_rays < 0 || S <= 0) return RFX_ERR_ARG;
    if (n_rays == 0) return RFX_OK;
    const int64_t n = n_rays * S;
    hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d,
                       z_vals, n_rays, S, make_box(bbox, bbox_f64), x01);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_composite_forward(const float* raw4, const float* z_vals, int64_t n_rays, int S, float trunc, float sc_factor,
                          float* rgb, float* depth, float* weights, rfx_stream stream) {
    if (n_rays == 0) return RFX_OK;
    if (!raw4 || !z_vals || !rgb || !depth || n_rays < 0 || S <= 0 || !(trunc > 0.f)) return RFX_ERR_ARG;
    if (S > MAX_S) return RFX_ERR_UNSUPPORTED;
    if (n_rays == 0) return RFX_OK;
    hipLaunchKernelGGL(composite_forward_kernel, dim3(ray_grid(n_rays)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(raw4), z_vals, n_rays, S, trunc, sc_factor, rgb, depth, weights);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_composite_backward(const float* raw4, const float* z_vals, int64_t n_rays, int S, float trunc, float sc_factor,
                           const float* d_rgb, const float* d_depth, float* d_raw4, rfx_stream stream) {
    if (n_rays == 0) return RFX_OK;
    if (!raw4 || !z_vals || !d_rgb || !d_depth || !d_raw4 || n_rays < 0 || S <= 0 || !(trunc > 0.f)) return RFX_ERR_ARG;
    if (S > MAX_S) return RFX_ERR_UNSUPPORTED;
    if (n_rays == 0) return RFX_OK;
    hipLaunchKernelGGL(composite_backward_kernel, dim3(ray_grid(n_rays)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(raw4), z_vals, n_rays, S, trunc, sc_factor, d_rgb, d_depth,
                       reinterpret_cast<float4*>(d_raw4));
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_render_rays(const rfx_field_desc* f, const rfx_sampler_desc* s, const float* rays_o, const float* rays_d,
                    const float* target_d, const float* u01, int64_t n_rays, const double bbox[6], int bbox_f64,
                    float sc_factor, float* rgb, float* depth, rfx_stream stream) {
    if (n_rays == 0) return RFX_OK;
    FieldK fk;
    int rc = make_fieldk(f, &fk);
    if (rc) return rc;
    SamplerK sk;
    rc = make_sampler(s, &sk);
    if (rc) return rc;
    if (!rays_o || !rays_d || !target_d || !bbox || !rgb || !depth || n_rays < 0) return RFX_ERR_ARG;
    if (n_rays == 0) return RFX_OK;
    const bool one_pass = sk.n_range_d + sk.n_samples_d <= 64;
    auto kern = fk.pos_fp16 ? (one_pass ? render_rays_kernel<true, 1> : render_rays_kernel<true, 2>)
                            : (one_pass ? render_rays_kernel<false, 1> : render_rays_kernel<false, 2>);
    hipLaunchKernelGGL(kern, dim3(ray_grid(n_rays)), dim3(256), 0, as_stream(stream), fk, sk,
                       make_box(bbox, bbox_f64), rays_o, rays_d, target_d, u01, n_rays, sc_factor, rgb, depth);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_mapping_loss_forward(const float* raw4, const float* z_vals, const float* rgb_map, const float* depth_map,
                             const float* target_rgb, const float* target_d, int64_t n_rays, int S, float trunc_loss,
                             float depth_trunc, int rgb_missing_on, double* sums8, float* losses4, float* coef4,
                             rfx_stream stream) {
    if (n_rays == 0) return RFX_OK;
    if (!raw4 || !z_vals || !rgb_map || !depth_map || !target_rgb || !target_d || !sums8 || !losses4 || !coef4) return RFX_ERR_ARG;
    if (n_rays < 0 || S <= 0) return RFX_ERR_ARG;
    hipStream_t st = as_stream(stream);
    LossK L; L.trunc_loss = trunc_loss; L.depth_trunc = depth_trunc; L.rgb_missing_on = rgb_missing_on ? 1 : 0;
    const int blocks = (int)std::min<int64_t>((n_rays + LOSS_THREADS / 64 - 1) / (LOSS_THREADS / 64), LOSS_BLOCKS);
    hipLaunchKernelGGL(mapping_loss_forward_kernel, dim3(blocks), dim3(LOSS_THREADS), 0, st, L,
                       reinterpret_cast<const float4*>(raw4), z_vals, rgb_map, depth_map, target_rgb, target_d, n_rays, S, sums8);
    RFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mapping_loss_finalize_kernel, dim3(1), dim3(256), 0, st, sums8, blocks, n_rays, S, losses4, coef4);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// ---- the forward in two halves, for a ray batch that is spread over several GPUs: each rank reduces its rays to the
// eight sums, the sums are added across ranks (all-reduce of 64 bytes), and every rank forms losses and coefficients
// of the WHOLE batch from them.
__global__ __launch_bounds__(256) void mapping_loss_total_kernel(const double* __restrict__ partial, int n_partials,
                                                                 double* __restrict__ total8) {
    __shared__ double part[32][8];
    const int v = threadIdx.x & 7, q = threadIdx.x >> 3;
    double a = 0.0;
    for (int k = q; k < n_partials; k += 32) a += partial[k * 8 + v];
    part[q][v] = a;
    __syncthreads();
    if (threadIdx.x < 8) {
        double t = 0.0;
        for (int qq = 0; qq < 32; ++qq) t += part[qq][threadIdx.x];
        total8[threadIdx.x] = t;
    }
}

int rfx_mapping_loss_sums(const float* raw4, const float* z_vals, const float* rgb_map, const float* depth_map,
                          const float* target_rgb, const float* target_d, int64_t n_rays, int S, float trunc_loss,
                          float depth_trunc, int rgb_missing_on, double* scratch, double* total8, rfx_stream stream) {
    if (!scratch || !total8 || n_rays < 0 || S <= 0) return RFX_ERR_ARG;
    hipStream_t st = as_stream(stream);
    if (n_rays == 0) {
        RFX_HIP_TRY(hipMemsetAsync(total8, 0, 8 * sizeof(double), st));
        return RFX_OK;
    }
    if (!raw4 || !z_vals || !rgb_map || !depth_map || !target_rgb || !target_d) return RFX_ERR_ARG;
    LossK L; L.trunc_loss = trunc_loss; L.depth_trunc = depth_trunc; L.rgb_missing_on = rgb_missing_on ? 1 : 0;
    const int blocks = (int)std::min<int64_t>((n_rays + LOSS_THREADS / 64 - 1) / (LOSS_THREADS / 64), LOSS_BLOCKS);
    hipLaunchKernelGGL(mapping_loss_forward_kernel, dim3(blocks), dim3(LOSS_THREADS), 0, st, L,
                       reinterpret_cast<const float4*>(raw4), z_vals, rgb_map, depth_map, target_rgb, target_d, n_rays, S, scratch);
    RFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mapping_loss_total_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, total8);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_mapping_loss_finalize(const double* total8, int64_t n_rays_total, int S, float* losses4, float* coef4, rfx_stream stream) {
    if (!total8 || !losses4 || !coef4 || n_rays_total <= 0 || S <= 0) return RFX_ERR_ARG;
    hipLaunchKernelGGL(mapping_loss_finalize_kernel, dim3(1), dim3(256), 0, as_stream(stream), total8, 1, n_rays_total, S, losses4, coef4);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_mapping_loss_backward(const float* raw4, const float* z_vals, const float* rgb_map, const float* depth_map,
                              const float* target_rgb, const float* target_d, int64_t n_rays, int S, float trunc,
                              float sc_factor, float trunc_loss, float depth_trunc, int rgb_missing_on, const float* coef4,
                              const float* gout4, const float* g_rgb_map, const float* g_depth_map, float* d_raw4,
                              rfx_stream stream) {
    if (n_rays == 0) return RFX_OK;
    if (!raw4 || !z_vals || !rgb_map || !depth_map || !target_rgb || !target_d || !coef4 || !gout4 || !d_raw4) return RFX_ERR_ARG;
    if (n_rays < 0 || S <= 0 || !(trunc > 0.f)) return RFX_ERR_ARG;
    if (S > MAX_S) return RFX_ERR_UNSUPPORTED;
    LossK L; L.trunc_loss = trunc_loss; L.depth_trunc = depth_trunc; L.rgb_missing_on = rgb_missing_on ? 1 : 0;
    hipLaunchKernelGGL(mapping_loss_backward_kernel<false>, dim3(ray_grid(n_rays)), dim3(256), 0, as_stream(stream), L,
                       reinterpret_cast<const float4*>(raw4), z_vals, rgb_map, depth_map, target_rgb, target_d, n_rays, S, trunc,
                       sc_factor, coef4, gout4, g_rgb_map, g_depth_map, reinterpret_cast<float4*>(d_raw4), (const double*)nullptr, 0,
                       (float*)nullptr, (int*)nullptr);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tv_forward(const float* feat, int P, int C, double* sum1, rfx_stream stream) {
    if (!feat || !sum1 || P <= 0 || C <= 0) return RFX_ERR_ARG;
    hipStream_t st = as_stream(stream);
    RFX_HIP_TRY(hipMemsetAsync(sum1, 0, sizeof(double), st));
    const int64_t total = (int64_t)P * P * P * C;
    hipLaunchKernelGGL(tv_forward_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 256)), dim3(256), 0, st, feat, P, C, sum1);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tv_backward(const float* feat, int P, int C, float scale, const float* gscale_dev, float* dfeat, rfx_stream stream) {
    if (!feat || !dfeat || P <= 0 || C <= 0) return RFX_ERR_ARG;
    const int64_t total = (int64_t)P * P * P * C;
    hipLaunchKernelGGL(tv_backward_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0,
                       as_stream(stream), feat, P, C, scale, gscale_dev, dfeat);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tv_lattice(const float* u6, int P, float voxel, float margin, const double bbox[6], int bbox_f64, int normalise,
                   float* pts, rfx_stream stream) {
    if (!u6 || !bbox || !pts || P <= 0 || !(voxel > 0.f)) return RFX_ERR_ARG;
    LatticeK L;
    for (int d = 0; d < 3; ++d) { L.lo[d] = bbox[2 * d]; L.hi[d] = bbox[2 * d + 1]; }
    L.f64 = bbox_f64 ? 1 : 0; L.normalise = normalise ? 1 : 0; L.P = P; L.voxel = voxel; L.margin = margin;
    const int64_t n = (int64_t)P * P * P;
    hipLaunchKernelGGL(tv_lattice_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), L, u6, pts);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_uniform_draws(uint64_t seed, int stream_id, int64_t n, float* out, rfx_stream stream) {
    if (n == 0) return RFX_OK;
    if (!out || n < 0 || stream_id < 0) return RFX_ERR_ARG;
    hipLaunchKernelGGL(uniform_draws_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), seed,
                       (uint32_t)stream_id, n, out);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_random_subset(uint64_t seed, int64_t population, int64_t k, int64_t* out, rfx_stream stream) {
    if (k == 0) return RFX_OK;
    if (!out || population <= 0 || k < 0 || k > population) return RFX_ERR_ARG;
    const int half_bits = feistel_half_bits(population);
    if (half_bits > 31) return RFX_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(random_subset_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, as_stream(stream), seed,
                       population, k, half_bits, out);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_random_subset_dev(uint64_t seed, const int64_t* population_dev, int64_t fallback_population, int64_t k, int64_t* out,
                          int32_t* used_fallback_dev, rfx_stream stream) {
    if (k == 0) return RFX_OK;
    if (!out || !population_dev || fallback_population <= 0 || k < 0 || k > fallback_population) return RFX_ERR_ARG;
    if (feistel_half_bits(fallback_population) > 31) return RFX_ERR_UNSUPPORTED;       // (*population_dev <= fallback is the caller's contract)
    hipLaunchKernelGGL(random_subset_dev_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, as_stream(stream), seed,
                       population_dev, fallback_population, k, out, used_fallback_dev);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

static int make_gather(const float* kf_rays, int64_t rays_per_kf, int64_t num_kf, const int64_t* kf_frame_ids, int keyframe_every,
                       const float* cur_rays, int64_t cur_population, int64_t n_kf_samples, int64_t n_cur, uint64_t seed_kf,
                       uint64_t seed_cur, const float* poses16, int K, GatherK* out) {
    if (n_kf_samples < 0 || n_cur < 0 || K <= 0 || keyframe_every <= 0 || !poses16) return RFX_ERR_ARG;
    GatherK g{};
    g.kf_rays = kf_rays; g.rays_per_kf = rays_per_kf; g.kf_population = rays_per_kf * num_kf; g.kf_frame_ids = kf_frame_ids;
    g.keyframe_every = keyframe_every; g.cur_rays = cur_rays; g.cur_population = cur_population;
    g.n_kf = n_kf_samples; g.n_cur = n_cur; g.seed_kf = seed_kf; g.seed_cur = seed_cur; g.poses = poses16; g.K = K;
    if (n_kf_samples > 0) {
        if (!kf_rays || !kf_frame_ids || rays_per_kf <= 0 || num_kf <= 0 || n_kf_samples > g.kf_population) return RFX_ERR_ARG;
        g.hb_kf = feistel_half_bits(g.kf_population);
        if (g.hb_kf > 31) return RFX_ERR_UNSUPPORTED;
    }
    if (n_cur > 0) {
        if (!cur_rays || cur_population <= 0 || n_cur > cur_population) return RFX_ERR_ARG;
        g.hb_cur = feistel_half_bits(cur_population);
        if (g.hb_cur > 31) return RFX_ERR_UNSUPPORTED;
    }
    *out = g;
    return RFX_OK;
}

int rfx_gather_rays(const float* kf_rays, int64_t rays_per_kf, int64_t num_kf, const int64_t* kf_frame_ids, int keyframe_every,
                    const float* cur_rays, int64_t cur_population, int64_t n_kf_samples, int64_t n_cur, uint64_t seed_kf,
                    uint64_t seed_cur, const float* poses16, int K, float* rays_o, float* rays_d, float* target_rgb,
                    float* target_d, float* d_cam, int32_t* pose_idx, rfx_stream stream) {
    const int64_t n = n_kf_samples + n_cur;
    if (n == 0) return RFX_OK;
    if (!rays_o || !rays_d || !target_rgb || !target_d || !d_cam || !pose_idx) return RFX_ERR_ARG;
    GatherK g;
    const int rc = make_gather(kf_rays, rays_per_kf, num_kf, kf_frame_ids, keyframe_every, cur_rays, cur_population, n_kf_samples,
                               n_cur, seed_kf, seed_cur, poses16, K, &g);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), g, rays_o, rays_d,
                       target_rgb, target_d, d_cam, pose_idx);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_pose_grad(const float* g_o, const float* g_d, const float* d_cam, const int32_t* pose_idx, int64_t n, int K,
                  float* dposes16, rfx_stream stream) {
    if (K <= 0 || !dposes16 || n < 0) return RFX_ERR_ARG;
    if (n > 0 && (!d_cam || !pose_idx)) return RFX_ERR_ARG;
    hipLaunchKernelGGL(pose_grad_kernel, dim3((unsigned)K), dim3(256), 0, as_stream(stream), g_o, g_d, d_cam, pose_idx, n, dposes16);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"

// ---- fused forms used by rfx_ba_forward_backward (internal: declared in rfx_common.h) --------------------------------
namespace rfx {

// rfx_gather_rays + rfx_sample_z + rfx_ray_points, and beside them in the same launch (ba_prologue_kernel)
// rfx_field_stage_weights (field with a `staged` image: refreshed) and rfx_tv_lattice + rfx_grid_encode_forward of the
// lattice (tv_pts / tv_feat given).  seed_u != 0: the jitter and lattice uniforms are drawn in the kernel, u01 / u6 are
// not read.
int ba_prologue(const float* kf_rays, int64_t rays_per_kf, int64_t num_kf, const int64_t* kf_frame_ids, int keyframe_every,
                const float* cur_rays, int64_t cur_population, int64_t n_kf_samples, int64_t n_cur, uint64_t seed_kf,
                uint64_t seed_cur, const float* poses16, int K, const rfx_sampler_desc* sampler, const float* u01, uint64_t seed_u,
                const double bbox[6], int bbox_f64, float* rays_o, float* rays_d, float* target_rgb, float* target_d,
                float* d_cam, int32_t* pose_idx, float* z_vals, float* x01, const rfx_field_desc* field, const float* u6, int tv_P,
                float tv_voxel, float tv_margin, int tv_normalise, float* tv_pts, float* tv_feat, float* zero, int64_t zero_floats,
                float trunc_loss, float depth_trunc, double* count_partials, int* n_count_partials, rfx_stream stream,
                const rfx_grid_desc* tv_grid, int parts) {
    // parts: the launch is made of independent block ranges; a level-partitioned iteration issues the ray batch first (its
    // features go into the all-to-all) and the lattice + zero-fill while that exchange is in flight (rfx_ba_shard_lookup_tv)
    if (!(parts & 2)) { tv_pts = nullptr; tv_feat = nullptr; zero = nullptr; zero_floats = 0; }
    if (n_count_partials) *n_count_partials = 0;
    const int64_t n = n_kf_samples + n_cur;
    if (n == 0) return RFX_OK;
    if (!rays_o || !rays_d || !target_rgb || !target_d || !d_cam || !pose_idx || !z_vals || !x01 || !bbox) return RFX_ERR_ARG;
    GatherK g;
    int rc = make_gather(kf_rays, rays_per_kf, num_kf, kf_frame_ids, keyframe_every, cur_rays, cur_population, n_kf_samples, n_cur,
                         seed_kf, seed_cur, poses16, K, &g);
    if (rc) return rc;
    SamplerK k;
    rc = make_sampler(sampler, &k);
    if (rc) return rc;
    FieldK fk = {};
    int nb_stage = 0;
    float* staged = nullptr;
    if (field && field->staged && (parts & 1)) {
        rc = make_fieldk(field, &fk);
        if (rc) return rc;
        staged = const_cast<float*>(field->staged);
        fk.staged = nullptr;
        nb_stage = (ALL_SLOTS * 64 + 255) / 256;
    }
    TvEncK tv = {};
    int nb_tv = 0;
    if (tv_pts) {
        if (!field || !tv_feat || tv_P <= 0 || !(tv_voxel > 0.f) || (!u6 && !seed_u)) return RFX_ERR_ARG;
        const rfx_grid_desc& tg = tv_grid ? *tv_grid : field->hash;        // (a level-partitioned table: the own levels only)
        if (tg.n_feat != 2 || tg.n_levels < 1 || tg.n_levels > 16) return RFX_ERR_UNSUPPORTED;
        for (int d = 0; d < 3; ++d) { tv.L.lo[d] = bbox[2 * d]; tv.L.hi[d] = bbox[2 * d + 1]; }
        tv.L.f64 = bbox_f64 ? 1 : 0; tv.L.normalise = tv_normalise ? 1 : 0; tv.L.P = tv_P; tv.L.voxel = tv_voxel; tv.L.margin = tv_margin;
        tv.g = tg; tv.table = field->hash_table; tv.u6 = seed_u ? nullptr : u6; tv.pts = tv_pts; tv.feat = tv_feat;
        nb_tv = (int)(((int64_t)tv_P * tv_P * tv_P * 16 + 255) / 256);
    }
    const RayOut out = {rays_o, rays_d, target_rgb, target_d, d_cam, pose_idx, z_vals, x01};
    const int nb_rays = (parts & 1) ? ray_grid(n) : 0;
    if (zero_floats < 0 || (zero_floats > 0 && !zero)) return RFX_ERR_ARG;
    const int64_t nb_zero = zero ? (zero_floats + PROLOGUE_ZERO - 1) / PROLOGUE_ZERO : 0;
    if (nb_rays + nb_stage + nb_tv + nb_zero > 0x7fffffff) return RFX_ERR_UNSUPPORTED;
    if (nb_rays + nb_stage + nb_tv + nb_zero == 0) return RFX_OK;
    hipLaunchKernelGGL(ba_prologue_kernel, dim3((unsigned)(nb_rays + nb_stage + nb_tv + nb_zero)), dim3(256), 0, as_stream(stream), g, k,
                       make_box(bbox, bbox_f64), seed_u ? nullptr : u01, seed_u, out, nb_rays, fk, staged, nb_stage, tv, nb_tv, zero,
                       zero_floats, LossCountK{trunc_loss, depth_trunc, count_partials});
    if (count_partials && n_count_partials && (parts & 1)) *n_count_partials = nb_rays;
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int ba_count_partials(int64_t n_rays) { return n_rays > 0 ? ray_grid(n_rays) : 0; }

// rfx_composite_forward + rfx_mapping_loss_forward + rfx_mapping_loss_backward in one launch (composite_loss_grad_tv_kernel), for a batch whose three
// loss counts the prologue left in count_partials.  sums: LOSS_GRAD_BLOCKS x 8 doubles; *n_partials of them are written and
// wait for a finalize (loss_finalize: the selection's launch takes it along, or loss_finalize_launch below).
int composite_loss_grad(const float* raw4, const float* z_vals, const float* target_rgb, const float* target_d, int64_t n_rays, int S,
                        float trunc, float sc_factor, float trunc_loss, float depth_trunc, int rgb_missing_on, float* rgb_map,
                        float* depth_map, double* sums, int* n_partials, const double* count_partials, int n_count_partials,
                        const float* gout4, float* d_raw4, int32_t* ray_counts, const float* tv_feat, int tv_P, int tv_C, float tv_scale,
                        float* tv_dfeat, rfx_stream stream, int64_t n_rays_total) {
    *n_partials = 0;
    if (n_rays_total < n_rays) n_rays_total = n_rays;
    if (n_rays == 0) return RFX_OK;
    if (!raw4 || !z_vals || !rgb_map || !depth_map || !target_rgb || !target_d || !sums || !count_partials || !gout4 || !d_raw4)
        return RFX_ERR_ARG;
    if (n_rays < 0 || S <= 0 || !(trunc > 0.f) || n_count_partials <= 0) return RFX_ERR_ARG;
    if (S > MAX_S) return RFX_ERR_UNSUPPORTED;
    if (tv_dfeat && (!tv_feat || tv_P <= 0 || tv_C <= 0)) return RFX_ERR_ARG;
    LossK L; L.trunc_loss = trunc_loss; L.depth_trunc = depth_trunc; L.rgb_missing_on = rgb_missing_on ? 1 : 0;
    const int blocks = (int)std::min<int64_t>((n_rays + LOSS_THREADS / 64 - 1) / (LOSS_THREADS / 64), LOSS_GRAD_BLOCKS);
    TvBackK tv = {tv_feat, tv_P, tv_C, tv_scale, tv_dfeat};
    int nb_tv = 0;
    if (tv_dfeat) nb_tv = (int)std::min<int64_t>(((int64_t)tv_P * tv_P * tv_P * tv_C + 255) / 256, 2048);
    hipLaunchKernelGGL(composite_loss_grad_tv_kernel, dim3(blocks + nb_tv), dim3(LOSS_THREADS), 0, as_stream(stream), L,
                       reinterpret_cast<const float4*>(raw4), z_vals, target_rgb, target_d, n_rays, S, trunc, sc_factor, rgb_map, depth_map,
                       sums, count_partials, n_count_partials, gout4, reinterpret_cast<float4*>(d_raw4), ray_counts, blocks, tv,
                       n_rays_total);
    RFX_LAUNCH_CHECK();
    *n_partials = blocks;
    return RFX_OK;
}

int loss_finalize_launch(const double* sums, int n_partials, int64_t n_rays, int S, float* lc8, rfx_stream stream) {
    if (!sums || !lc8 || n_partials <= 0 || n_rays <= 0 || S <= 0) return RFX_ERR_ARG;
    hipLaunchKernelGGL(mapping_loss_finalize_kernel, dim3(1), dim3(256), 0, as_stream(stream), sums, n_partials, n_rays, S, lc8, lc8 + 4);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // namespace rfx
