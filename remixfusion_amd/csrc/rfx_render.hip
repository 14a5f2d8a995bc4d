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
struct BoxK { double lo[3], hi[3]; int f64; };

__device__ __forceinline__ float normalise(const BoxK& b, int d, float p) {
    if (b.f64) return (float)(((double)p - b.lo[d]) / (b.hi[d] - b.lo[d]));
    const float lo = (float)b.lo[d], hi = (float)b.hi[d];
    return (p - lo) / (hi - lo);
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
__global__ __launch_bounds__(256, FWD_WAVES) void render_rays_kernel(FieldK f, SamplerK s, BoxK box,
                                                                    const float* __restrict__ rays_o,
                                                                    const float* __restrict__ rays_d,
                                                                    const float* __restrict__ target_d,
                                                                    const float* __restrict__ u01, int64_t n_rays,
                                                                    float sc, float* __restrict__ rgb,
                                                                    float* __restrict__ depth) {
    __shared__ float wl[FWD_SLOTS * 64];
    __shared__ float zsh[4][MAX_S];
    stage_weights(f, wl, FWD_SLOTS);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = s.n_range_d + s.n_samples_d;
    float* zs = zsh[wv];
    for (int64_t ray = (int64_t)blockIdx.x * 4 + wv; ray < n_rays; ray += (int64_t)gridDim.x * 4) {
        sample_ray(s, target_d[ray], zs, lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
        const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
        float sdf[2], z[2];
        float4 rv[2];
        bool valid[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = lane + 64 * c;
            valid[c] = j < S;
            z[c] = 0.f; sdf[c] = 0.f; rv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
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
            mlp_forward_123<false>(f, x, wl, lane, e, m);
            float raw[4];
            mlp_forward_4(wl, lane, e, m, raw);
            rv[c] = make_float4(raw[0], raw[1], raw[2], raw[3]);
            sdf[c] = raw[3];
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
    for (int i = 0; i < 3; ++i) { b.lo[i] = bbox[2 * i]; b.hi[i] = bbox[2 * i + 1]; }
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
    hipLaunchKernelGGL(render_rays_kernel, dim3(ray_grid(n_rays)), dim3(256), 0, as_stream(stream), fk, sk,
                       make_box(bbox, bbox_f64), rays_o, rays_d, target_d, u01, n_rays, sc_factor, rgb, depth);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"
