// rfx_field_device.h -- device building blocks of the residual neural field on gfx950:
// tiny-cuda-nn-compatible multi-resolution grid lookup (hash / dense), OneBlob, and the
// fp32-MFMA helpers of the fused MLP.  Conventions: see oracle/field_oracle.py (same contract).
#pragma once
#include "rfx_common.h"
#include <hip/hip_fp16.h>

namespace rfx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------ grid lookup
struct Level {
    float scale;
    unsigned res, size, offset, hashed;
};

__device__ __forceinline__ Level get_level(const rfx_grid_desc& g, int l) {
    Level L;
    L.scale = g.scale[l]; L.res = g.res[l]; L.size = g.size[l]; L.offset = g.offset[l]; L.hashed = g.hashed[l];
    return L;
}

// rare path: a dense-level vertex outside the table (points outside the unit cube) wraps modulo the
// level size; kept out of line so the ~40-instruction u32 modulo is not replicated at every corner.
__device__ __noinline__ unsigned wrap_index(unsigned idx, unsigned size) { return idx % size; }

// tiny-cuda-nn grid_index<3>.  Its stride walk (`for dim: if stride <= size: index += g*stride`) only
// ever stops early on levels that end up hashed, so: hashed level -> coherent prime hash, size is a
// power of two -> mask; dense level -> full x + y*res + z*res^2, wrapped modulo size when out of range.
// Everything wraps at 32 bits.
__device__ __forceinline__ unsigned grid_index(const Level& L, unsigned gx, unsigned gy, unsigned gz) {
    unsigned idx;
    if (L.hashed) {
        idx = (gx ^ (gy * 2654435761u) ^ (gz * 805459861u)) & (L.size - 1u);
    } else {
        idx = gx + (gy + gz * L.res) * L.res;
        if (idx >= L.size) idx = wrap_index(idx, L.size);
    }
#if defined(FIELD_DBG) && FIELD_DBG == 1
    idx &= 7;                            // timing experiment: all lanes hit the same line
#endif
    return idx;
}

struct Cell {            // per (point, level): base vertex and fractional offsets
    unsigned g[3];
    float f[3];
};

__device__ __forceinline__ Cell locate(const Level& L, const float x[3]) {
    Cell c;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float pos = fmaf(L.scale, x[d], 0.5f);
        float fl = floorf(pos);
        c.g[d] = (unsigned)(int)fl;
        c.f[d] = pos - fl;
    }
    return c;
}

__device__ __forceinline__ float corner_weight(const Cell& c, int corner) {
    float w = 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) w *= ((corner >> d) & 1) ? c.f[d] : (1.0f - c.f[d]);
    return w;
}

__device__ __forceinline__ unsigned corner_index(const Level& L, const Cell& c, int corner) {
    return grid_index(L, c.g[0] + (corner & 1), c.g[1] + ((corner >> 1) & 1), c.g[2] + ((corner >> 2) & 1));
}

// all eight corner indices of a cell.  One level-uniform branch picks the hash; a dense cell that lies
// entirely inside the grid (every point inside the unit cube) needs one multiply-add chain and seven adds,
// no per-corner range checks; border / outside cells take the generic wrapped form.
__device__ __forceinline__ void corner_indices(const Level& L, const Cell& c, unsigned idx[8]) {
    if (L.hashed) {
        const unsigned m = L.size - 1u;
        const unsigned x0 = c.g[0], x1 = c.g[0] + 1u;
        const unsigned y0 = c.g[1] * 2654435761u, y1 = y0 + 2654435761u;
        const unsigned z0 = c.g[2] * 805459861u, z1 = z0 + 805459861u;
        const unsigned a = y0 ^ z0, b = y1 ^ z0, cc = y0 ^ z1, d = y1 ^ z1;
        idx[0] = (x0 ^ a) & m; idx[1] = (x1 ^ a) & m; idx[2] = (x0 ^ b) & m; idx[3] = (x1 ^ b) & m;
        idx[4] = (x0 ^ cc) & m; idx[5] = (x1 ^ cc) & m; idx[6] = (x0 ^ d) & m; idx[7] = (x1 ^ d) & m;
#if defined(FIELD_DBG) && FIELD_DBG == 1
        for (int k = 0; k < 8; ++k) idx[k] &= 7;
#endif
        return;
    }
    const unsigned r = L.res;
    if (c.g[0] < r - 1u && c.g[1] < r - 1u && c.g[2] < r - 1u) {        // unsigned compare: negative coordinates (incl. -1) fail too
        const unsigned base = c.g[0] + (c.g[1] + c.g[2] * r) * r;          // < res^3 <= size: never wraps
        const unsigned rr = r * r;
        idx[0] = base; idx[1] = base + 1u; idx[2] = base + r; idx[3] = base + r + 1u;
        idx[4] = base + rr; idx[5] = base + rr + 1u; idx[6] = base + rr + r; idx[7] = base + rr + r + 1u;
#if defined(FIELD_DBG) && FIELD_DBG == 1
        for (int k = 0; k < 8; ++k) idx[k] &= 7;
#endif
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) idx[k] = corner_index(L, c, k);
}

// element idx of a table whose byte size is below 4 GiB (every grid here: <= 332 MB), addressed as uniform base + 32-bit byte
// offset: the load then takes its address as (SGPR base, VGPR offset) -- one shift per gather instead of a 64-bit shift-add
// and a zeroed high word (128 gathers per point in the fused kernels)
template <typename T>
__device__ __forceinline__ T at32(const T* __restrict__ base, unsigned idx) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(idx * (unsigned)sizeof(T)));
}

// F = 2 lookup (hash grid): returns the two interpolated features of one level.
__device__ __forceinline__ float2 lookup2(const float* __restrict__ table, const Level& L, const float x[3]) {
    const Cell c = locate(L, x);
    const float2* __restrict__ t = reinterpret_cast<const float2*>(table) + L.offset;
    unsigned idx[8];
    corner_indices(L, c, idx);
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = at32(t, idx[k]);
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = corner_weight(c, k);
        acc.x = fmaf(w, v[k].x, acc.x);
        acc.y = fmaf(w, v[k].y, acc.y);
    }
    return acc;
}

// F = 4 lookup (GBV) and F = 1 (GBW): single dense level.
__device__ __forceinline__ float4 lookup4(const float* __restrict__ table, const Level& L, const float x[3]) {
    const Cell c = locate(L, x);
    const float4* __restrict__ t = reinterpret_cast<const float4*>(table) + L.offset;
    unsigned idx[8];
    corner_indices(L, c, idx);
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = at32(t, idx[k]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = corner_weight(c, k);
        acc.x = fmaf(w, v[k].x, acc.x); acc.y = fmaf(w, v[k].y, acc.y);
        acc.z = fmaf(w, v[k].z, acc.z); acc.w = fmaf(w, v[k].w, acc.w);
    }
    return acc;
}

__device__ __forceinline__ float lookup1(const float* __restrict__ table, const Level& L, const float x[3]) {
    const Cell c = locate(L, x);
    const float* __restrict__ t = table + L.offset;
    unsigned idx[8];
    corner_indices(L, c, idx);
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = at32(t, idx[k]);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fmaf(corner_weight(c, k), v[k], acc);
    return acc;
}

// d(feature)/dx for one level: sum over the 4 edges along each dim (tcnn kernel_grid_backward_input
// semantics for Linear interpolation).  g = dL/dfeature (F values); returns dL/dx contribution.
template <int F>
__device__ __forceinline__ void lookup_dx(const float* __restrict__ table, const Level& L, const float x[3],
                                          const float* g, float dx[3]) {
    const Cell c = locate(L, x);
    const float* __restrict__ t = table + (size_t)L.offset * F;
    float dot[8];
    unsigned idx8[8];
    corner_indices(L, c, idx8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned idx = idx8[k];
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) s = fmaf(t[(size_t)idx * F + f], g[f], s);
        dot[k] = s;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float w = ((k >> d) & 1) ? 1.0f : -1.0f;
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (e != d) w *= ((k >> e) & 1) ? c.f[e] : (1.0f - c.f[e]);
            s = fmaf(w, dot[k], s);
        }
        dx[d] = fmaf(L.scale, s, dx[d]);
    }
}

// scatter dL/dfeature of one level into the gradient table (F = 2), one float atomic per value.
__device__ __forceinline__ void scatter2(float* __restrict__ dtable, const Level& L, const float x[3], float g0,
                                         float g1) {
    const Cell c = locate(L, x);
    float* __restrict__ t = dtable + (size_t)L.offset * 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned idx = corner_index(L, c, k);
        const float w = corner_weight(c, k);
        atomicAdd(t + (size_t)idx * 2, w * g0);
        atomicAdd(t + (size_t)idx * 2 + 1, w * g1);
    }
}

// ------------------------------------------------------------------------------ OneBlob
__device__ __forceinline__ float quartic_cdf(float t, float n) {
    const float u = t * n;
    const float u2 = u * u;
    const float u4 = u2 * u2;
    return fmaxf(0.0f, fminf(1.0f, (15.0f / 16.0f) * u * (1.0f - (2.0f / 3.0f) * u2 + (1.0f / 5.0f) * u4) + 0.5f));
}
// derivative of the (unclamped) cdf polynomial w.r.t. t: 15/16 n (1-u^2)^2 inside |u|<1, else 0
__device__ __forceinline__ float quartic_pdf(float t, float n) {
    const float u = t * n;
    const float q = fmaxf(1.0f - u * u, 0.0f);
    return (15.0f / 16.0f) * n * q * q;
}

__device__ __forceinline__ float round_fp16(float v) { return __half2float(__float2half_rn(v)); }

// out[k] for k = 0..NB-1 of one input dim (tcnn one_blob_subwarp_aligned):
//   out[k] = L(k+1) - L(k),  L(k) = cdf(t) + cdf(t-1) + cdf(t+1),  t = k/NB - x,  L(NB) := L(0) + 1.
// The kernel radius (1/NB) is far below the copy spacing (1), so at most one of the three copies is
// unsaturated and the other two are exactly 0 or 1:  L(k) = cdf(t - round(t)) + (1 + round(t)).
// One polynomial per boundary instead of three (differs from the literal sum by <= 1 ulp of L).
template <int NB>
__device__ __forceinline__ void oneblob_dim(float x, bool fp16, float* out) {
#if defined(FIELD_DBG) && FIELD_DBG == 2
    for (int k = 0; k < NB; ++k) out[k] = x * (float)k;   // timing experiment: no cdf evaluation
    return;
#endif
    const float n = (float)NB;
    float Lk[NB + 1];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float t = (float)k / n - x;   // k/NB exact for power-of-two NB
        const float r = (t > 0.5f) ? 1.0f : ((t < -0.5f) ? -1.0f : 0.0f);
        Lk[k] = quartic_cdf(t - r, n) + (1.0f + r);
    }
    Lk[NB] = Lk[0] + 1.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float v = Lk[k + 1] - Lk[k];
        out[k] = fp16 ? round_fp16(v) : v;
    }
}

// dL/dx through OneBlob for one dim given dL/dout[k] (straight-through across the fp16 rounding)
template <int NB>
__device__ __forceinline__ float oneblob_dim_dx(float x, const float* g) {
    const float n = (float)NB;
    float dL[NB + 1];   // dLk/dx = -(pdf(..)+pdf(..)+pdf(..))
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float lb = (float)k / n;
        dL[k] = -(quartic_pdf(lb - x, n) + quartic_pdf(lb - x - 1.0f, n) + quartic_pdf(lb - x + 1.0f, n));
    }
    dL[NB] = dL[0];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) s = fmaf(g[k], dL[k + 1] - dL[k], s);
    return s;
}

// ------------------------------------------------------------------------------ MFMA helpers
// v_permlane32_swap: swaps the upper 32 lanes of `a` with the lower 32 lanes of `b`.
// own-point values (a = feature 2s, b = feature 2s+1 of lane's point) -> (tile0 operand, tile1 operand)
// where a tile operand holds [k even of points 0..31 | k odd of points 0..31]; the same instruction
// turns a pair of D-layout registers (tile0, tile1) back into two own-point rows.
__device__ __forceinline__ void swap32(float& a, float& b) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#if defined(FIELD_DBG) && (FIELD_DBG == 5 || FIELD_DBG == 6)
    // timing experiment (results are wrong by construction): every fp32 k-step issued as ONE bf16 32x32x16 MFMA (half the
    // matrix-pipe passes, and on the pipe that co-executes with the VALU) -- an upper bound for what a split-bf16 MLP, which
    // needs 3-6 such products per 16 k but 8x fewer k-steps, could gain; FIELD_DBG == 6: no MFMA at all
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 av = {a, a, a, a}, bv = {b, b, b, b};
#if FIELD_DBG == 5
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
#else
    c[0] += a * b;
    return c;
#endif
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// row index held by (accumulator register r, lane half h) of a 32x32 D tile
__host__ __device__ __forceinline__ int krow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

}  // namespace rfx
