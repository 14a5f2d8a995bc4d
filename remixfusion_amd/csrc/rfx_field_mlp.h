// rfx_field_mlp.h -- the fused encode + MFMA-MLP device path shared by the field kernels
// (rfx_field.hip) and the fused ray renderer (rfx_render.hip).  See rfx_field.hip for the design.
#pragma once
#include "rfx_field_device.h"

namespace rfx {

// ---------------------------------------------------------------- staged weight layout (LDS)
#ifndef HASH_GROUP
#define HASH_GROUP 2          // hash levels per trip of the lookup loop (loads in flight = 8 * HASH_GROUP)
#endif
#define RFX_PRAGMA(x) _Pragma(#x)
#define RFX_UNROLL(n) RFX_PRAGMA(unroll n)
#ifndef FWD_WAVES
#define FWD_WAVES 2
#endif
#ifndef RENDER_WAVES
#define RENDER_WAVES FWD_WAVES      // the fused renderer (millions of points: throughput) may want more waves per SIMD than
#endif                              // the point-batch forward (1e5 points: latency, and no spills at 2 waves)
#ifndef BWD_WAVES
#define BWD_WAVES 2
#endif
constexpr int N_EMB = 32, N_POS = 48, N_IN1 = 81, N_H = 32, N_OUT2 = 16, N_IN3 = 66, N_OUT4 = 3;
constexpr int NPOS_SLOTS = 27;      // pos region of a layer: 48 columns x 36 floats (sparse fp32 image); the fp16 form uses 24
constexpr int POS_STRIDE = 36;      // floats per column of the sparse image (32 + 4 pad: conflict-free ds_read_b128)
constexpr int S1 = 16 + NPOS_SLOTS + 1, S2 = 16, S3 = NPOS_SLOTS + 10, S4 = 16;
constexpr int OFF1 = 0, OFF2 = OFF1 + S1, OFF3 = OFF2 + S2, OFF4 = OFF3 + S3, FWD_SLOTS = OFF4 + S4;   // 107
constexpr int OFFB4 = FWD_SLOTS, OFFB3 = OFFB4 + 2, OFFB2 = OFFB3 + 48, OFFB1 = OFFB2 + 9, ALL_SLOTS = OFFB1 + 48;  // 214

struct FieldK {            // by-value kernel argument
    rfx_grid_desc hash;
    const float* table;
    const float* gbv;
    Level gbv_level;
    const float *w1, *w2, *w3, *w4;
    float c_trunc, trunc, clamp_hi;
    int clamp_mode, pos_fp16;
    const float* staged;       // ALL_SLOTS*64 floats in operand order (rfx_field_stage_weights) or nullptr
};

// ---- OneBlob x weights on the fp16 matrix pipe (pos_fp16 = 1, tinycudann's default) -----------------
// The 48 OneBlob inputs are *exactly* fp16 numbers, so pos . W can run on v_mfma_f32_32x32x16_f16
// (16 k per instruction at 32 cycles instead of 2 k at 64 on the fp32 form) without rounding the
// activations.  The fp32 weight is split W = hi + lo with hi = fp16(W), lo = fp16((W - hi) * 2^11): the
// lo products are accumulated FIRST into the zeroed tile, the tile is scaled by 2^-11 (exact), and the hi
// products follow -- 22 significant bits of W, fp32 accumulation, no extra accumulator registers.
// The 24 fp32 slots of the pos part of a layer hold instead 6 groups (k-block 0..2) x (lo, hi) of one
// 16-byte A fragment per lane: lane (m = l & 31, h = l >> 5) owns W[m][16 kb + 8 h + 0..7].
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
constexpr float POS_LO_SCALE = 2048.0f;          // 2^11

__device__ __forceinline__ unsigned pack_half2(float a, float b) {     // round-to-nearest-even, like __float2half_rn
    half2v v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pack_half2(_Float16 a, _Float16 b) {
    half2v v = {a, b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float w16_part(float w, int hi_part) {
    const float hi = (float)(_Float16)w;
    return hi_part ? hi : (w - hi) * POS_LO_SCALE;
}
// dword i (0 .. 24*64) of the packed pos region of a layer whose pos columns start at w[m*ld + col0]
__device__ inline float packed_pos_weight(const float* __restrict__ w, int ld, int col0, int i) {
    const int g = i >> 8, rem = i & 255, l = rem >> 2, j2 = rem & 3;      // group = kb*2 + (0 lo | 1 hi)
    const int kb = g >> 1, part = g & 1, m = l & 31, h = l >> 5;
    const int p = 16 * kb + 8 * h + 2 * j2;
    const float* row = w + m * ld + col0 + p;
    return __uint_as_float(pack_half2(w16_part(row[0], part), w16_part(row[1], part)));
}

// ---- OneBlob x weights as a sparse fp32 dot product (pos_fp16 = 0: the reference's precision) -----------
// The reference builds OneBlob with dtype=torch.float (model/encodings.py:73): 48 fp32 inputs.  With 16 bins the
// kernel is one bin wide, so at most THREE bins per coordinate are non-zero: W[:, pos cols] . pos is a 9-term dot
// product per hidden unit, 288 fp32 FMAs per point and layer on the VALU, instead of 24 k-steps x 2 tiles of
// v_mfma_f32_32x32x2_f32 (3 072 matrix-pipe cycles per wave and layer) that multiply 39 zeros per point.
// The result initialises the layer's accumulator tiles; the remaining inputs follow on the matrix pipe.
// LDS image of one layer's pos columns (NPOS_SLOTS slots = 1 728 floats): column col occupies POS_STRIDE = 36 floats,
//   float 36 col + a (a < 32) = W[row(a)][col], row(a) = krow(a, 0) for a < 16, krow(a - 16, 1) for a >= 16
//   (so acc[0..15] / acc[16..31] become the two D-layout tiles with one v_permlane32_swap per register).
// A lane reads its column with eight ds_read_b128 at immediate offsets from ONE address register.  36 col mod 64
// takes sixteen different multiples of 4 over the sixteen columns of a coordinate, so the sixteen 4-bank windows
// of lanes that pick different columns never overlap: conflict-free (same column = same address = broadcast).
__device__ inline float sparse_pos_weight(const float* __restrict__ w, int ld, int col0, int i) {
    const int col = i / POS_STRIDE, a = i % POS_STRIDE;
    if (col >= N_POS || a >= 32) return 0.f;
    const int m = a < 16 ? krow(a, 0) : krow(a - 16, 1);
    return w[m * ld + col0 + col];
}

// value of staged slot `slot`, lane `l` (see header comment of each layer below)
__device__ inline float staged_weight(const FieldK& f, int slot, int l) {
    const int lo = l & 31, h = l >> 5;
    if (slot >= OFF1 + 16 && slot < OFF1 + 16 + NPOS_SLOTS) {
        const int i = (slot - (OFF1 + 16)) * 64 + l;
        return f.pos_fp16 ? (i < 24 * 64 ? packed_pos_weight(f.w1, N_IN1, N_EMB, i) : 0.f) : sparse_pos_weight(f.w1, N_IN1, N_EMB, i);
    }
    if (slot >= OFF3 && slot < OFF3 + NPOS_SLOTS) {
        const int i = (slot - OFF3) * 64 + l;
        return f.pos_fp16 ? (i < 24 * 64 ? packed_pos_weight(f.w3, N_IN3, 0, i) : 0.f) : sparse_pos_weight(f.w3, N_IN3, 0, i);
    }
    if (slot < OFF2) {                       // L1: A[hid][k=2s+h] = W1[hid][k]: 16 hash-level slots, [pos region], (cin, 0)
        const int s = slot - OFF1;
        const int k = s < 16 ? 2 * s + h : (h == 0 ? N_EMB + N_POS : -1);
        return k >= 0 ? f.w1[lo * N_IN1 + k] : 0.f;
    } else if (slot < OFF3) {                // L2: k-slot r = previous accumulator register r
        const int k = krow(slot - OFF2, h);
        return lo < N_OUT2 ? f.w2[lo * N_H + k] : 0.f;
    } else if (slot < OFF4) {                // L3: [pos pairs | h2 regs 0..7 (out o -> geo o-1) | ex_r,ex_g | ex_b,0]
        const int s = slot - OFF3;
        int k;
        if (s < NPOS_SLOTS) k = -1;            // pos region: handled above
        else if (s < NPOS_SLOTS + 8) { const int o = krow(s - NPOS_SLOTS, h); k = o >= 1 ? N_POS + o - 1 : -1; }
        else if (s == NPOS_SLOTS + 8) k = 63 + h;
        else k = h == 0 ? 65 : -1;
        return k >= 0 ? f.w3[lo * N_IN3 + k] : 0.f;
    } else if (slot < OFFB4) {               // L4 runs on the VALU (3 of 32 rows would be used on the MFMA):
        const int i = (slot - OFF4) * 64 + l;    //   flat [c][r][h] table: W4[c][krow(r,h)]
        if (i >= N_OUT4 * 16 * 2) return 0.f;
        const int c = i / 32, r = (i % 32) / 2, hh = i % 2;
        return f.w4[c * N_H + krow(r, hh)];
    } else if (slot < OFFB3) {               // B4: dH3[hid] = sum_o W4[o][hid] dY4[o]; k = o = 2s+h
        const int o = 2 * (slot - OFFB4) + h;
        return o < N_OUT4 ? f.w4[o * N_H + lo] : 0.f;
    } else if (slot < OFFB2) {               // B3: dX3[i] = sum_hid W3[hid][i] dH3[hid]; slot = mt*16 + r
        const int s = slot - OFFB3, mt = s >> 4, r = s & 15, i = 32 * mt + lo;
        return i < N_IN3 ? f.w3[krow(r, h) * N_IN3 + i] : 0.f;
    } else if (slot < OFFB1) {               // B2: dH1[hid] = sum_o W2[o][hid] dY2[o]
        const int s = slot - OFFB2;
        if (s < 8) {                         //   k-slot = dX3 M-tile-1 register 8+s: row q -> geo j=q-16 -> o=j+1
            const int j = krow(8 + s, h) - 16;
            return j <= 14 ? f.w2[(j + 1) * N_H + lo] : 0.f;
        }
        return h == 0 ? f.w2[0 * N_H + lo] : 0.f;   // (d_sdf, 0)
    } else {                                 // B1: dX1[i] = sum_hid W1[hid][i] dH1[hid]
        const int s = slot - OFFB1, mt = s >> 4, r = s & 15, i = 32 * mt + lo;
        return i < N_IN1 ? f.w1[krow(r, h) * N_IN1 + i] : 0.f;
    }
}

__device__ inline void stage_weights(const FieldK& f, float* wl, int n_slots) {
    if (f.staged) {          // pre-staged image: a straight 16-byte copy (L2-resident after the first block)
        const float4* __restrict__ src = reinterpret_cast<const float4*>(f.staged);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = threadIdx.x; i < n_slots * 16; i += blockDim.x) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < n_slots * 64; i += blockDim.x) wl[i] = staged_weight(f, i >> 6, i & 63);
    }
    __syncthreads();
}

// ---------------------------------------------------------------- per-point encodings (own-point layout)
struct Enc {
    unsigned pos16[N_POS / 2];   // POS16 path: fp16 pairs; after mlp_forward_123: tile-operand form
    float ex[4];      // GBV (tsdf in c_trunc units, r, g, b)
    float tres;       // tsdf rescaled/clamped: the residual added to the sdf output
    float cin;        // tsdf fed to the decoder
};

__device__ __forceinline__ void encode_point(const FieldK& f, const float x[3], Enc& e) {
    const float4 g = lookup4(f.gbv, f.gbv_level, x);
    e.ex[0] = g.x; e.ex[1] = g.y; e.ex[2] = g.z; e.ex[3] = g.w;
    float t = g.x * f.c_trunc;
    t = t / f.trunc;
    if (f.clamp_mode) {
        t = fminf(fmaxf(t, -f.clamp_hi), f.clamp_hi);
        e.cin = fminf(fmaxf(t, -1.0f), 1.0f);
    } else {
        t = fminf(fmaxf(t, -1.0f), 1.0f);
        e.cin = t;
    }
    e.tres = t;
    // OneBlob is evaluated inside mlp_forward_123
}

// ---------------------------------------------------------------- MLP forward on the matrix cores
struct Mlp {
    f32x16 h1[2], h2[2], h3[2];
};

// Hash-grid levels are looked up and fed to the matrix cores level by level (level l = k-step l),
// so the 32 features are never all live (EMB / POSST below: backward staging).
// consumes e.cin / e.ex; computes OneBlob itself (POS16: packed fp16 fragments in e.pos16, else the nine non-zero fp32 bins,
// multiplied on the VALU: sparse_pos_tiles)
// (historic note, dense fp32 form: e.pos in
// tile-operand form).  STAGE also writes pos (own-point fp32) to x1row[32..79].
// OneBlob (16 bins) of one coordinate as 16 fp16 values in 8 dwords (half k in dword k/2): the values of
// oneblob_dim<16>(x, fp16 = true) (to within 2^-24 absolute: a boundary a hair outside the kernel can round one
// ulp short of saturation in the full evaluation and leak one fp16 subnormal into a bin that is 0 here).
// The kernel is 1/16 wide, so only the boundaries L(kb-1..kb+2) around
// kb = floor(16 frac(x)) are unsaturated: four polynomial evaluations instead of sixteen, and the three
// non-zero outputs are dropped into place with a dword rotation.  Valid while |k/16 - x| < 1.5 for every k
// (the three periodic copies tinycudann sums cover that range); other x take the full evaluation.
__device__ __forceinline__ float oneblob_L16(int k, float x) {
    const float t = (float)k * 0.0625f - x;
    const float r = (t > 0.5f) ? 1.0f : ((t < -0.5f) ? -1.0f : 0.0f);
    return quartic_cdf(t - r, 16.0f) + (1.0f + r);
}

__device__ __forceinline__ void oneblob_dim_packed16(float x, unsigned dw[8]) {
#if defined(FIELD_DBG) && FIELD_DBG == 2
    for (int i = 0; i < 8; ++i) dw[i] = pack_half2(x * (float)i, x);      // timing experiment
    return;
#endif
    if (x >= -0.5f && x < 1.5f) {
        const float y = x - floorf(x);
        int kb = (int)floorf(y * 16.0f);
        kb = min(max(kb, 0), 15);
        const int c0 = (kb + 15) & 15, c1 = kb, c2 = (kb + 1) & 15, c3 = (kb + 2) & 15;
        const float L0 = oneblob_L16(c0, x), L1 = oneblob_L16(c1, x), L2 = oneblob_L16(c2, x), L3 = oneblob_L16(c3, x);
        const _Float16 h0 = (_Float16)(((c1 == 0) ? L1 + 1.0f : L1) - L0);      // out[c0] = L(c0 + 1) - L(c0), L(16) = L(0) + 1
        const _Float16 h1 = (_Float16)(((c2 == 0) ? L2 + 1.0f : L2) - L1);      // out[c1]
        const _Float16 h2 = (_Float16)(((c3 == 0) ? L3 + 1.0f : L3) - L2);      // out[c2]
        const _Float16 z = (_Float16)0.0f;
        const bool odd = (c0 & 1) != 0;
        const unsigned A = odd ? pack_half2(z, h0) : pack_half2(h0, h1);
        const unsigned B = odd ? pack_half2(h1, h2) : pack_half2(h2, z);
        const int q = c0 >> 1, q1 = (q + 1) & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) dw[i] = (i == q) ? A : ((i == q1) ? B : 0u);
    } else {
        float v[16];
        oneblob_dim<16>(x, false, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) dw[i] = pack_half2(v[2 * i], v[2 * i + 1]);
    }
}

__device__ __forceinline__ half8 frag16(unsigned a, unsigned b, unsigned c, unsigned d) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(half8, v);
}
__device__ __forceinline__ half8 lds_frag16(const float* __restrict__ wl, int slot0, int group, int lane) {
    return *reinterpret_cast<const half8*>(wl + slot0 * 64 + group * 256 + lane * 4);
}
// acc(tile0, tile1) += W_pos . pos for the three k-blocks of one part (0 = lo, 1 = hi)
__device__ __forceinline__ void pos_mfma16(const float* __restrict__ wl, int slot0, int part, int lane, const unsigned* p16,
                                           f32x16& t0, f32x16& t1) {
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
        const half8 a = lds_frag16(wl, slot0, kb * 2 + part, lane);
        const half8 b0 = frag16(p16[8 * kb], p16[8 * kb + 2], p16[8 * kb + 4], p16[8 * kb + 6]);
        const half8 b1 = frag16(p16[8 * kb + 1], p16[8 * kb + 3], p16[8 * kb + 5], p16[8 * kb + 7]);
        t0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, t1, 0, 0, 0);
    }
}
__device__ __forceinline__ void scale16(f32x16& t, float s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] *= s;
}

// OneBlob of one coordinate in sparse fp32 form.  tinycudann's formula (oracle/field_oracle.py::oneblob_encode):
//   out[k] = L(k+1) - L(k),  L(k) = cdf(k/16 - x) + cdf(k/16 - x - 1) + cdf(k/16 - x + 1),  L(16) := L(0) + 1,
// evaluated per boundary by oneblob_L16 (valid for every x).  The three kernel copies share frac(x), so the only
// boundaries that are not saturated are those next to kb = floor(16 frac(x)): bins c0 = kb-1, c1 = kb, c2 = kb+1
// (mod 16) carry the values, evaluated literally (wrap rule included).  The bins sum to L(16) - L(0) = 1 by
// construction and are all >= 0, so whatever the three do not hold sits in bin 15 through the wrap rule: for x inside
// about (-0.94, 1.94) that remainder is rounding noise (|.| < 1e-6, dropped below 4e-6 so that a point's result never
// depends on its wave neighbours), for points far outside the bound it is the whole unit mass (tinycudann's e_15).
// It is carried as a fourth (column, value) pair per coordinate that a wave multiplies only if some lane needs it.
struct PosBins {
    float v[9];        // [3 d + j]: values of the three bins around x[d]
    int c[9];          // their columns 16 d + bin
    float rest[3];     // value of column 16 d + 15 not covered by v (0 in range)
};
__device__ __forceinline__ void oneblob_dim_sparse(float x, int d, PosBins& pb) {
    const float y = x - floorf(x);
    int kb = (int)floorf(y * 16.0f);
    kb = min(max(kb, 0), 15);
    const int c0 = (kb + 15) & 15, c1 = kb, c2 = (kb + 1) & 15, c3 = (kb + 2) & 15;
    const float L0 = oneblob_L16(c0, x), L1 = oneblob_L16(c1, x), L2 = oneblob_L16(c2, x), L3 = oneblob_L16(c3, x);
    const float v0 = ((c1 == 0) ? L1 + 1.0f : L1) - L0;      // out[c0] = L(c0 + 1) - L(c0), L(16) = L(0) + 1
    const float v1 = ((c2 == 0) ? L2 + 1.0f : L2) - L1;
    const float v2 = ((c3 == 0) ? L3 + 1.0f : L3) - L2;
    pb.v[3 * d + 0] = v0; pb.v[3 * d + 1] = v1; pb.v[3 * d + 2] = v2;
    pb.c[3 * d + 0] = 16 * d + c0; pb.c[3 * d + 1] = 16 * d + c1; pb.c[3 * d + 2] = 16 * d + c2;
    const float rest = 1.0f - ((v0 + v1) + v2);
    const bool has15 = c0 == 15 || c1 == 15 || c2 == 15;
    pb.rest[d] = (!has15 && fabsf(rest) > 4e-6f) ? rest : 0.0f;
}

// the dense 16 bins of coordinate d (backward staging, standalone encoder): zeros except the pairs of pb
__device__ __forceinline__ void oneblob_dense_from_sparse(const PosBins& pb, int d, float full[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int col = 16 * d + k;
        float v = col == pb.c[3 * d] ? pb.v[3 * d] : (col == pb.c[3 * d + 1] ? pb.v[3 * d + 1]
                 : (col == pb.c[3 * d + 2] ? pb.v[3 * d + 2] : 0.f));
        if (k == 15) v += pb.rest[d];
        full[k] = v;
    }
}

// acc[a] += v * W[row(a)][col] for the 32 accumulator positions; wp = the layer's sparse image
__device__ __forceinline__ void sparse_col_fma(const float* __restrict__ wp, int col, float v, float acc[32]) {
    const float* __restrict__ base = wp + col * POS_STRIDE;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(base + 4 * q);
        acc[4 * q] = fmaf(v, w.x, acc[4 * q]); acc[4 * q + 1] = fmaf(v, w.y, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(v, w.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(v, w.w, acc[4 * q + 3]);
    }
}

// (t0, t1) += W[:, pos cols] . OneBlob(x) as D-layout tiles.  `extra` (wave-uniform): some lane has a non-zero
// pb.rest (a point far outside the scene bound): three more columns, wave-uniform addresses (LDS broadcast).
__device__ __forceinline__ void sparse_pos_tiles(const float* __restrict__ wp, const PosBins& pb, bool extra,
                                                 f32x16& t0, f32x16& t1) {
    float acc[32];
#pragma unroll
    for (int a = 0; a < 32; ++a) acc[a] = 0.f;
    // one column (8 x ds_read_b128 + 32 FMAs) at a time: left alone, the scheduler hoists all 72 reads of the nine
    // columns to the top (288 live registers, spills)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        sparse_col_fma(wp, pb.c[j], pb.v[j], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (extra) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sparse_col_fma(wp, 16 * d + 15, pb.rest[d], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = acc[r], b = acc[16 + r];
        swap32(a, b);           // a: [own half-0 rows | lower partner's half-1 rows] = tile 0;  b = tile 1
        t0[r] += a; t1[r] += b;
    }
}

// Backward staging: emb_row / pos_row point at the lane's own point inside a piece-major tile of 64 points (rfx_field.hip,
// workspace layout): float4 piece q of the row is ROW_PIECE floats after piece q - 1, so a wave's store (or load) of one
// piece is contiguous.
//   EMB 0: look the hash features up.  1: look them up and store them to emb_row (8 pieces).  2: load them from emb_row,
//          where an earlier kernel (the forward of the same iteration) left them: no gathers, no interpolation; emb_copy
//          (optional) receives them again, at the row the caller stages this point at.
//   POSST: store the dense fp32 OneBlob row to pos_row (12 pieces).
constexpr int ROW_PIECE = 256;
__device__ __forceinline__ float* row_piece(float* row, int col) { return row + (col >> 2) * ROW_PIECE + (col & 3); }

template <int EMB, bool POSST, bool POS16>
__device__ __forceinline__ void mlp_forward_123(const FieldK& f, const float x[3], const float* __restrict__ wl,
                                                int lane, Enc& e, Mlp& m, float* emb_row = nullptr, float* pos_row = nullptr,
                                                bool valid = true, float* emb_copy = nullptr) {
    m.h1[0] = zero16(); m.h1[1] = zero16();
    if (POS16) {
        // OneBlob first: its lo-weight products must enter the empty accumulators (see above)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            unsigned dw[8];
            oneblob_dim_packed16(x[d], dw);
            if (POSST && valid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const half2v p0 = __builtin_bit_cast(half2v, dw[2 * i]), p1 = __builtin_bit_cast(half2v, dw[2 * i + 1]);
                    *reinterpret_cast<float4*>(row_piece(pos_row, 16 * d + 4 * i)) = make_float4((float)p0[0], (float)p0[1], (float)p1[0], (float)p1[1]);
                }
            }
            // dwords 0..3 = k 0..7 (lane half 0's fragment), 4..7 = k 8..15; (a_i, b_i) pairs for the swap
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = __uint_as_float(dw[i]);
                float b = __uint_as_float(dw[4 + i]);
                swap32(a, b);                // a: tile-0 fragment dword i, b: tile-1 fragment dword i
                e.pos16[8 * d + 2 * i] = __float_as_uint(a);
                e.pos16[8 * d + 2 * i + 1] = __float_as_uint(b);
            }
        }
        pos_mfma16(wl, OFF1 + 16, 0, lane, e.pos16, m.h1[0], m.h1[1]);
        scale16(m.h1[0], 1.0f / POS_LO_SCALE); scale16(m.h1[1], 1.0f / POS_LO_SCALE);
        pos_mfma16(wl, OFF1 + 16, 1, lane, e.pos16, m.h1[0], m.h1[1]);
    }
    // HASH_GROUP levels per trip, in three phases: (1) cells and corner indices of all the group's levels, (2) their
    // 8 x HASH_GROUP gathers issued back to back, (3) interpolation and the matrix steps.  Written as one lookup2() per level
    // the loop made one memory round trip per LEVEL (index arithmetic with its level-uniform branches, eight loads, wait
    // for all eight, two MFMAs; sixteen dependent trips per batch): the compiler does not move loads across the branches
    // of the next level's index arithmetic.  A real loop over the groups: level constants are fetched per trip instead of
    // keeping all 16 x 5 of them live in SGPRs.
    constexpr int HG = HASH_GROUP;
    static_assert(16 % HG == 0, "HASH_GROUP must divide the 16 levels");
    if (EMB == 2) {
        float4 q[8];                  // levels 2i, 2i+1 are piece i of the stashed row: eight contiguous loads, issued together
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = *reinterpret_cast<const float4*>(emb_row + i * ROW_PIECE);
        if (emb_copy && valid) {      // the staged copy the weight-gradient kernel reads (the chain's own point order)
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(emb_copy + i * ROW_PIECE) = q[i];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float a = (s & 1) ? q[s >> 1].z : q[s >> 1].x, b = (s & 1) ? q[s >> 1].w : q[s >> 1].y;
            swap32(a, b);
            const float w = wl[(OFF1 + s) * 64 + lane];
            m.h1[0] = mfma32(w, a, m.h1[0]);
            m.h1[1] = mfma32(w, b, m.h1[1]);
        }
    }
#pragma unroll 1
    for (int s0 = 0; s0 < (EMB == 2 ? 0 : 16); s0 += HG) {
        float2 v[HG];
#if defined(FIELD_DBG) && FIELD_DBG == 4
#pragma unroll
        for (int g = 0; g < HG; ++g) v[g] = make_float2(x[0] * (float)(s0 + g), x[1]);      // timing experiment: no hash lookups
#else
        Cell cell[HG];
        unsigned idx[HG][8];
        const float2* tl[HG];
#pragma unroll
        for (int g = 0; g < HG; ++g) {
            const Level L = get_level(f.hash, s0 + g);
            cell[g] = locate(L, x);
            corner_indices(L, cell[g], idx[g]);
            tl[g] = reinterpret_cast<const float2*>(f.table) + L.offset;
        }
        float2 cv[HG][8];
#pragma unroll
        for (int g = 0; g < HG; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) cv[g][k] = at32(tl[g], idx[g][k]);
#pragma unroll
        for (int g = 0; g < HG; ++g) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = corner_weight(cell[g], k);
                acc.x = fmaf(w, cv[g][k].x, acc.x);
                acc.y = fmaf(w, cv[g][k].y, acc.y);
            }
            v[g] = acc;
        }
#endif
#pragma unroll
        for (int g = 0; g < HG; ++g) {
            const int s = s0 + g;
            if (EMB == 1 && valid) {
                if (HG % 2 == 0) {      // levels 2i, 2i+1 are one piece
                    if (g % 2 == 0) *reinterpret_cast<float4*>(row_piece(emb_row, 2 * s)) = make_float4(v[g].x, v[g].y, v[(g + 1) % HG].x, v[(g + 1) % HG].y);
                } else {
                    *reinterpret_cast<float2*>(row_piece(emb_row, 2 * s)) = v[g];
                }
            }
            float a = v[g].x, b = v[g].y;
            swap32(a, b);
            const float w = wl[(OFF1 + s) * 64 + lane];
            m.h1[0] = mfma32(w, a, m.h1[0]);
            m.h1[1] = mfma32(w, b, m.h1[1]);
        }
    }
    PosBins pb;
    bool extra = false;
    if (!POS16) {
#pragma unroll
        for (int d = 0; d < 3; ++d) oneblob_dim_sparse(x[d], d, pb);
        extra = __any(pb.rest[0] != 0.0f || pb.rest[1] != 0.0f || pb.rest[2] != 0.0f) != 0;
        if (POSST && valid) {       // dense fp32 row for the weight-gradient kernel
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float full[16];
                oneblob_dense_from_sparse(pb, d, full);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(row_piece(pos_row, 16 * d + 4 * i)) = make_float4(full[4 * i], full[4 * i + 1], full[4 * i + 2], full[4 * i + 3]);
            }
        }
        sparse_pos_tiles(wl + (OFF1 + 16) * 64, pb, extra, m.h1[0], m.h1[1]);
    }
    {
        float a = e.cin, b = 0.f;
        swap32(a, b);
        const float w = wl[(OFF1 + 16 + NPOS_SLOTS) * 64 + lane];
        m.h1[0] = mfma32(w, a, m.h1[0]);
        m.h1[1] = mfma32(w, b, m.h1[1]);
    }
    m.h2[0] = zero16(); m.h2[1] = zero16();
#if defined(FIELD_DBG) && FIELD_DBG == 3
    m.h2[0] = m.h1[0]; m.h2[1] = m.h1[1]; m.h3[0] = m.h1[1]; m.h3[1] = m.h1[0];   // timing experiment: no layer 2/3 MFMAs
    return;
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float w = wl[(OFF2 + r) * 64 + lane];
        m.h2[0] = mfma32(w, fmaxf(m.h1[0][r], 0.f), m.h2[0]);
        m.h2[1] = mfma32(w, fmaxf(m.h1[1][r], 0.f), m.h2[1]);
    }
    m.h3[0] = zero16(); m.h3[1] = zero16();
    if (POS16) {
        pos_mfma16(wl, OFF3, 0, lane, e.pos16, m.h3[0], m.h3[1]);
        scale16(m.h3[0], 1.0f / POS_LO_SCALE); scale16(m.h3[1], 1.0f / POS_LO_SCALE);
        pos_mfma16(wl, OFF3, 1, lane, e.pos16, m.h3[0], m.h3[1]);
    } else {
        sparse_pos_tiles(wl + OFF3 * 64, pb, extra, m.h3[0], m.h3[1]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float w = wl[(OFF3 + NPOS_SLOTS + r) * 64 + lane];
        m.h3[0] = mfma32(w, m.h2[0][r], m.h3[0]);
        m.h3[1] = mfma32(w, m.h2[1][r], m.h3[1]);
    }
    {
        float a = e.ex[1], b = e.ex[2];
        swap32(a, b);
        float w = wl[(OFF3 + NPOS_SLOTS + 8) * 64 + lane];
        m.h3[0] = mfma32(w, a, m.h3[0]);
        m.h3[1] = mfma32(w, b, m.h3[1]);
        a = e.ex[3]; b = 0.f;
        swap32(a, b);
        w = wl[(OFF3 + NPOS_SLOTS + 9) * 64 + lane];
        m.h3[0] = mfma32(w, a, m.h3[0]);
        m.h3[1] = mfma32(w, b, m.h3[1]);
    }
}

// layer 4 + residual add -> raw4 of the lane's own point.  rgb = W4 relu(h3): each lane holds 16 of the
// 32 hidden units of its tile's point (D layout), so it forms a partial dot product; one
// v_permlane32_swap per channel brings the two halves of every point together.
__device__ __forceinline__ void mlp_forward_4(const float* __restrict__ wl, int lane, const Enc& e, const Mlp& m,
                                              float raw[4]) {
    const float* __restrict__ w4 = wl + OFF4 * 64;
    const int h = lane >> 5;
    float p0[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float a0 = fmaxf(m.h3[0][r], 0.f), a1 = fmaxf(m.h3[1][r], 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float w = w4[(c * 16 + r) * 2 + h];
            p0[c] = fmaf(a0, w, p0[c]);
            p1[c] = fmaf(a1, w, p1[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = p0[c], b = p1[c];
        swap32(a, b);                 // a = lower-half partials of the own point, b = upper-half partials
        raw[c] = (a + b) + e.ex[c + 1];
    }
    float a = m.h2[0][0], b = m.h2[1][0];
    swap32(a, b);
    raw[3] = a + e.tres;
}

// ---------------------------------------------------------------- the HALF pass: 32 points per wave (round 6)
// A launch of P 64-point passes on S wave slots costs ceil(P / S) pass-times on the SIMDs that take an extra wave-pass
// (profiles/r5_notes.md: 2 124 passes on 2 048 slots: 55.7 us against 43.9 us for 1 888).  The passes beyond the last full round
// are therefore dealt out as HALF passes, one per SIMD: 32 points, for which BOTH halves of the wave work -- lane l and lane
// l + 32 serve point l.  The lower half looks up hash levels 0-7, the upper half levels 8-15, in the same instructions
// (level constants selected per lane); after v_permlane32_swap the pair (a, b) is the tile-0 operand of level s and of level
// s + 8: two k-steps into ONE accumulator tile.  The OneBlob columns are split 5 + 4 the same way and the two partial sums
// meet in the swap that turns the 32 own-point sums into the D layout.  Half the gathers, half the column products and half the
// MFMAs of a pass per lane; the GBV lookup is done by both halves.  fp32 OneBlob only (the reference's precision); results are
// the full pass's up to the order of the fp32 additions of one point (levels interleave s, s + 8 instead of ascending).
struct MlpH { f32x16 h1, h2, h3; };

__device__ __forceinline__ Level select_level(const Level& a, const Level& b, bool hi) {
    Level L;
    L.scale = hi ? b.scale : a.scale; L.res = hi ? b.res : a.res; L.size = hi ? b.size : a.size;
    L.offset = hi ? b.offset : a.offset; L.hashed = hi ? b.hashed : a.hashed;
    return L;
}

// (t) += W[:, pos cols] . OneBlob(x): lane half `hi` multiplies columns j = 5 hi .. 5 hi + 4 (the upper half's fifth is empty);
// the two halves' sums for the same point are added by the swap
__device__ __forceinline__ void sparse_pos_tile_half(const float* __restrict__ wp, const PosBins& pb, bool extra, bool hi, f32x16& t) {
    float acc[32];
#pragma unroll
    for (int a = 0; a < 32; ++a) acc[a] = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int col = hi ? (i < 4 ? pb.c[5 + i] : 0) : pb.c[i];
        const float v = hi ? (i < 4 ? pb.v[5 + i] : 0.f) : pb.v[i];
        sparse_col_fma(wp, col, v, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (extra) {                      // wave-uniform: some lane's point lies far outside the bound (see sparse_pos_tiles)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sparse_col_fma(wp, 16 * d + 15, hi ? 0.f : pb.rest[d], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = acc[r], b = acc[16 + r];
        swap32(a, b);           // lower lane: a = own rows half 0, b = the partner's rows half 0; upper lane: a = the partner's rows half 1, b = own
        t[r] += a + b;
    }
}

template <int EMB>
__device__ __forceinline__ void mlp_forward_123_half(const FieldK& f, const float x[3], const float* __restrict__ wl, int lane, Enc& e,
                                                     MlpH& m, float* emb_row, bool valid) {
    const bool hi = lane >= 32;
    m.h1 = zero16();
    constexpr int HG = HASH_GROUP;
    static_assert(8 % HG == 0, "HASH_GROUP must divide the 8 level pairs of a half pass");
    const float2* __restrict__ t2 = reinterpret_cast<const float2*>(f.table);
#pragma unroll 1
    for (int s0 = 0; s0 < 8; s0 += HG) {
        Cell cell[HG];
        unsigned idx[HG][8];
#pragma unroll
        for (int g = 0; g < HG; ++g) {
            const Level L = select_level(get_level(f.hash, s0 + g), get_level(f.hash, s0 + g + 8), hi);
            cell[g] = locate(L, x);
            corner_indices(L, cell[g], idx[g]);
#pragma unroll
            for (int k = 0; k < 8; ++k) idx[g][k] += L.offset;          // (the table is below 4 GiB: a 32-bit element index, at32)
        }
        float2 cv[HG][8];
#pragma unroll
        for (int g = 0; g < HG; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) cv[g][k] = at32(t2, idx[g][k]);
        float2 v[HG];
#pragma unroll
        for (int g = 0; g < HG; ++g) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = corner_weight(cell[g], k);
                acc.x = fmaf(w, cv[g][k].x, acc.x);
                acc.y = fmaf(w, cv[g][k].y, acc.y);
            }
            v[g] = acc;
        }
#pragma unroll
        for (int g = 0; g < HG; ++g) {
            const int s = s0 + g;
            if (EMB == 1 && valid) {          // the stash row of the point: this lane's level (s or s + 8)
                float* dst = row_piece(emb_row, 2 * (s + (hi ? 8 : 0)));
                if (HG % 2 == 0) {
                    if (g % 2 == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[g].x, v[g].y, v[(g + 1) % HG].x, v[(g + 1) % HG].y);
                } else {
                    *reinterpret_cast<float2*>(dst) = v[g];
                }
            }
            float a = v[g].x, b = v[g].y;
            swap32(a, b);                     // a: (x, y) of level s for point l; b: (x, y) of level s + 8 for point l
            m.h1 = mfma32(wl[(OFF1 + s) * 64 + lane], a, m.h1);
            m.h1 = mfma32(wl[(OFF1 + s + 8) * 64 + lane], b, m.h1);
        }
    }
    PosBins pb;
#pragma unroll
    for (int d = 0; d < 3; ++d) oneblob_dim_sparse(x[d], d, pb);
    const bool extra = __any(pb.rest[0] != 0.0f || pb.rest[1] != 0.0f || pb.rest[2] != 0.0f) != 0;
    sparse_pos_tile_half(wl + (OFF1 + 16) * 64, pb, extra, hi, m.h1);
    {
        float a = e.cin, b = 0.f;
        swap32(a, b);
        m.h1 = mfma32(wl[(OFF1 + 16 + NPOS_SLOTS) * 64 + lane], a, m.h1);
    }
    m.h2 = zero16();
#pragma unroll
    for (int r = 0; r < 16; ++r) m.h2 = mfma32(wl[(OFF2 + r) * 64 + lane], fmaxf(m.h1[r], 0.f), m.h2);
    m.h3 = zero16();
    sparse_pos_tile_half(wl + OFF3 * 64, pb, extra, hi, m.h3);
#pragma unroll
    for (int r = 0; r < 8; ++r) m.h3 = mfma32(wl[(OFF3 + NPOS_SLOTS + r) * 64 + lane], m.h2[r], m.h3);
    {
        float a = e.ex[1], b = e.ex[2];
        swap32(a, b);
        m.h3 = mfma32(wl[(OFF3 + NPOS_SLOTS + 8) * 64 + lane], a, m.h3);
        a = e.ex[3]; b = 0.f;
        swap32(a, b);
        m.h3 = mfma32(wl[(OFF3 + NPOS_SLOTS + 9) * 64 + lane], a, m.h3);
    }
}

// layer 4 + residual add of a half pass: valid in the LOWER lanes (point = lane)
__device__ __forceinline__ void mlp_forward_4_half(const float* __restrict__ wl, int lane, const Enc& e, const MlpH& m, float raw[4]) {
    const float* __restrict__ w4 = wl + OFF4 * 64;
    const int h = lane >> 5;
    float p0[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float a0 = fmaxf(m.h3[r], 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) p0[c] = fmaf(a0, w4[(c * 16 + r) * 2 + h], p0[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = p0[c], b = p0[c];
        swap32(a, b);                 // lower lane: a = own partial (rows half 0), b = the partner's (rows half 1)
        raw[c] = (a + b) + e.ex[c + 1];
    }
    float a = m.h2[0], b = m.h2[0];
    swap32(a, b);
    raw[3] = a + e.tres;
}

__device__ __forceinline__ void load_point(const float* __restrict__ x01, int64_t p, int64_t n, float x[3]) {
    if (p < n) { x[0] = x01[p * 3]; x[1] = x01[p * 3 + 1]; x[2] = x01[p * 3 + 2]; }
    else { x[0] = x[1] = x[2] = 0.5f; }
}


// host: build the by-value kernel argument from the public descriptor (defined in rfx_field.hip)
int make_fieldk(const rfx_field_desc* d, FieldK* k);

}  // namespace rfx
