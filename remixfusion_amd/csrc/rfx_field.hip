// rfx_field.hip -- residual neural field on gfx950: multires hash grid + OneBlob + dense GBV lookup
// + the two bias-free MLPs (81->32->16, 66->32->3), forward and backward.
//
// Replaces tinycudann.Encoding (HashGrid / OneBlob / dense Grid) and the torch nn.Linear MLP of the
// reference for JointEncoding.query_color_sdf & friends (model/scene_rep.py:212-349,
// model/encodings.py:33-76, model/decoder.py:116-146).
//
// Design (MI355X):
//  * one lane = one sample point, one wave = 64 points.  Encodings are computed per lane in
//    registers (gathers from the L2/Infinity-Cache resident tables), never written to HBM.
//  * the MLP runs on the matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32) in the *transposed*
//    form  H^T[out x pts] = W[out x in] . X^T[in x pts] : weights are the A operand (staged once
//    per block in LDS, already in operand order), points are the B operand.  A lane's features
//    become B operands with one v_permlane32_swap per feature pair, and the 32x32 accumulator of
//    one layer *is* the B operand of the next layer (its K order is folded into the staged
//    weights), so activations never leave registers.
//  * backward = recompute forward, chain the same trick for dX, stage the per-point rows needed
//    for the weight gradients in a workspace, then (a) a streaming MFMA kernel reduces
//    dW = dY^T X over points with deterministic two-stage partial sums and (b) a per-point kernel
//    scatters the hash-grid gradients (float atomics) and evaluates d/dx of the encodings.
#include "rfx_field_mlp.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

namespace rfx {

// ---------------------------------------------------------------- Q1 forward kernel
// STASH: also leave the interpolated hash features in `emb` (piece-major tiles, see the backward workspace layout) for the
// backward chain of the same iteration, which then needs no hash lookups of its own.
// EMB (mlp_forward_123): 0 look the hash features up, 1 look them up and stash them, 2 read them from the stash (the table is not
// touched: a level-partitioned table's features arrive from the ranks that own the levels, rfx_field_stash_put).
// n_full: the points taken in 64-point passes (all of them unless the host deals the tail out in halves); n_half: number of
// 32-point half passes behind them, one per wave with index < n_half (mlp_forward_123_half in rfx_field_mlp.h)
template <bool POS16, int EMB>
__global__ __launch_bounds__(256, FWD_WAVES) void field_forward_kernel(FieldK f, const float* __restrict__ x01, int64_t n,
                                                            float* __restrict__ raw4, float* __restrict__ emb, int64_t n_full,
                                                            int n_half) {
    __shared__ __attribute__((aligned(16))) float wl[FWD_SLOTS * 64];
    stage_weights(f, wl, FWD_SLOTS);
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    if (!POS16 && EMB != 2 && wave < n_half) {          // wave-uniform
        const int64_t p = n_full + wave * 32 + (lane & 31);
        float x[3];
        load_point(x01, p, n, x);
        Enc e;
        encode_point(f, x, e);
        MlpH m;
        mlp_forward_123_half<EMB>(f, x, wl, lane, e, m, EMB ? emb + (p >> 6) * (8 * ROW_PIECE) + (p & 63) * 4 : nullptr, p < n);
        float raw[4];
        mlp_forward_4_half(wl, lane, e, m, raw);
        if (p < n && lane < 32) reinterpret_cast<float4*>(raw4)[p] = make_float4(raw[0], raw[1], raw[2], raw[3]);
    }
    for (int64_t base = wave * 64; base < n_full; base += n_waves * 64) {
        const int64_t p = base + lane;
        float x[3];
        load_point(x01, p, n, x);
        Enc e;
        encode_point(f, x, e);
        Mlp m;
        mlp_forward_123<EMB, false, POS16>(f, x, wl, lane, e, m, EMB ? emb + (p >> 6) * (8 * ROW_PIECE) + (p & 63) * 4 : nullptr,
                                           nullptr, p < n);
        float raw[4];
        mlp_forward_4(wl, lane, e, m, raw);
        if (p < n) reinterpret_cast<float4*>(raw4)[p] = make_float4(raw[0], raw[1], raw[2], raw[3]);
    }
}

// Q2: query_sdf_res (always +-1 clamp) / query_color_residual (decoder fed the raw GBV tsdf)
template <int MODE, bool POS16>   // MODE 0: sdf, 1: colour
__global__ __launch_bounds__(256, FWD_WAVES) void field_query_kernel(FieldK f, const float* __restrict__ x01, int64_t n,
                                                          float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float wl[FWD_SLOTS * 64];
    stage_weights(f, wl, FWD_SLOTS);
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t base = wave * 64; base < n; base += n_waves * 64) {
        const int64_t p = base + lane;
        float x[3];
        load_point(x01, p, n, x);
        Enc e;
        encode_point(f, x, e);
        if (MODE == 1) e.cin = e.ex[0];          // scene_rep.py:294: ex_Trgb[...,:1] unscaled
        Mlp m;
        mlp_forward_123<0, false, POS16>(f, x, wl, lane, e, m);
        if (MODE == 0) {
            float a = m.h2[0][0], b = m.h2[1][0];
            swap32(a, b);
            if (p < n) out[p] = a + e.tres;
        } else {
            float raw[4];
            mlp_forward_4(wl, lane, e, m, raw);
            if (p < n) { out[p * 3] = raw[0]; out[p * 3 + 1] = raw[1]; out[p * 3 + 2] = raw[2]; }
        }
    }
}

// ---------------------------------------------------------------- E1 / E2 standalone
__global__ __launch_bounds__(256) void grid_encode_forward_kernel(rfx_grid_desc g, const float* __restrict__ table,
                                                                  const float* __restrict__ x01, int64_t n,
                                                                  float* __restrict__ feat) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float x[3] = {x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]};
    const int Fd = g.n_feat, L = g.n_levels;
    float* o = feat + p * (int64_t)(L * Fd);
    for (int l = 0; l < L; ++l) {
        const Level lv = get_level(g, l);
        if (Fd == 2) {
            const float2 v = lookup2(table, lv, x);
            reinterpret_cast<float2*>(o)[l] = v;
        } else if (Fd == 4) {
            const float4 v = lookup4(table, lv, x);
            reinterpret_cast<float4*>(o)[l] = v;
        } else {
            o[l] = lookup1(table, lv, x);
        }
    }
}

// Level-parallel forms for F = 2: thread = (point, level), the LP lanes of a point next to each other (LP = levels
// rounded up to a power of two).  A thread that walks all levels of its point strings 16 dependent gather round trips
// together (26 us for the 3e4-point TV lattice, whatever the point count); here each thread makes one, and the level
// loop becomes parallelism across waves.  A point's features are written as one contiguous 8 * L byte run.
__global__ __launch_bounds__(256) void grid_encode_forward_lp_kernel(rfx_grid_desc g, const float* __restrict__ table,
                                                                     const float* __restrict__ x01, int64_t n, int lp_shift,
                                                                     float* __restrict__ feat) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = gid >> lp_shift;
    const int l = (int)(gid & ((1 << lp_shift) - 1));
    if (p >= n || l >= g.n_levels) return;
    const float x[3] = {x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]};
    reinterpret_cast<float2*>(feat + p * (int64_t)(g.n_levels * 2))[l] = lookup2(table, get_level(g, l), x);
}

// dL/dx01 through the grid (no table gradient): per-level contributions summed over the LP lanes of the point
__global__ __launch_bounds__(256) void grid_encode_dx_lp_kernel(rfx_grid_desc g, const float* __restrict__ table,
                                                                const float* __restrict__ x01, int64_t n,
                                                                const float* __restrict__ dfeat, int ld, int lp_shift,
                                                                float* __restrict__ dx01, const int* __restrict__ perm,
                                                                const int* __restrict__ n_sel) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = gid >> lp_shift;
    const int l = (int)(gid & ((1 << lp_shift) - 1));
    const int64_t px = (perm && p < n) ? perm[p] : p;          // dfeat row p belongs to point px
    float dx[3] = {0.f, 0.f, 0.f};
    if (p < n && (!n_sel || p < *n_sel) && l < g.n_levels) {
        const float x[3] = {x01[px * 3], x01[px * 3 + 1], x01[px * 3 + 2]};
        const float2 gv = reinterpret_cast<const float2*>(dfeat + p * (int64_t)ld)[l];
        const float gg[2] = {gv.x, gv.y};
        lookup_dx<2>(table, get_level(g, l), x, gg, dx);
    }
    for (int o = 1; o < (1 << lp_shift); o <<= 1) {          // whole waves reach this (no early return above)
        dx[0] += __shfl_xor(dx[0], o); dx[1] += __shfl_xor(dx[1], o); dx[2] += __shfl_xor(dx[2], o);
    }
    if (p < n && l == 0) { dx01[px * 3] = dx[0]; dx01[px * 3 + 1] = dx[1]; dx01[px * 3 + 2] = dx[2]; }    // zeros past n_sel
}

static inline int lp_shift_of(int n_levels) {
    int s = 0;
    while ((1 << s) < n_levels) ++s;
    return s;
}

static void launch_encode_dx(const rfx_grid_desc& g, const float* table, const float* x01, int64_t n, const float* dfeat, int ld,
                             float* dx01, hipStream_t st, const int* perm = nullptr, const int* n_sel = nullptr) {
    const int sh = lp_shift_of(g.n_levels);
    const int64_t threads = n << sh;
    hipLaunchKernelGGL(grid_encode_dx_lp_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, g, table, x01, n, dfeat, ld,
                       sh, dx01, perm, n_sel);
}

// dfeat rows have stride `ld` floats (>= L*F) so that the field backward can point it at its workspace.
// Hash-grid gradient scatter.  Scattered fp32 atomics are the bottleneck on MI355X (one 64-B memory-
// side request per lane, ~2e10/s chip-wide), so contributions are first reduced inside the wave:
// points arrive ray-major, and a ray visits every grid cell in ONE contiguous run of samples, so lanes
// that share a cell form contiguous segments.  A segmented inclusive scan (shuffles) sums each
// segment's 8x2 corner contributions and only the segment's last lane issues atomics -- ~3-4x fewer
// atomics on the finest levels, >10x on coarse ones.
__global__ __launch_bounds__(256) void grid_encode_backward_kernel(rfx_grid_desc g, const float* __restrict__ table,
                                                                   const float* __restrict__ x01, int64_t n,
                                                                   const float* __restrict__ dfeat, int ld,
                                                                   float* __restrict__ dtable, float* __restrict__ dx01,
                                                                   int dx_accumulate, const int* __restrict__ perm = nullptr,
                                                                   const int* __restrict__ n_sel = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < n && (!n_sel || p < *n_sel);        // whole waves stay alive for the shuffles
    const int64_t px = (perm && p < n) ? perm[p] : p;          // dfeat row p belongs to point px
    float x[3] = {0.5f, 0.5f, 0.5f};
    if (valid) { x[0] = x01[px * 3]; x[1] = x01[px * 3 + 1]; x[2] = x01[px * 3 + 2]; }
    const float* gr = dfeat + (valid ? p : 0) * (int64_t)ld;
    float dx[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < g.n_levels; ++l) {
        const Level lv = get_level(g, l);
        float2 gv = reinterpret_cast<const float2*>(gr)[l];
        if (!valid) gv = make_float2(0.f, 0.f);
        if (dtable) {
            const Cell c = locate(lv, x);
            // segment structure of this level
            const unsigned p0 = __shfl_up(c.g[0], 1), p1 = __shfl_up(c.g[1], 1), p2 = __shfl_up(c.g[2], 1);
            const bool head = lane == 0 || p0 != c.g[0] || p1 != c.g[1] || p2 != c.g[2];
            const unsigned long long heads = __ballot(head);
            const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
            // start lane of my segment = highest set bit of heads at or below my lane
            const unsigned long long below = heads & (~0ull >> (63 - lane));
            const int start = 63 - __clzll((long long)below);
            int run = lane - start + 1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) run = max(run, __shfl_xor(run, o));   // longest segment in the wave
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = corner_weight(c, k);
                float v0 = w * gv.x, v1 = w * gv.y;
                for (int d = 1; d < run; d <<= 1) {                                  // wave-uniform trip count
                    const float t0 = __shfl_up(v0, d), t1 = __shfl_up(v1, d);
                    if (lane - d >= start) { v0 += t0; v1 += t1; }
                }
                if (tail && (v0 != 0.f || v1 != 0.f)) {
                    float* t = dtable + ((size_t)lv.offset + corner_index(lv, c, k)) * 2;
                    atomicAdd(t, v0);
                    atomicAdd(t + 1, v1);
                }
            }
        }
        if (dx01 && valid) {
            const float gg[2] = {gv.x, gv.y};
            lookup_dx<2>(table, lv, x, gg, dx);
        }
    }
    if (dx01 && p < n) {            // rows past n_sel belong to points without a gradient: they get zeros
        if (dx_accumulate) { dx01[px * 3] += dx[0]; dx01[px * 3 + 1] += dx[1]; dx01[px * 3 + 2] += dx[2]; }
        else { dx01[px * 3] = dx[0]; dx01[px * 3 + 1] = dx[1]; dx01[px * 3 + 2] = dx[2]; }
    }
}

// ---------------------------------------------------------------- E1 backward, LDS-privatised scatter
// Scattered fp32 atomics are executed by the memory side one lane at a time; the same additions done in
// LDS cost a few cycles and the table is then updated with *contiguous* atomics (one 256 B request per
// wave), which the memory side retires ~16x faster.  Each block owns one segment of one level
// (SCATTER_SEG entries = 128 KB of LDS) and one slice ("chunk") of the points; it walks its points, keeps
// the corner contributions that land in its segment, and flushes the non-zero accumulators.
// A level of `size` entries is cut into ceil(size / SCATTER_SEG) segments: 1..4 for T = 2^16.
//
// Points arrive ray-major (or lattice-major for the TV term), so neighbours share grid cells on the
// coarse levels -- and LDS serialises same-address atomics.  Two measures: (1) a thread owns a *run* of
// K consecutive points and accumulates the 8x2 corner sums in registers while the cell stays the same,
// touching LDS only when it changes; (2) the lanes of a wave are therefore K points apart.  A staging
// pass re-orders the inputs (chunk, iteration, thread)-major so those strided runs are read coalesced.
#ifndef SCATTER_SEG_ENTRIES
#define SCATTER_SEG_ENTRIES 16384
#endif
#ifndef SCATTER_BATCH
#define SCATTER_BATCH 4
#endif
#ifndef SCATTER_TARGET_BLOCKS
#define SCATTER_TARGET_BLOCKS 512
#endif
constexpr unsigned SCATTER_SEG = SCATTER_SEG_ENTRIES;
constexpr int SCATTER_THREADS = 1024;
#ifndef SCATTER_MAX_SEGMENTS_N
#define SCATTER_MAX_SEGMENTS_N 2048
#endif
constexpr int SCATTER_MAX_SEGMENTS = SCATTER_MAX_SEGMENTS_N;     // T = 2^21: ~1300 segments
constexpr int64_t SCATTER_MIN_POINTS = 4096;

#if defined(SCATTER_DBG) || defined(BIN_DBG) || defined(FIELD_DBG)
#ifndef RFX_DEV_BUILD
#error "SCATTER_DBG / BIN_DBG / FIELD_DBG produce WRONG results on purpose (timing builds): they need -DRFX_DEV_BUILD as well"
#endif
#endif

struct ScatterPlan {
    int seg_start[RFX_MAX_LEVELS + 1];
    int chunks;          // slices of the point list (1 since round 6: the sweep's blocks take ROW ranges, see pos_*)
    int K;               // points per thread and chunk; a chunk covers K * SCATTER_THREADS points
    int64_t slots;       // chunks * K * SCATTER_THREADS
    // Dispatch order of the sweep (round 6).  The grid is one-dimensional; position k holds the blocks of level pos_level[k]:
    // (segments of the level) x pos_parts[k] blocks, block = (segment, part), and a part is a range of the K staged rows (a row =
    // one point of each of the 1 024 threads).  The positions are ordered by the cost of one of their blocks, dearest first,
    // and the parts are chosen per level class so that the whole sweep is ONE round of blocks on the chip where it can be
    // (sweep_plan() below).
    unsigned swept;      // bit l: level l goes through the sweep -- the only planes the staging pass writes (a binned level's sort
                         // reads the gradient rows themselves: at T >= 2^19 two thirds of the staging traffic went unread)
    int n_pos;
    int pos_level[RFX_MAX_LEVELS];
    int pos_parts[RFX_MAX_LEVELS];
    int pos_start[RFX_MAX_LEVELS + 1];
    // The first source may be shorter on the device than on the host (the field backward puts the points with a non-zero
    // loss gradient first and counts them, n_sel): the kernels then deal out min(n_a, *n_sel) points, K_eff <= K per thread.
    int64_t n_a, n_b;
    const int* n_sel;
};

struct ScatterShare { int64_t n_a, per_a, per_b; int K; };
__device__ __forceinline__ ScatterShare scatter_share(const ScatterPlan& p) {
    ScatterShare s;
    s.n_a = p.n_sel ? min(p.n_a, (int64_t)*p.n_sel) : p.n_a;
    s.per_a = (s.n_a + p.chunks - 1) / p.chunks;
    s.per_b = (p.n_b + p.chunks - 1) / p.chunks;
    s.K = (int)((s.per_a + s.per_b + SCATTER_THREADS - 1) / SCATTER_THREADS);
    return s;
}

// The sweep's schedule (round 6; rounds 2-5 cut the point list into `chunks` slices and launched segments x chunks equal-looking
// blocks, picked by a two-parameter model whose fixed cost was a tenth of the measured one: 510 blocks at office0, two rounds on
// 256 CUs, the dear blocks of the small dense levels starting at 43 us of a 120 us launch -- tools/scatter_prof.py).
// One block per CU (128 KB of LDS) and a block costs fixed(level) + points * per_point(level):
//   fixed      zeroing and flushing the segment's accumulators: ~12 us for a full 8 192-entry segment
//   per_point  a hashed level cut into >= 8 segments (membership test first, few corners land): 0.7 ns;
//              the other levels (every corner lands, same-cell runs merged in registers, same-address atomics): 1.2-1.6 ns
// (measured with -DSCATTER_PROF on office0's 201 k points).  A block takes a RANGE OF ROWS of the staged points (row i = point
// t K + i of every thread t: any range has the sources' mix), `parts` ranges per segment of its level -- at office0 2 parts for the
// 88 segments of the cut levels and 4-5 for the 14 of the dense ones: ~240 blocks, ONE round, the dearest first.
struct SweepCost { double fixed, per_point; bool cut; };
static SweepCost sweep_cost(const rfx_grid_desc& g, int l, unsigned seg_entries, bool f64) {
    SweepCost c;
    const unsigned n_seg = (g.size[l] + seg_entries - 1) / seg_entries;
    c.cut = g.hashed[l] && g.size[l] >= (f64 ? 8u : 16u) * seg_entries;
    const double fill = (double)std::min(g.size[l], seg_entries) / seg_entries;
    c.fixed = 2.0 + 10.0 * fill;
    c.per_point = (c.cut ? 0.7e-3 : n_seg == 1 ? 1.6e-3 : 1.2e-3) * (f64 ? 1.0 : 3.0);      // (ds_add_f32: tools/micro/lds_atomic)
    return c;
}

// fills plan->n_pos / pos_level / pos_parts / pos_start for the non-binned levels; returns the number of blocks.
// parts of a level = the fewest that bring its block under T, T = the smallest block time for which all blocks fit on the chip
// at once (bisection: the block count falls as T rises); then the levels in the order of their block cost, dearest first.
// When no T fits one round (more segments than CUs) every level gets one part: blocks of the table's size are then many and
// short, and the dispatcher's order does the rest.
static int sweep_plan(const rfx_grid_desc& g, const bool* binned, unsigned seg_entries, bool f64, int64_t n_est, int K, int cus,
                      ScatterPlan* plan) {
    int lv[RFX_MAX_LEVELS], segs[RFX_MAX_LEVELS], parts[RFX_MAX_LEVELS], n_lv = 0;
    SweepCost sc[RFX_MAX_LEVELS];
    double walk[RFX_MAX_LEVELS];
    const int cap = std::max(1, std::min(K, 128));
    int min_blocks = 0;
    double t_hi = 0.0, t_lo = 0.0;
    for (int l = 0; l < g.n_levels; ++l)
        if (!binned[l]) {
            lv[n_lv] = l; sc[n_lv] = sweep_cost(g, l, seg_entries, f64);
            segs[n_lv] = (int)((g.size[l] + seg_entries - 1) / seg_entries);
            walk[n_lv] = (double)n_est * sc[n_lv].per_point;
            min_blocks += segs[n_lv];
            t_hi = std::max(t_hi, sc[n_lv].fixed + walk[n_lv]);
            t_lo = std::max(t_lo, sc[n_lv].fixed + walk[n_lv] / cap);
            ++n_lv;
        }
    auto blocks_at = [&](double T, int* out) {
        int total = 0;
        for (int i = 0; i < n_lv; ++i) {
            const double room = T - sc[i].fixed;
            int p = room > 0 ? (int)std::ceil(walk[i] / room - 1e-9) : cap;
            p = std::max(1, std::min(cap, p));
            if (out) out[i] = p;
            total += segs[i] * p;
        }
        return total;
    };
    if (min_blocks > cus || n_lv == 0) {
        for (int i = 0; i < n_lv; ++i) parts[i] = 1;
    } else {
        double lo = t_lo, hi = t_hi;           // blocks_at(hi) = min_blocks <= cus
        if (blocks_at(lo, nullptr) <= cus) hi = lo;
        for (int it = 0; it < 40 && hi - lo > 0.05; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (blocks_at(mid, nullptr) <= cus) hi = mid; else lo = mid;
        }
        blocks_at(hi, parts);
    }
    int order[RFX_MAX_LEVELS];
    double c_l[RFX_MAX_LEVELS];
    for (int i = 0; i < n_lv; ++i) { c_l[i] = sc[i].fixed + walk[i] / parts[i]; order[i] = i; }
    std::stable_sort(order, order + n_lv, [&](int x, int y) { return c_l[x] > c_l[y]; });
    plan->n_pos = n_lv;
    int total = 0;
    for (int k = 0; k < RFX_MAX_LEVELS; ++k) {
        plan->pos_start[k] = total;
        if (k < n_lv) {
            plan->pos_level[k] = lv[order[k]]; plan->pos_parts[k] = parts[order[k]];
            total += segs[order[k]] * parts[order[k]];
        } else {
            plan->pos_level[k] = 0; plan->pos_parts[k] = 1;
        }
    }
    plan->pos_start[RFX_MAX_LEVELS] = total;
    static const bool debug = getenv("RFX_DEBUG_SWEEP") != nullptr;
    if (debug) {
        fprintf(stderr, "[sweep] %lld points, %d rows, %d CUs, %d blocks:", (long long)n_est, K, cus, total);
        for (int k = 0; k < n_lv; ++k) fprintf(stderr, " L%d x%d (%.0f us)", lv[order[k]], parts[order[k]], c_l[order[k]]);
        fprintf(stderr, "\n");
    }
    return total;
}

// upper bound of the staged slots (one chunk since round 6: n rounded up to whole rows of 1 024; the bound still covers the
// chunked layouts of rounds 2-5).  The same
// buffer holds the records of the binned levels afterwards, one level at a time when it is no larger than this minimum:
// 24 floats per point + the [blk][seg] counters of the level with the most segments the binned path accepts -- more than
// the sweep's (2 L + 3) floats per point when the grid has 8 levels or fewer (a sub-grid of a level-partitioned table).
static size_t scatter_scratch_floats(int64_t n, int n_levels) {
    const size_t slots = (size_t)n + (size_t)((n + SCATTER_MIN_POINTS - 1) / SCATTER_MIN_POINTS + 1) * SCATTER_THREADS;
    const size_t sweep = slots * (size_t)(2 * n_levels + 3);
    if (n < SCATTER_MIN_POINTS) return sweep;              // below it nothing is staged or binned (direct atomics)
    const size_t n_blk = (size_t)((n + 2 * 1024 - 1) / (2 * 1024));         // BIN_THREADS * BIN_PPT_DENSE points per sort block (a hashed level's take twice as many)
    const size_t one_binned = n_blk * (4 * 1024 * 8) * 4 + n_blk * (1024 + 1) + 8;           // BIN_BLOCK_RECS 16-byte records; SCATTER_BIN_MAX_SEGMENTS = 1024
    return std::max(sweep, one_binned);
}

// LDS slot of table entry r of a segment.  On the dense levels the vertices of neighbouring cells are res or res^2
// entries apart -- multiples of 16 for res = 16, 20, 24, 36, ... -- so points that differ only in y or z (a TV lattice,
// rays along an axis) would all hit ONE bank; XOR-ing the low four bits with the next three nibbles spreads them
// (a permutation inside every aligned group of 16 entries).
// Hashed levels scatter their entries anyway: the permutation (six instructions per corner on the flush path) is only
// applied on dense levels (pm = 15) and is the identity otherwise (pm = 0); a block works on one level, so pm is uniform.
__device__ __forceinline__ unsigned lds_slot(unsigned r, unsigned pm) { return r ^ (((r >> 4) ^ (r >> 8) ^ (r >> 12)) & pm); }

// slot (chunk c, iteration i, thread t) <- point c*K*1024 + t*K + i.  Planes: L x float2[slots], then x,y,z.
// The point list is the concatenation of up to two sources (e.g. the ray samples and the TV lattice), so
// that one sweep over the table segments serves both.
struct ScatterSrc {
    const float* dfeat; int ld; const float* x01; int64_t n;
    const int* perm;          // row p of dfeat belongs to point perm[p] of x01 (null: p)
    const int* n_sel;         // device-side row count, <= n (null: n)
};

// Second stage of the decoder's weight gradients (field_dw_reduce_kernel below: 16 slices of the partial list, each added in
// order, then the slices in order) for a 256-thread block: 16 outputs x 16 slices, the same sums in the same order.  It rides
// in the staging launch of the table scatter, which follows the weight-gradient kernel in a BA iteration and has nothing
// to do with it: one kernel boundary less.
constexpr int DW_TOTAL = N_H * N_IN1 + N_OUT2 * N_H + N_H * N_IN3 + N_OUT4 * N_H;   // 5312
struct DwJob { const float* partial; int n_partials; float *dw1, *dw2, *dw3, *dw4; };
constexpr int DW_JOB_BLOCKS = (DW_TOTAL + 15) / 16;

__device__ __forceinline__ void dw_reduce_overwrite_16(const DwJob& j, int block, float (*red)[16]) {
    const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int i = block * 16 + col;
    float s = 0.f;
    if (i < DW_TOTAL) {
#pragma unroll 8
        for (int k = slice; k < j.n_partials; k += 16) s += j.partial[(size_t)k * DW_TOTAL + i];
    }
    red[slice][col] = s;
    __syncthreads();
    if (slice != 0 || i >= DW_TOTAL) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][col];
    const int o1 = N_H * N_IN1, o2 = o1 + N_OUT2 * N_H, o3 = o2 + N_H * N_IN3;
    float* d = i < o1 ? (j.dw1 ? j.dw1 + i : nullptr) : i < o2 ? (j.dw2 ? j.dw2 + (i - o1) : nullptr)
             : i < o3 ? (j.dw3 ? j.dw3 + (i - o2) : nullptr) : (j.dw4 ? j.dw4 + (i - o3) : nullptr);
    if (d) *d = s;
}

__global__ __launch_bounds__(256) void scatter_stage_kernel(ScatterSrc a, ScatterSrc b, ScatterPlan plan, int n_levels,
                                                            float* __restrict__ scratch, int nb_stage, DwJob dw) {
    if ((int)blockIdx.x >= nb_stage) {
        __shared__ float red[16][16];
        dw_reduce_overwrite_16(dw, (int)blockIdx.x - nb_stage, red);
        return;
    }
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t slots = plan.slots;
    if (s >= slots) return;
    const ScatterShare sh = scatter_share(plan);
    const int64_t per = (int64_t)plan.K * SCATTER_THREADS;
    const int64_t c = s / per, r = s % per;
    if (r / SCATTER_THREADS >= sh.K) return;                                // iterations the sweep never reads
    const int64_t j = (r % SCATTER_THREADS) * sh.K + r / SCATTER_THREADS;   // point j of chunk c
    // every chunk takes an equal share of BOTH sources (a lattice dumped into one chunk would make that
    // chunk's blocks the tail of the launch: its points collide on the coarse levels)
    const bool in_a = j < sh.per_a;
    const ScatterSrc src = in_a ? a : b;
    const int64_t jj = in_a ? j : j - sh.per_a;
    const int64_t p = c * (in_a ? sh.per_a : sh.per_b) + jj;
    const bool live = jj < (in_a ? sh.per_a : sh.per_b) && p < (in_a ? sh.n_a : b.n);
    float2* __restrict__ planes = reinterpret_cast<float2*>(scratch);
    float* __restrict__ xs = scratch + (size_t)slots * 2 * n_levels;
    if (live) {
        const float* __restrict__ rowf = src.dfeat + p * (int64_t)src.ld;
        if (n_levels == RFX_MAX_LEVELS && (src.ld & 3) == 0 && (((uintptr_t)src.dfeat) & 15) == 0) {
            // the whole 128-byte row in eight 16-byte loads issued together, then the sixteen plane stores
            float4 r4[8];
            if (plan.swept == 0xFFFFu) {           // every level swept (T <= 2^16): straight-line, all loads in flight at once
#pragma unroll
                for (int q = 0; q < 8; ++q) r4[q] = reinterpret_cast<const float4*>(rowf)[q];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    planes[(int64_t)(2 * q) * slots + s] = make_float2(r4[q].x, r4[q].y);
                    planes[(int64_t)(2 * q + 1) * slots + s] = make_float2(r4[q].z, r4[q].w);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if ((plan.swept >> (2 * q)) & 3u) r4[q] = reinterpret_cast<const float4*>(rowf)[q];          // (uniform)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if ((plan.swept >> (2 * q)) & 1u) planes[(int64_t)(2 * q) * slots + s] = make_float2(r4[q].x, r4[q].y);
                    if ((plan.swept >> (2 * q + 1)) & 1u) planes[(int64_t)(2 * q + 1) * slots + s] = make_float2(r4[q].z, r4[q].w);
                }
            }
        } else {
            const float2* __restrict__ row = reinterpret_cast<const float2*>(rowf);
            for (int l = 0; l < n_levels; ++l)
                if ((plan.swept >> l) & 1u) planes[(int64_t)l * slots + s] = row[l];
        }
        const int64_t px = src.perm ? src.perm[p] : p;
        xs[s] = src.x01[px * 3]; xs[slots + s] = src.x01[px * 3 + 1]; xs[2 * slots + s] = src.x01[px * 3 + 2];
    } else {
        for (int l = 0; l < n_levels; ++l)
            if ((plan.swept >> l) & 1u) planes[(int64_t)l * slots + s] = make_float2(0.f, 0.f);
        xs[s] = 0.5f; xs[slots + s] = 0.5f; xs[2 * slots + s] = 0.5f;
    }
}

// ACC / SEG: float accumulators over 16 384-entry segments, or double accumulators over 8 192-entry segments (both
// 128 KB).  ds_add_f32 retires one lane per ~2.6 clocks per CU, ds_add_f64 a whole wave in 19 (tools/micro/lds_atomic.hip):
// with doubles the atomics stop being the bound, but every point is visited by twice as many blocks, so the index
// arithmetic doubles.  Measured (tools/time_scatter.py, merged ray + TV points): T = 2^16 0.23 -> 0.15 ms with doubles,
// T = 2^19 0.93 -> 1.21 ms, T = 2^21 1.5 -> 2.6 ms: doubles are used while no level is cut into 16 or more float segments.
#ifdef SCATTER_PROF      // dev builds only (tools/scatter_prof.py, tools/bin_prof.py): start / end clock of every block of the last sweep / sort / reduce
__device__ unsigned long long g_scatter_prof[2 * 8192];
__device__ unsigned long long g_bin_prof[2][2 * 8192];
struct ProfStamp {       // thread 0's entry and exit (whichever return it takes)
    unsigned long long* p;
    __device__ explicit ProfStamp(unsigned long long* q) : p(q) { if (threadIdx.x == 0 && p) p[0] = wall_clock64(); }
    __device__ ~ProfStamp() { if (threadIdx.x == 0 && p) p[1] = wall_clock64(); }
};
#endif

template <typename ACC, unsigned SEG>
__global__ __launch_bounds__(SCATTER_THREADS) void grid_scatter_lds_kernel(rfx_grid_desc g, ScatterPlan plan, int n_levels,
                                                                           const float* __restrict__ scratch,
                                                                           float* __restrict__ dtable) {
    extern __shared__ __attribute__((aligned(16))) unsigned char acc_raw[];
    ACC* acc = reinterpret_cast<ACC*>(acc_raw);
#ifdef SCATTER_PROF
    const unsigned pid = blockIdx.x;
    if (threadIdx.x == 0 && pid < 8192) g_scatter_prof[2 * pid] = wall_clock64();
#endif
    int pos = 0;
    while (pos + 1 < plan.n_pos && (int)blockIdx.x >= plan.pos_start[pos + 1]) ++pos;
    const int l = plan.pos_level[pos];
    const int n_seg = plan.seg_start[l + 1] - plan.seg_start[l];
    const int seg = ((int)blockIdx.x - plan.pos_start[pos]) % n_seg, part = ((int)blockIdx.x - plan.pos_start[pos]) / n_seg;
    const Level lv = get_level(g, l);
    const unsigned pm = lv.hashed ? 0u : 15u;
    const unsigned base = (unsigned)seg * SEG;
    const unsigned cnt = min(SEG, lv.size - base);
    for (unsigned i = threadIdx.x; i < ((cnt + 15u) & ~15u) * 2; i += SCATTER_THREADS) acc[i] = (ACC)0;
    __syncthreads();
    const int64_t slots = plan.slots;
    const float2* __restrict__ gvp = reinterpret_cast<const float2*>(scratch) + (int64_t)l * slots;
    const float* __restrict__ xs = scratch + (size_t)slots * 2 * n_levels;
    // this block's rows [r0, r0 + K_eff) of the K_all staged ones (K_all <= plan.K: see ScatterPlan)
    const int K_all = scatter_share(plan).K;
    const int rows_per = (K_all + plan.pos_parts[pos] - 1) / plan.pos_parts[pos];
    const int r0 = min(part * rows_per, K_all);
    const int K_eff = min(rows_per, K_all - r0);
    int64_t s = (int64_t)r0 * SCATTER_THREADS + threadIdx.x;

    Cell cur;                       // cell of the running register accumulation
    bool open = false;
    float a0[8], a1[8];
    auto flush = [&]() {
        unsigned idx8[8];
        corner_indices(lv, cur, idx8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned r = idx8[k] - base;
#if defined(SCATTER_DBG) && SCATTER_DBG == 2
            if (r < cnt) { acc[2 * lds_slot(r, pm)] = a0[k]; acc[2 * lds_slot(r, pm) + 1] = a1[k]; }
#elif defined(SCATTER_DBG) && SCATTER_DBG == 4
            if (r == 0x7fffffffu) { acc[2 * lds_slot(r, pm)] = a0[k]; acc[2 * lds_slot(r, pm) + 1] = a1[k]; }
#else
            if (r < cnt) {
                atomicAdd(&acc[2 * lds_slot(r, pm)], (ACC)a0[k]);
                atomicAdd(&acc[2 * lds_slot(r, pm) + 1], (ACC)a1[k]);
            }
#endif
        }
    };
    // The loop is bound by the latency of its streaming loads (16 waves per CU, one 128 KB block), so the points are
    // taken SCATTER_BATCH at a time with the next batch's loads already in flight while this one is processed.
    constexpr int NB = SCATTER_BATCH;
    float2 gv_n[NB];
    float xn[NB][3];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool live = i0 + j < K_eff;
            const int64_t sj = s + (int64_t)j * SCATTER_THREADS;
            gv_n[j] = live ? gvp[sj] : make_float2(0.f, 0.f);
            xn[j][0] = live ? xs[sj] : 0.5f; xn[j][1] = live ? xs[slots + sj] : 0.5f; xn[j][2] = live ? xs[2 * slots + sj] : 0.5f;
        }
        s += (int64_t)NB * SCATTER_THREADS;
    };
    fetch(0);
    for (int i = 0; i < K_eff; i += NB) {
        float2 gvb[NB];
        float xb[NB][3];
#pragma unroll
        for (int j = 0; j < NB; ++j) { gvb[j] = gv_n[j]; xb[j][0] = xn[j][0]; xb[j][1] = xn[j][1]; xb[j][2] = xn[j][2]; }
        if (i + NB < K_eff) fetch(i + NB);
        if (lv.hashed && lv.size >= (sizeof(ACC) == 8 ? 8u : 16u) * SEG) {
            // hashed level cut into many segments (>= 16 with float accumulators, >= 8 with double ones, whose atomics are
            // cheap enough that merging runs in registers no longer pays): nearly every corner falls into another
            // segment, so test segment membership first and add only the corners that land here
            // (measured: 1.25-1.6x faster at 128 segments, neutral at 32, slower at 4)
            if (lv.res <= SEG) {
                // Round 6.  These blocks are bound by the NUMBER of LDS instructions a wave issues (a ds_add costs the CU ~10
                // clocks however few lanes are active; round 4: profiles/r4_notes.md), and the per-corner form issues 16 per
                // point with an eighth of the lanes in each.  The two corners of an x-PAIR always share a segment (x enters the
                // hash un-multiplied and res <= SEG keeps it below the segment bits inside the unit cube), so a point has 4 candidates, not 8; each
                // lane then serves its in-segment pairs ONE PER ROUND -- first, second, ... -- and the wave goes round while
                // any lane has one left: 2.4 rounds x 4 instructions on average instead of 16.  The contributions are the
                // per-corner form's, bit for bit (same weight products in the same order); only the order of the adds changes.
                const unsigned msk = lv.size - 1u;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float2 gv = gvb[j];
                    const Cell c = locate(lv, xb[j]);
                    const unsigned y0 = c.g[1] * 2654435761u, y1 = y0 + 2654435761u;
                    const unsigned z0 = c.g[2] * 805459861u, z1 = z0 + 805459861u;
                    unsigned i0[4] = {(c.g[0] ^ y0 ^ z0) & msk, (c.g[0] ^ y1 ^ z0) & msk, (c.g[0] ^ y0 ^ z1) & msk, (c.g[0] ^ y1 ^ z1) & msk};
                    const unsigned xx = (c.g[0] ^ (c.g[0] + 1u)) & msk;            // partner index = i0 ^ xx
                    const float fx0 = 1.0f - c.f[0], fx1 = c.f[0];
                    if (__ballot((xx & ~(SEG - 1u)) != 0u) == 0ull) {
                        // No lane's pair can straddle a segment boundary (x + 1 stays below the segment bits: every point inside
                        // the unit cube): one test per pair instead of two, and a round's adds need none at all.  Same contributions.
                        unsigned m = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) m |= ((i0[k] - base) < cnt ? 1u : 0u) << k;
                        if (gv.x == 0.f && gv.y == 0.f) m = 0;
                        while (__ballot(m != 0u)) {                                // wave-uniform
                            const int k = __ffs((int)m) - 1;
                            const unsigned ia = (k & 1) ? ((k & 2) ? i0[3] : i0[1]) : ((k & 2) ? i0[2] : i0[0]);
                            const float wy = (k & 1) ? c.f[1] : 1.0f - c.f[1], wz = (k & 2) ? c.f[2] : 1.0f - c.f[2];
                            if (m) {
                                const unsigned ra = ia - base, rb = (ia ^ xx) - base;
                                const float wa = (fx0 * wy) * wz, wb = (fx1 * wy) * wz;
                                atomicAdd(&acc[2 * ra], (ACC)(wa * gv.x)); atomicAdd(&acc[2 * ra + 1], (ACC)(wa * gv.y));
                                atomicAdd(&acc[2 * rb], (ACC)(wb * gv.x)); atomicAdd(&acc[2 * rb + 1], (ACC)(wb * gv.y));
                            }
                            m &= m - 1u;
                        }
                        continue;
                    }
                    unsigned m = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) m |= (((i0[k] - base) < cnt || ((i0[k] ^ xx) - base) < cnt) ? 1u : 0u) << k;
                    if (gv.x == 0.f && gv.y == 0.f) m = 0;
                    while (__ballot(m != 0u)) {                                    // wave-uniform
                        const int k = __ffs((int)m) - 1;
                        const unsigned ia = (k & 1) ? ((k & 2) ? i0[3] : i0[1]) : ((k & 2) ? i0[2] : i0[0]);
                        const float wy = (k & 1) ? c.f[1] : 1.0f - c.f[1], wz = (k & 2) ? c.f[2] : 1.0f - c.f[2];
                        if (m) {
                            const unsigned ra = ia - base, rb = (ia ^ xx) - base;
                            const float wa = (fx0 * wy) * wz, wb = (fx1 * wy) * wz;   // corner_weight()'s products, in its order
                            // (a pair straddles two segments only where x + 1 crosses a multiple of SEG: a point far outside the
                            // unit cube; each segment's block then adds its own corner)
                            if (ra < cnt) { atomicAdd(&acc[2 * ra], (ACC)(wa * gv.x)); atomicAdd(&acc[2 * ra + 1], (ACC)(wa * gv.y)); }
                            if (rb < cnt) { atomicAdd(&acc[2 * rb], (ACC)(wb * gv.x)); atomicAdd(&acc[2 * rb + 1], (ACC)(wb * gv.y)); }
                        }
                        m &= m - 1u;
                    }
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float2 gv = gvb[j];
                if (gv.x == 0.f && gv.y == 0.f) continue;
                const Cell c = locate(lv, xb[j]);
                unsigned idx8[8];
                corner_indices(lv, c, idx8);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned r = idx8[k] - base;
                    if (r < cnt) {
                        const float w = corner_weight(c, k);
                        atomicAdd(&acc[2 * lds_slot(r, pm)], (ACC)(w * gv.x));
                        atomicAdd(&acc[2 * lds_slot(r, pm) + 1], (ACC)(w * gv.y));
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float2 gv = gvb[j];
            if (gv.x == 0.f && gv.y == 0.f) continue;
            const Cell c = locate(lv, xb[j]);
            const bool same = open && c.g[0] == cur.g[0] && c.g[1] == cur.g[1] && c.g[2] == cur.g[2];
            if (!same) {
                if (open) flush();
                cur = c;
                open = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) { a0[k] = 0.f; a1[k] = 0.f; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = corner_weight(c, k);
                a0[k] += w * gv.x;
                a1[k] += w * gv.y;
            }
        }
    }
    if (open) flush();
    __syncthreads();
    float* __restrict__ out = dtable + ((size_t)lv.offset + base) * 2;
    for (unsigned i = threadIdx.x; i < cnt * 2; i += SCATTER_THREADS) {
        const float v = (float)acc[2 * lds_slot(i >> 1, pm) + (i & 1)];
#if defined(SCATTER_DBG) && SCATTER_DBG == 1
        if (v == 12345.f) out[i] = v;
#elif defined(SCATTER_DBG) && SCATTER_DBG == 3
        if (v != 0.f) out[i] = v;
#else
        if (v != 0.f) atomicAdd(out + i, v);
#endif
    }
#ifdef SCATTER_PROF
    __syncthreads();
    if (threadIdx.x == 0 && pid < 8192) g_scatter_prof[2 * pid + 1] = wall_clock64();
#endif
}

// ---------------------------------------------------------------- E1 backward, binned scatter (large tables)
// A level cut into many LDS segments makes every segment's block scan every point for the few corners that land in it:
// 32-256 scans of the point list per level at T = 2^19-2^21.  Such a level (>= SCATTER_BIN_MIN_SEGMENTS segments) is
// instead sorted by segment first, block by block (the levels in groups, see BinLevels):
//   bin_sort   : a block takes 4 096 points, ONE walk: it locates and hashes every corner once, ranks each x-PAIR of corners
//                inside its segment with an LDS counter (the rank stays in a register), scans the <= 1 024 counters in LDS and
//                writes each pair as a 16-byte record at `offset of its segment + rank` inside the block's OWN region: the
//                records of a block come out sorted by segment, and the block leaves its exclusive offsets [n_seg + 1] for the
//                readers.  No global histogram, no global scan, no second walk over the points (rounds 3-5 ran bin_count ->
//                bin_scan -> bin_records: the points located and hashed twice, and one 1 024-thread block per level of pure
//                latency in between).
//   bin_reduce : a block owns one segment and one share of the sort blocks; its 16 waves walk the runs those blocks hold for the
//                segment, add them into 128 KB of LDS (double accumulators) and add the non-zero sums to the table with
//                contiguous float atomics.
// Why pairs.  The kernels move records at the memory system's pace (measured, round 6: 12-byte records per corner, 96 B per
// point and level written and read again, at 3.7 TB/s), so the record is what to shrink.  The two corners of a cell that
// differ in x only fall into the SAME segment: on a dense level they are neighbours in the table, on a hashed one the x
// coordinate enters the index un-multiplied (tiny-cuda-nn's first prime is 1) and a level's resolution is below the 8 192
// entries of a segment, so x never reaches the segment bits.  One record (both slots, fx, w_yz g0, w_yz g1) serves both:
// 64 B per point and level instead of 96, and half the rank atomics.  A pair that does straddle a segment boundary (a
// dense level's entries 8 191 | 8 192 mod 8 192, or a resolution above 8 192) becomes two one-corner records.
#ifndef SCATTER_BIN_MIN_SEGMENTS
#define SCATTER_BIN_MIN_SEGMENTS 12      // with grouped launches: 12 against 16 = -5 % at scene0000 (its 15-segment level), 10 and 8 no better
#endif
#ifndef SCATTER_BIN_MAX_SEGMENTS
#define SCATTER_BIN_MAX_SEGMENTS 1024
#endif
constexpr int BIN_THREADS = 1024;
#ifndef BIN_SEG_SHIFT_N
#define BIN_SEG_SHIFT_N 13
#endif
constexpr unsigned BIN_SEG_SHIFT = BIN_SEG_SHIFT_N, BIN_SEG = 1u << BIN_SEG_SHIFT;        // 8 192 entries x 2 doubles = 128 KB
constexpr int BIN_MAX_SEGS = 1024;

// slots: slot of the x = 0 corner | slot of the x = 1 corner << 13 | (1 << 26 when the second corner is part of the record).
// The corners receive (1 - fx) * (a, b) and fx * (a, b); a one-corner record has fx = 0 and its whole weight in (a, b).
struct __attribute__((aligned(16))) BinRec { unsigned slots; float fx, a, b; };
static_assert(sizeof(BinRec) == 16, "records are moved as one 16-byte access");
constexpr unsigned BIN_REC_FLOATS = sizeof(BinRec) / sizeof(float);

// The BIN_PPT points of a thread (j0, j0 + stride, ...), all loads issued together: a load under a per-lane condition makes the
// compiler wait for the one before it (the destination keeps its old value in the masked lanes), which turned a thread's
// points into a chain of 3 BIN_PPT round trips.  So every lane loads -- inactive ones from a clamped, valid row -- and the
// conditions select afterwards: two round trips per thread (gradient rows + selection, then positions).
template <int N>
__device__ __forceinline__ void bin_load(const ScatterSrc& a, const ScatterSrc& b, int64_t j0, int64_t stride, int level,
                                         float x[N][3], float2 gv[N], bool act[N]) {
    const int64_t na = a.n_sel ? min(a.n, (int64_t)*a.n_sel) : a.n;       // (only the first source carries a selection)
    int64_t pa[N], pb[N];
    bool in_a[N];
    int px[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int64_t j = j0 + q * stride;
        in_a[q] = j < a.n;
        const int64_t p = in_a[q] ? j : j - a.n;
        act[q] = in_a[q] ? p < na : p < b.n;
        pa[q] = in_a[q] && act[q] ? p : 0;                                    // row 0 exists whenever the source has points
        pb[q] = !in_a[q] && act[q] ? p : 0;
    }
    // straight-line loads: a source without points lends the other one's (valid) arrays to the clamped row-0 loads
    const float* __restrict__ dfa = a.n > 0 ? a.dfeat : b.dfeat;
    const float* __restrict__ dfb = b.n > 0 ? b.dfeat : a.dfeat;
    const float* __restrict__ xsa = a.n > 0 ? a.x01 : b.x01;
    const float* __restrict__ xsb = b.n > 0 ? b.x01 : a.x01;
    const int64_t lda = a.n > 0 ? a.ld : b.ld, ldb = b.n > 0 ? b.ld : a.ld;
    const int* __restrict__ perm = a.n > 0 ? a.perm : nullptr;
    float2 ga[N], gb[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        ga[q] = reinterpret_cast<const float2*>(dfa + pa[q] * lda)[level];
        gb[q] = reinterpret_cast<const float2*>(dfb + pb[q] * ldb)[level];
        px[q] = (int)pa[q];
    }
    if (perm) {                       // block-uniform
#pragma unroll
        for (int q = 0; q < N; ++q) px[q] = perm[pa[q]];
    }
    float xa[N][3], xb[N][3];
#pragma unroll
    for (int q = 0; q < N; ++q) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { xa[q][d] = xsa[(int64_t)px[q] * 3 + d]; xb[q][d] = xsb[pb[q] * 3 + d]; }
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
        gv[q] = in_a[q] ? ga[q] : gb[q];
        x[q][0] = in_a[q] ? xa[q][0] : xb[q][0]; x[q][1] = in_a[q] ? xa[q][1] : xb[q][1]; x[q][2] = in_a[q] ? xa[q][2] : xb[q][2];
        act[q] = act[q] && (gv[q].x != 0.f || gv[q].y != 0.f);
        if (!act[q]) { gv[q] = make_float2(0.f, 0.f); x[q][0] = 0.5f; x[q][1] = 0.5f; x[q][2] = 0.5f; }
    }
}

// LDS atomics to ONE address from all 64 lanes serialise (a dense level's neighbouring points fall into the same segment):
// when the whole wave agrees on the segment, one lane adds for all and the lanes take consecutive ranks.
template <bool TRY_UNIFORM>
__device__ __forceinline__ unsigned bin_take(unsigned* counters, unsigned seg, bool active) {
    const unsigned long long m = __ballot(active);
    if (m == 0ull) return 0u;
    if (TRY_UNIFORM) {
        const int leader = __ffsll((long long)m) - 1;
        const unsigned seg0 = __shfl(seg, leader);
        const int lane = threadIdx.x & 63;
        if (__ballot(active && seg != seg0) == 0ull) {          // wave-uniform segment
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&counters[seg0], (unsigned)__popcll(m));
            base = __shfl(base, leader);
            return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    return active ? atomicAdd(&counters[seg], 1u) : 0u;
}

constexpr int BIN_PPT = 4;            // points per thread of the sort kernel: 4 096 points per block
constexpr unsigned BIN_BLOCK_RECS = BIN_THREADS * BIN_PPT * 8;      // records a sort block writes at most (every pair split): its region
static_assert(BIN_THREADS * BIN_PPT == 4 * 1024 && SCATTER_BIN_MAX_SEGMENTS <= 1024 && BIN_MAX_SEGS <= BIN_THREADS, "scatter_scratch_floats() prices one binned level with these; one scan thread per segment");
static_assert(BIN_BLOCK_RECS <= 65536, "ranks are packed as two 16-bit halves");

// The binned levels of one sweep that are in flight TOGETHER (as many as the caller's scratch holds records for):
// blockIdx.y (sort) or a flattened (level, segment, share) index (reduce) says which level a block works on.
struct BinLevels {
    int n;                                   // levels in the group
    Level lv[RFX_MAX_LEVELS];
    int level[RFX_MAX_LEVELS];               // index of the level in the grid (its column pair in dfeat)
    int n_seg[RFX_MAX_LEVELS];
    int n_blk[RFX_MAX_LEVELS];               // sort blocks of the level (a dense level's blocks take BIN_PPT_DENSE points per thread)
    int parts[RFX_MAX_LEVELS];               // reduce blocks per segment: each takes a contiguous share of the sort blocks
    int overwrite[RFX_MAX_LEVELS];           // 1: the level's part of dtable is WRITTEN (every segment by its one block, zeros included),
                                             //    not added to: the caller skipped its zero-fill (scatter_overwrite_from_level)
    int blk_base[RFX_MAX_LEVELS + 1];        // first reduce block of each level (n_seg * parts blocks per level)
    unsigned* excl[RFX_MAX_LEVELS];          // [n_blk][n_seg + 1]: where each segment's run starts inside a sort block's region
    BinRec* rec[RFX_MAX_LEVELS];             // [n_blk][BIN_BLOCK_RECS]
};
#ifndef BIN_DENSE_FACTOR
#define BIN_DENSE_FACTOR 4.0
#endif
#ifndef BIN_CHUNK_RECORDS
#define BIN_CHUNK_RECORDS 32768
#endif
// (pair) records one reduce block is meant to add (sets `parts`).  32 768 since the second session of round 6 (16 384 before): at
// T = 2^19 a hashed level's 64 segments get ~33 k records each from a BA iteration's 530 k points -- ONE block per segment then: a
// third of the blocks' fixed cost (zero + write-back of 128 KB) gone, plain read-modify-write instead of float atomics, and the
// map phase's no-zero-fill / no-read-back form (scatter_overwrite_from_level: needs parts == 1) reaches scene0000's levels too:
// 736 -> 760 frames/s there, nothing at T = 2^21 whose segments (8 k records each) had one block already (tools/r6_chunk_ab.sh)
constexpr unsigned BIN_CHUNK = BIN_CHUNK_RECORDS;

template <bool DENSE>
__device__ __forceinline__ void bin_sort_body(const BinLevels& B, int g, const ScatterSrc& a, const ScatterSrc& b, unsigned* h, unsigned* wsum) {
    const Level lv = B.lv[g];
    const int n_seg = B.n_seg[g], level = B.level[g];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t < n_seg) h[t] = 0u;
    __syncthreads();
    // ---- the one walk: corners, their segment and the pair's rank inside (block, segment)
    unsigned i0[BIN_PPT][4], i1[BIN_PPT][4], rank[BIN_PPT][4];
    float fr[BIN_PPT][3];
    float2 gv[BIN_PPT];
    bool act[BIN_PPT];
    float xq[BIN_PPT][3];
    bin_load<BIN_PPT>(a, b, (int64_t)blockIdx.x * BIN_PPT * BIN_THREADS + t, BIN_THREADS, level, xq, gv, act);
#pragma unroll
    for (int q = 0; q < BIN_PPT; ++q) {
        const Cell c = locate(lv, xq[q]);
        fr[q][0] = c.f[0]; fr[q][1] = c.f[1]; fr[q][2] = c.f[2];
        unsigned idx[8];
        corner_indices(lv, c, idx);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            i0[q][k] = idx[2 * k]; i1[q][k] = idx[2 * k + 1];
            const unsigned s0 = idx[2 * k] >> BIN_SEG_SHIFT, s1 = idx[2 * k + 1] >> BIN_SEG_SHIFT;
#if defined(BIN_DBG) && (BIN_DBG & 8)
            const unsigned r0 = (unsigned)(q * 4 + k), r1 = 0;
#else
            const unsigned r0 = bin_take<DENSE>(h, s0, act[q]);
            const unsigned r1 = bin_take<false>(h, s1, act[q] && s1 != s0);       // (a wave without a split pair leaves at its ballot)
#endif
            rank[q][k] = r0 | r1 << 16;
        }
    }
    __syncthreads();
    // ---- exclusive scan of the segment counters, one thread per segment
    const unsigned v = t < n_seg ? h[t] : 0u;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        const unsigned w = lane < BIN_THREADS / 64 ? wsum[lane] : 0u;
        unsigned wi = w;
#pragma unroll
        for (int d = 1; d < BIN_THREADS / 64; d <<= 1) {
            const unsigned u = __shfl_up(wi, d);
            if (lane >= d) wi += u;
        }
        if (lane < BIN_THREADS / 64) wsum[lane] = wi - w;
    }
    __syncthreads();
    const unsigned excl = incl - v + wsum[wv];
    unsigned* __restrict__ eo = B.excl[g] + (size_t)blockIdx.x * (n_seg + 1);
    if (t < n_seg) { h[t] = excl; eo[t] = excl; }
    if (t == n_seg - 1) eo[n_seg] = excl + v;
    __syncthreads();
    // ---- the records, from the registers
    BinRec* __restrict__ rec = B.rec[g] + (size_t)blockIdx.x * BIN_BLOCK_RECS;
#pragma unroll
    for (int q = 0; q < BIN_PPT; ++q) {
        if (!act[q]) continue;
        const float fx = fr[q][0];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wyz = ((k & 1) ? fr[q][1] : 1.0f - fr[q][1]) * ((k >> 1) ? fr[q][2] : 1.0f - fr[q][2]);
            const unsigned s0 = i0[q][k] >> BIN_SEG_SHIFT, s1 = i1[q][k] >> BIN_SEG_SHIFT;
            const unsigned l0 = i0[q][k] & (BIN_SEG - 1u), l1 = i1[q][k] & (BIN_SEG - 1u);
            const float ga = wyz * gv[q].x, gb = wyz * gv[q].y;
            if (s0 == s1) {
                BinRec r;
                r.slots = l0 | l1 << BIN_SEG_SHIFT | 1u << 26; r.fx = fx; r.a = ga; r.b = gb;
#if defined(BIN_DBG) && (BIN_DBG & 4)
                if (fx == 12345.f)
#endif
                rec[h[s0] + (rank[q][k] & 0xffffu)] = r;
            } else {
                BinRec r;
                r.slots = l0; r.fx = 0.f; r.a = (1.0f - fx) * ga; r.b = (1.0f - fx) * gb;
                rec[h[s0] + (rank[q][k] & 0xffffu)] = r;
                r.slots = l1; r.a = fx * ga; r.b = fx * gb;
                rec[h[s1] + (rank[q][k] >> 16)] = r;
            }
        }
    }
}

// DENSE levels.  Neighbouring points share grid cells there (a ray crosses a cell of a coarse level with a run of consecutive
// samples -- 48 of a ray's 59 samples sit within half a metre of the surface --, the TV lattice's z-neighbours likewise), and one
// record per point and pair means runs of records for ONE table slot: the reduce kernel's double atomics then serialise on a
// single LDS address, lane after lane (measured, round 6: a dense level cost 70-97 us against 45 for a hashed one, with the
// ray samples ALONE costing as much as rays + lattice).  So the lanes of a wave -- consecutive points -- first merge: a
// segmented scan over the runs of lanes that share a cell (and have a gradient) sums the four corner contributions of every
// x-pair, and only the LAST lane of a run ranks and writes records: two one-corner records per pair and run instead of one
// pair record per pair and point.  Same contributions as the per-point form (corner_weight()'s products in its order), added
// first along the run in fp32 (the sweep kernel of the small levels merges its runs the same way).
constexpr int BIN_PPT_DENSE = 2;      // points per thread: the run sums (16 per point) have to stay in registers across the block's scan
__device__ __forceinline__ void bin_sort_dense(const BinLevels& B, int g, const ScatterSrc& a, const ScatterSrc& b, unsigned* h, unsigned* wsum) {
    constexpr int PPT = BIN_PPT_DENSE;
    const Level lv = B.lv[g];
    const int n_seg = B.n_seg[g], level = B.level[g];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t < n_seg) h[t] = 0u;
    __syncthreads();
    unsigned ia[PPT][4], ib[PPT][4], rank[PPT][4];
    float sum[PPT][4][4];
    bool emit[PPT];
    {
        float xq[PPT][3];
        float2 gv[PPT];
        bool act[PPT];
        bin_load<PPT>(a, b, (int64_t)blockIdx.x * PPT * BIN_THREADS + t, BIN_THREADS, level, xq, gv, act);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const Cell c = locate(lv, xq[q]);
            unsigned idx[8];
            corner_indices(lv, c, idx);
            // runs: lanes in a row with the same cell, all with a gradient
            const unsigned p0 = __shfl_up(c.g[0], 1), p1 = __shfl_up(c.g[1], 1), p2 = __shfl_up(c.g[2], 1);
            const int pact = __shfl_up((int)act[q], 1);
            const bool head = lane == 0 || !act[q] || !pact || p0 != c.g[0] || p1 != c.g[1] || p2 != c.g[2];
            const unsigned long long heads = __ballot(head);
            const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
            const int start = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
            int run = lane - start + 1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) run = max(run, __shfl_xor(run, o));      // longest run of the wave
            emit[q] = tail && act[q];
            const float fx0 = 1.0f - c.f[0], fx1 = c.f[0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float wy = (k & 1) ? c.f[1] : 1.0f - c.f[1], wz = (k >> 1) ? c.f[2] : 1.0f - c.f[2];
                const float w0 = (fx0 * wy) * wz, w1 = (fx1 * wy) * wz;
                float v[4] = {w0 * gv[q].x, w0 * gv[q].y, w1 * gv[q].x, w1 * gv[q].y};
                for (int d = 1; d < run; d <<= 1) {                                    // wave-uniform trip count
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float u = __shfl_up(v[i], d);
                        if (lane - d >= start) v[i] += u;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) sum[q][k][i] = v[i];
                ia[q][k] = idx[2 * k]; ib[q][k] = idx[2 * k + 1];
                const unsigned r0 = bin_take<true>(h, idx[2 * k] >> BIN_SEG_SHIFT, emit[q]);
                const unsigned r1 = bin_take<true>(h, idx[2 * k + 1] >> BIN_SEG_SHIFT, emit[q]);
                rank[q][k] = r0 | r1 << 16;
            }
        }
    }
    __syncthreads();
    const unsigned v = t < n_seg ? h[t] : 0u;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        const unsigned w = lane < BIN_THREADS / 64 ? wsum[lane] : 0u;
        unsigned wi = w;
#pragma unroll
        for (int d = 1; d < BIN_THREADS / 64; d <<= 1) {
            const unsigned u = __shfl_up(wi, d);
            if (lane >= d) wi += u;
        }
        if (lane < BIN_THREADS / 64) wsum[lane] = wi - w;
    }
    __syncthreads();
    const unsigned excl = incl - v + wsum[wv];
    unsigned* __restrict__ eo = B.excl[g] + (size_t)blockIdx.x * (n_seg + 1);
    if (t < n_seg) { h[t] = excl; eo[t] = excl; }
    if (t == n_seg - 1) eo[n_seg] = excl + v;
    __syncthreads();
    BinRec* __restrict__ rec = B.rec[g] + (size_t)blockIdx.x * BIN_BLOCK_RECS;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (!emit[q]) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            BinRec r;
            r.fx = 0.f;
            r.slots = ia[q][k] & (BIN_SEG - 1u); r.a = sum[q][k][0]; r.b = sum[q][k][1];
            rec[h[ia[q][k] >> BIN_SEG_SHIFT] + (rank[q][k] & 0xffffu)] = r;
            r.slots = ib[q][k] & (BIN_SEG - 1u); r.a = sum[q][k][2]; r.b = sum[q][k][3];
            rec[h[ib[q][k] >> BIN_SEG_SHIFT] + (rank[q][k] >> 16)] = r;
        }
    }
}

__global__ __launch_bounds__(BIN_THREADS) void bin_sort_kernel(BinLevels B, ScatterSrc a, ScatterSrc b) {
    __shared__ unsigned h[BIN_MAX_SEGS + 1];
    __shared__ unsigned wsum[BIN_THREADS / 64];
    const int g = blockIdx.y;
    if ((int)blockIdx.x >= B.n_blk[g]) return;                          // (the grid is as wide as the level with the most sort blocks)
#ifdef SCATTER_PROF
    const unsigned prof_id = blockIdx.y * gridDim.x + blockIdx.x;
    ProfStamp prof_stamp(prof_id < 8192 ? &g_bin_prof[0][2 * prof_id] : nullptr);
#endif
    if (B.lv[g].hashed) bin_sort_body<false>(B, g, a, b, h, wsum);      // block-uniform
    else bin_sort_dense(B, g, a, b, h, wsum);
}

#ifndef BIN_RUNS_N
#define BIN_RUNS_N 4
#endif
constexpr int BIN_RUNS = BIN_RUNS_N;       // runs a wave of the reduce kernel reads together (2 x 64 records of each in flight)

__device__ __forceinline__ void bin_add(double* acc, const BinRec& q) {
#if defined(BIN_DBG) && (BIN_DBG & 1)
    if (q.fx != 12345.f) return;
#endif
    const unsigned l0 = q.slots & (BIN_SEG - 1u), l1 = (q.slots >> BIN_SEG_SHIFT) & (BIN_SEG - 1u);
    const float u = 1.0f - q.fx;
    atomicAdd(&acc[2 * l0], (double)(u * q.a));
    atomicAdd(&acc[2 * l0 + 1], (double)(u * q.b));
    if (q.slots >> 26) {
        atomicAdd(&acc[2 * l1], (double)(q.fx * q.a));
        atomicAdd(&acc[2 * l1 + 1], (double)(q.fx * q.b));
    }
}

// One block adds, for ONE segment, the runs a contiguous share of the sort blocks hold for it.  Shares, because on a dense
// level a segment is a slab of space and a scene fills a few of them: `parts` blocks per segment keep the fullest segment's
// work spread (launch_binned_levels picks it per level); a (segment, share) without records leaves before it touches its LDS.
// Wave w takes the sort blocks b0 + w, b0 + w + 16, ...: a lane per block fetches the run's bounds up front (no dependent
// round trip per run), then the runs are read BIN_RUNS at a time, 128 records of each, before any of them is added (a block's time is a chain of
// round trips: 16 waves, one block per CU).
__global__ __launch_bounds__(BIN_THREADS) void bin_reduce_kernel(BinLevels B, float* __restrict__ dtable) {
#ifdef SCATTER_PROF
    ProfStamp prof_stamp(blockIdx.x < 8192 ? &g_bin_prof[1][2 * blockIdx.x] : nullptr);
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char acc_raw[];
    double* acc = reinterpret_cast<double*>(acc_raw);
    int g = 0;
    while (g + 1 < B.n && (int)blockIdx.x >= B.blk_base[g + 1]) ++g;
    const Level lv = B.lv[g];
    const int n_seg = B.n_seg[g], parts = B.parts[g], n_blk = B.n_blk[g];
    const unsigned local = blockIdx.x - (unsigned)B.blk_base[g];
    const unsigned seg = local / (unsigned)parts, part = local % (unsigned)parts;
    const int b0 = (int)((int64_t)part * n_blk / parts), b1 = (int)((int64_t)(part + 1) * n_blk / parts);
    const unsigned* __restrict__ ex = B.excl[g];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NW = BIN_THREADS / 64;
    const unsigned base = seg << BIN_SEG_SHIFT;
    const unsigned cnt = min(BIN_SEG, lv.size - base);
    bool zeroed = false;
    // 64 x NW sort blocks per round (one round up to 1 024 sort blocks = 4 M points)
    for (int r0 = b0; r0 < b1; r0 += 64 * NW) {
        const int mine = r0 + wv + NW * lane;            // lane j holds the bounds of the wave's j-th run
        unsigned e0 = 0, e1 = 0;
        if (mine < b1) {
            const unsigned* q = ex + (size_t)mine * (n_seg + 1) + seg;
            e0 = q[0]; e1 = q[1];
        }
        const int any = __syncthreads_or(e1 > e0);       // block-uniform
        if (!any) continue;
        if (!zeroed) {
            for (unsigned i = threadIdx.x; i < cnt * 2; i += BIN_THREADS) acc[i] = 0.0;
            __syncthreads();
            zeroed = true;
        }
        const unsigned long long have = __ballot(e1 > e0);
        const int n_runs = have ? 64 - __clzll((long long)have) : 0;          // wave-uniform
        for (int j = 0; j < n_runs; j += BIN_RUNS) {
            unsigned s0[BIN_RUNS], s1[BIN_RUNS];
            const BinRec* __restrict__ rp[BIN_RUNS];
            BinRec q[BIN_RUNS][2];
            bool ok[BIN_RUNS][2];
#pragma unroll
            for (int u = 0; u < BIN_RUNS; ++u) {
                const int jj = min(j + u, 63);
                s0[u] = __shfl(e0, jj); s1[u] = j + u < 64 ? __shfl(e1, jj) : s0[u];
                rp[u] = B.rec[g] + (size_t)min(r0 + wv + NW * jj, b1 - 1) * BIN_BLOCK_RECS;          // (a valid region for the lanes past the last run)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    ok[u][k] = s0[u] + 64u * k + lane < s1[u];
                    q[u][k] = rp[u][min(s0[u] + 64u * k + lane, BIN_BLOCK_RECS - 1u)];        // unconditional: see bin_load
                }
            }
#pragma unroll
            for (int u = 0; u < BIN_RUNS; ++u)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (ok[u][k]) bin_add(acc, q[u][k]);
#pragma unroll
            for (int u = 0; u < BIN_RUNS; ++u) {          // runs longer than two waves (dense levels): the rest, four loads in flight
                if (s1[u] - s0[u] <= 128u) continue;      // wave-uniform
                for (unsigned r = s0[u] + 128u + lane; r < s1[u]; r += 256u) {
                    BinRec t4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) t4[k] = rp[u][min(r + 64u * k, BIN_BLOCK_RECS - 1u)];
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (r + 64u * k < s1[u]) bin_add(acc, t4[k]);
                }
            }
        }
    }
    float* __restrict__ out = dtable + ((size_t)lv.offset + base) * 2;
    if (!zeroed) {                                        // block-uniform: no record for this (segment, share)
        if (B.overwrite[g]) {                             // ... but the segment is this block's to define
            float4* __restrict__ o4 = reinterpret_cast<float4*>(out);
            for (unsigned i = threadIdx.x; i < cnt / 2; i += BIN_THREADS) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
#if defined(BIN_DBG) && (BIN_DBG & 2)
    if (cnt != 12345u) return;
#endif
    __syncthreads();
    #ifndef BIN_FLUSH_ATOMIC
    if (parts == 1 && (cnt & 1u) == 0u && ((uintptr_t)out & 15) == 0) {
        // this block is the segment's only writer in the launch (and the other kernels of a scatter touch other levels): a plain
        // read-modify-write, 16 bytes per lane.  The float atomics this replaces ran at 2e11 per second chip-wide -- 4.2 M of them
        // per level of 2^21 entries were most of the kernel (round 6: 254 us for 12 levels with them)
        float4* __restrict__ o4 = reinterpret_cast<float4*>(out);
        constexpr int NV = BIN_SEG / 2 / BIN_THREADS;       // 4: all of a thread's loads are issued before the first add
        if (B.overwrite[g]) {
            // the caller left this level's part of the gradient buffer UNINITIALISED (no zero-fill: 16.8 MB per level of 2^21
            // entries not written, and not read back here): the sums are stored, zeros included
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const unsigned i = threadIdx.x + k * BIN_THREADS;
                if (i < cnt / 2) o4[i] = make_float4((float)acc[4 * i], (float)acc[4 * i + 1], (float)acc[4 * i + 2], (float)acc[4 * i + 3]);
            }
            return;
        }
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = o4[min(threadIdx.x + k * BIN_THREADS, cnt / 2 - 1u)];          // unconditional: see bin_load
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const unsigned i = threadIdx.x + k * BIN_THREADS;
            if (i < cnt / 2) {
                v[k].x += (float)acc[4 * i]; v[k].y += (float)acc[4 * i + 1]; v[k].z += (float)acc[4 * i + 2]; v[k].w += (float)acc[4 * i + 3];
                o4[i] = v[k];
            }
        }
        return;
    }
#endif
    for (unsigned i = threadIdx.x; i < cnt * 2; i += BIN_THREADS) {
        const float v = (float)acc[i];
        if (v != 0.f) atomicAdd(out + i, v);                // contiguous float atomics: the memory side's fast path
    }
}

// a DENSE level is binned from 8 segments on: every corner of every point lands in its sweep blocks (2.7 ns per point and
// segment at scene0000, the lattice's points piling onto the coarse cells, against 0.7 ns on a cut hashed level), so its
// share of the sweep outgrows a binned level's cost earlier (scene0000's 10-segment level: merged scatter 559 -> 544 us; 5: the
// same; 3: 570; cafeteria / apartment have no such level)
#ifndef SCATTER_BIN_MIN_SEGMENTS_DENSE
#define SCATTER_BIN_MIN_SEGMENTS_DENSE 8
#endif
static bool level_is_binned(const rfx_grid_desc& g, int l) {
    const unsigned segs = (g.size[l] + BIN_SEG - 1) / BIN_SEG;
    return segs >= (unsigned)(g.hashed[l] ? SCATTER_BIN_MIN_SEGMENTS : SCATTER_BIN_MIN_SEGMENTS_DENSE) && segs <= (unsigned)SCATTER_BIN_MAX_SEGMENTS;
}

static inline size_t bin_sort_blocks(int64_t n_all, bool hashed) {
    const int64_t per = (int64_t)BIN_THREADS * (hashed ? BIN_PPT : BIN_PPT_DENSE);
    return (size_t)((n_all + per - 1) / per);
}

// scratch one binned level needs: the sort blocks' record regions + their exclusive offsets [n_blk][n_seg + 1]
static size_t binned_level_floats(const rfx_grid_desc& g, int l, int64_t n_all) {
    const size_t n_seg = (g.size[l] + BIN_SEG - 1) / BIN_SEG;
    const size_t n_blk = bin_sort_blocks(n_all, g.hashed[l] != 0);
    return n_blk * BIN_BLOCK_RECS * BIN_REC_FLOATS + ((n_blk * (n_seg + 1) + 3) & ~(size_t)3) + 4;      // (records stay 16-byte aligned; + 4: the alignment slack)
}

// the binned levels `levels[0..n_lv)` through the two kernels above, as many levels per group of launches as the scratch
// holds (at least one: the caller's minimum, rfx_grid_encode_backward_workspace_bytes = scatter_scratch_floats, covers one
// level of any admissible segment count for every n_levels)
static int bin_parts(const rfx_grid_desc& g, int l, int64_t n_all) {
    const int n_seg = (int)((g.size[l] + BIN_SEG - 1) / BIN_SEG);
    const int n_blk = (int)bin_sort_blocks(n_all, g.hashed[l] != 0);
    // reduce blocks per segment: ~BIN_CHUNK records each.  A hashed level spreads its 4 n_all pairs evenly over the
    // segments; a dense level's segments are slabs of space of which a scene (and the TV lattice, a small cube) fills
    // a fraction: priced as if a quarter of them held everything
    // (a dense level's runs are merged before they become records: two per pair and run, priced at half a pair record per point)
    const double per_seg = (double)n_all * 4.0 * (g.hashed[l] ? 1.0 : 0.5 * BIN_DENSE_FACTOR) / n_seg;
    return std::max(1, std::min(n_blk, (int)(per_seg / BIN_CHUNK + 0.5)));
}

static int launch_binned_levels(const rfx_grid_desc& g, const int* levels, int n_lv, const ScatterSrc& a, const ScatterSrc& b,
                                float* dtable, float* scratch, size_t scratch_floats, hipStream_t st, int overwrite_from_level) {
    const int64_t n_all = a.n + b.n;
    const size_t lds = (size_t)BIN_SEG * 2 * sizeof(double);
    {   // records are 16-byte accesses, the caller's workspace is 8-byte aligned: binned_level_floats() carries the slack
        const size_t skip = (size_t)((16 - ((uintptr_t)scratch & 15)) & 15) / sizeof(float);
        if (scratch_floats < skip) return RFX_ERR_WORKSPACE;
        scratch += skip; scratch_floats -= skip;
    }
    static bool attr_set[64] = {};            // per device (the attribute is per device; benign if raced)
    static const bool debug = getenv("RFX_DEBUG_BINS") != nullptr;
    int dev = 0;
    RFX_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        RFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(bin_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    int i = 0;
    while (i < n_lv) {
        BinLevels B;
        B.n = 0;
        B.blk_base[0] = 0;
        size_t used = 0;
        int max_blk = 0;
        while (i < n_lv && B.n < RFX_MAX_LEVELS) {
            const int l = levels[i];
            const size_t need = (binned_level_floats(g, l, n_all) + 3) & ~(size_t)3;
            if (used + need > scratch_floats) break;
            const int n_seg = (int)((g.size[l] + BIN_SEG - 1) / BIN_SEG);
            const int n_blk = (int)bin_sort_blocks(n_all, g.hashed[l] != 0);
            const int k = B.n++;
            B.n_blk[k] = n_blk;
            max_blk = std::max(max_blk, n_blk);
            B.lv[k].scale = g.scale[l]; B.lv[k].res = g.res[l]; B.lv[k].size = g.size[l]; B.lv[k].offset = g.offset[l]; B.lv[k].hashed = g.hashed[l];
            B.level[k] = l; B.n_seg[k] = n_seg;
            B.parts[k] = bin_parts(g, l, n_all);
            B.overwrite[k] = l >= overwrite_from_level ? 1 : 0;
            if (B.overwrite[k] && (!g.hashed[l] || B.parts[k] != 1 || ((uintptr_t)dtable & 15))) return RFX_ERR_ARG;      // (scatter_overwrite_from_level's promise)
            B.blk_base[k + 1] = B.blk_base[k] + n_seg * B.parts[k];
            float* base = scratch + used;
            B.rec[k] = reinterpret_cast<BinRec*>(base);
            B.excl[k] = reinterpret_cast<unsigned*>(base + (size_t)n_blk * BIN_BLOCK_RECS * BIN_REC_FLOATS);
            used += need;
            ++i;
        }
        if (debug) fprintf(stderr, "[bins] group of %d levels (%d of %d done), %zu of %zu floats, %lld points\n", B.n, i, n_lv, used, scratch_floats, (long long)n_all);
        if (B.n == 0) return RFX_ERR_WORKSPACE;
        for (int k = B.n + 1; k <= RFX_MAX_LEVELS; ++k) B.blk_base[k] = B.blk_base[B.n];
        for (int k = B.n; k < RFX_MAX_LEVELS; ++k) { B.n_blk[k] = 0; B.overwrite[k] = 0; }
        hipLaunchKernelGGL(bin_sort_kernel, dim3(max_blk, B.n), dim3(BIN_THREADS), 0, st, B, a, b);
        hipLaunchKernelGGL(bin_reduce_kernel, dim3(B.blk_base[B.n]), dim3(BIN_THREADS), lds, st, B, dtable);
        RFX_LAUNCH_CHECK();
    }
    return RFX_OK;
}

// dtable += scatter of dfeat (row stride ld).  With a scratch buffer of scatter_scratch_floats() the
// LDS path is taken when the table is small enough to sweep segment by segment; otherwise direct atomics.
static int launch_grid_scatter(const rfx_grid_desc& g, const float* table, const float* x01, int64_t n, const float* dfeat,
                               int ld, float* dtable, float* scratch, hipStream_t st, const float* x01_b = nullptr,
                               const float* dfeat_b = nullptr, int ld_b = 0, int64_t n_b = 0, const int* perm = nullptr,
                               const int* n_sel = nullptr, const DwJob* dw = nullptr, bool* dw_taken = nullptr,
                               size_t scratch_avail_floats = 0, int overwrite_from_level = RFX_MAX_LEVELS + 1) {
    ScatterPlan plan;
    if (dw_taken) *dw_taken = false;
    const int64_t n_all = n + n_b;
    const bool staged_ok = scratch && n_all >= SCATTER_MIN_POINTS;
    // levels cut into many segments are binned (one at a time, below); the others share one LDS sweep
    bool binned[RFX_MAX_LEVELS] = {};
    unsigned largest = 0;
    for (int l = 0; l < g.n_levels; ++l) {
        binned[l] = staged_ok && level_is_binned(g, l);
        if (!binned[l]) largest = std::max(largest, g.size[l]);
    }
    const bool f64 = largest < 16u * SCATTER_SEG;           // see grid_scatter_lds_kernel
    const unsigned seg_entries = f64 ? SCATTER_SEG / 2 : SCATTER_SEG;
    int total = 0;
    for (int l = 0; l < g.n_levels; ++l) {
        plan.seg_start[l] = total;
        if (!binned[l]) total += (int)((g.size[l] + seg_entries - 1) / seg_entries);
    }
    for (int l = g.n_levels; l <= RFX_MAX_LEVELS; ++l) plan.seg_start[l] = total;
    if (!staged_ok || total > SCATTER_MAX_SEGMENTS) {
        if (overwrite_from_level < g.n_levels) return RFX_ERR_ARG;          // (never promised on this path: scatter_overwrite_from_level)
        if (n > 0)
            hipLaunchKernelGGL(grid_encode_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, table, x01, n,
                               dfeat, ld, dtable, (float*)nullptr, 0, perm, n_sel);
        if (n_b > 0)
            hipLaunchKernelGGL(grid_encode_backward_kernel, dim3((unsigned)((n_b + 255) / 256)), dim3(256), 0, st, g, table, x01_b,
                               n_b, dfeat_b, ld_b, dtable, (float*)nullptr, 0, (const int*)nullptr, (const int*)nullptr);
        RFX_LAUNCH_CHECK();
        return RFX_OK;
    }
    const ScatterSrc a{dfeat, ld, x01, n, perm, n_sel}, b{dfeat_b, ld_b, x01_b, n_b, nullptr, nullptr};
    plan.n_a = n; plan.n_b = n_b; plan.n_sel = n_sel;
    const size_t scratch_floats = scatter_scratch_floats(n_all, g.n_levels);
    if (total > 0) {
        plan.chunks = 1;
        plan.swept = 0;
        for (int l = 0; l < g.n_levels; ++l) if (!binned[l]) plan.swept |= 1u << l;
        plan.K = (int)((n + n_b + SCATTER_THREADS - 1) / SCATTER_THREADS);
        plan.slots = (int64_t)plan.K * SCATTER_THREADS;
        static int cus_of[64] = {};          // per device
        int dev = 0;
        RFX_HIP_TRY(hipGetDevice(&dev));
        int cus = dev >= 0 && dev < 64 ? cus_of[dev] : 0;
        if (cus <= 0) {
            RFX_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            cus = std::max(cus, 1);
            if (dev >= 0 && dev < 64) cus_of[dev] = cus;
        }
        // the field backward hands over the rows with a gradient only (n_sel, counted on the device): ~0.65 of the batch in a BA
        // iteration -- the plan is made for that many (the costs' RATIO is what it depends on)
        const int64_t n_est = n_b + (n_sel ? (int64_t)(0.65 * (double)n) : n);
        const int n_blocks = sweep_plan(g, binned, seg_entries, f64, std::max<int64_t>(n_est, 1), plan.K, cus, &plan);
        if ((size_t)plan.slots * (2 * g.n_levels + 3) > scratch_floats) return RFX_ERR_WORKSPACE;
        const int nb_stage = (int)((plan.slots + 255) / 256);
        hipLaunchKernelGGL(scatter_stage_kernel, dim3((unsigned)(nb_stage + (dw ? DW_JOB_BLOCKS : 0))), dim3(256), 0, st, a, b, plan,
                           g.n_levels, scratch, nb_stage, dw ? *dw : DwJob{});
        RFX_LAUNCH_CHECK();
        if (dw && dw_taken) *dw_taken = true;
        const size_t lds = (size_t)SCATTER_SEG * 2 * sizeof(float);
        static bool attr_set[64] = {};       // the attribute is per device
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            RFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grid_scatter_lds_kernel<float, SCATTER_SEG>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            RFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grid_scatter_lds_kernel<double, SCATTER_SEG / 2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set[dev] = true;
        }
        // (an XCD-aware block order -- all segments of one (level, chunk) on one XCD's L2 -- was measured: no gain at
        // T = 2^16 / 2^19 and a loss at 2^21, the levels' costs differ too much to be dealt out per XCD)
        if (f64)
            hipLaunchKernelGGL((grid_scatter_lds_kernel<double, SCATTER_SEG / 2>), dim3(n_blocks), dim3(SCATTER_THREADS), lds, st, g,
                               plan, g.n_levels, scratch, dtable);
        else
            hipLaunchKernelGGL((grid_scatter_lds_kernel<float, SCATTER_SEG>), dim3(n_blocks), dim3(SCATTER_THREADS), lds, st, g, plan,
                               g.n_levels, scratch, dtable);
        RFX_LAUNCH_CHECK();
    }
    int bl[RFX_MAX_LEVELS], n_bl = 0;
    for (int l = 0; l < g.n_levels; ++l) if (binned[l]) bl[n_bl++] = l;
    if (n_bl > 0) {       // (after the sweep: the records re-use the staging buffer)
        const int rc = launch_binned_levels(g, bl, n_bl, a, b, dtable, scratch, std::max(scratch_floats, scratch_avail_floats), st, overwrite_from_level);
        if (rc) return rc;
    }
    return RFX_OK;
}

// First level from which the scatter of n_all points into `dtable` WRITES the levels' gradient instead of adding to it -- so the
// caller need not zero that part first: the trailing levels that are hashed, binned and reduced by exactly one block per segment
// (the levels of 2^19-2^21 entries: 8 B per entry of zero-fill and 8 B of read-back saved, 300 MB per iteration at T = 2^21).
// g.n_levels when there is none (small tables, the direct-atomics fallback, a misaligned buffer).  launch_grid_scatter() takes
// the same number and refuses a promise it cannot keep.
int scatter_overwrite_from_level(const rfx_grid_desc& g, int64_t n_all, bool have_scratch, const float* dtable) {
    if (!have_scratch || n_all < SCATTER_MIN_POINTS || ((uintptr_t)dtable & 15)) return g.n_levels;
    unsigned largest = 0;
    for (int l = 0; l < g.n_levels; ++l) if (!level_is_binned(g, l)) largest = std::max(largest, g.size[l]);
    const unsigned seg_entries = largest < 16u * SCATTER_SEG ? SCATTER_SEG / 2 : SCATTER_SEG;
    int total = 0;
    for (int l = 0; l < g.n_levels; ++l) if (!level_is_binned(g, l)) total += (int)((g.size[l] + seg_entries - 1) / seg_entries);
    if (total > SCATTER_MAX_SEGMENTS) return g.n_levels;
    int from = g.n_levels;
    for (int l = g.n_levels - 1; l >= 0; --l) {
        if (!(level_is_binned(g, l) && g.hashed[l] && bin_parts(g, l, n_all) == 1 && (g.size[l] & 1u) == 0 && (g.offset[l] & 1u) == 0)) break;
        from = l;
    }
    return from;
}

__global__ __launch_bounds__(256) void oneblob_forward_kernel(const float* __restrict__ x01, int64_t n, int fp16,
                                                              float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float v[48];
    if (fp16) {          // the packed path the fused kernels use (4 boundary evaluations per coordinate)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            unsigned dw[8];
            oneblob_dim_packed16(x01[p * 3 + d], dw);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const half2v h = __builtin_bit_cast(half2v, dw[i]);
                v[16 * d + 2 * i] = (float)h[0];
                v[16 * d + 2 * i + 1] = (float)h[1];
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) oneblob_dim<16>(x01[p * 3 + d], false, v + 16 * d);
    }
    float4* o = reinterpret_cast<float4*>(out + p * 48);
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// ---------------------------------------------------------------- backward: workspace layout
// per-point rows (fp32).  dX1 = [d_emb32 | d_pos48 | d_cin | d_ex_rgb3], row-major, one row of LD_DX1 (96, 84 used) per point.
// What only the weight-gradient kernel reads -- emb32 (8 float4 pieces; also the forward's stash for the chain), X1' =
// [pos48 | cin | 0 0 0] (13), G = [geo15 | ex_rgb3 | 0 0] (5), dY2 (4) -- is PIECE-MAJOR inside tiles of 64 points (row_piece() in rfx_field_mlp.h): piece q of point p sits at
// tile (p / 64) * pieces * 256 + q * 256 + (p % 64) * 4 floats.  The chain kernel has lane = point, so each of its float4
// stores is then one contiguous 1 KiB write per wave (a row-major row would make it 64 16-byte writes 384 B apart), and
// the weight-gradient kernel's LDS image of a batch is piece-major as well, so its fills are contiguous reads.
constexpr int LD_DX1 = 96;
constexpr int PC_EMB = 8, PC_X1 = 13, PC_G = 5, PC_DY2 = 4;      // float4 pieces per point
constexpr int DW_BLOCKS = 256;       // the weight-gradient kernel is persistent: at most one block per CU

// B-operand (k-step) images of W1..W4 for the weight-gradient kernel's recompute of H1 / H3, 84 slots of 64 lanes
constexpr int DWR_OFF1 = 0, DWR_OFF2 = 41, DWR_OFF3 = 49, DWR_OFF4 = 82, DWR_SLOTS = 84;

struct BwdWs {
    float *emb, *embc, *x1, *g, *dy2, *dx1, *demb_t, *partial, *wcopy;
    int *perm, *sel_hdr, *sel_counts;
};

// Selection (round 2): about 37 % of a mapping batch's sample points -- those behind the surface by more than the truncation
// -- have an exactly zero loss gradient d_raw, and contribute exactly nothing to any gradient.  For launches of at least
// SEL_MIN_POINTS the chain stage first builds a stable partition of the points (non-zero d_raw rows first) in the workspace:
// perm[q] = point staged at row q, sel_hdr[0] = number of non-zero rows.  Every later stage works on the first sel_hdr[0]
// rows (the count stays on the device: no host round trip), reads x01 / d_raw through perm and writes dx01 through it (zeros
// for the rest).  Same sums as without it, up to the order of the additions.
constexpr int64_t SEL_MIN_POINTS = 16384;
constexpr int SEL_PPB = 1024;                  // points per block of the two selection kernels
static inline bool sel_on(int64_t n) { return n >= SEL_MIN_POINTS; }
struct Sel { const int* perm; const int* n_sel; };


__host__ __device__ inline size_t ws_floats_per_point() { return 4 * (2 * PC_EMB + PC_X1 + PC_G + PC_DY2) + LD_DX1 + 1; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static BwdWs carve(void* ws, int64_t n) {
    BwdWs w;
    float* p = reinterpret_cast<float*>(ws);
    const size_t np = align_up((size_t)n, 64);
    w.emb = p; p += np * 4 * PC_EMB;             // the forward's stash: rows in the caller's point order
    w.embc = p; p += np * 4 * PC_EMB;            // the same rows in the chain's (selected) order, for the weight gradients
    w.x1 = p; p += np * 4 * PC_X1;
    w.g = p; p += np * 4 * PC_G;
    w.dy2 = p; p += np * 4 * PC_DY2;
    w.dx1 = p; p += np * LD_DX1;
    w.partial = p; p += (size_t)DW_BLOCKS * DW_TOTAL;
    w.wcopy = p; p += DWR_SLOTS * 64;                  // k-step images of W1..W4 as the chain saw them
    w.perm = reinterpret_cast<int*>(p); p += np;
    w.sel_hdr = reinterpret_cast<int*>(p); p += 16;
    w.sel_counts = reinterpret_cast<int*>(p); p += align_up(np / SEL_PPB + 1, 4);
    w.demb_t = p;                                      // staged d_emb / points of the LDS scatter
    return w;
}

// write 4 consecutive floats of the lane's own row
__device__ __forceinline__ void st4(float* row, int col, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(row + col) = make_float4(a, b, c, d);
}

// D-layout tile pair (t0,t1) -> own-point rows, written to `dst` rows of stride ld at column col0 + row.
// After swap32(t0[r], t1[r]): t0[r] = row krow(r,0) of own point, t1[r] = row krow(r,1).
// Rows come out in groups of four consecutive indices: regs 4q..4q+3 -> rows 8q..8q+3 (t0) / 8q+4..8q+7 (t1).
__device__ __forceinline__ void store_tiles_as_rows(f32x16 t0, f32x16 t1, float* row, int col0, bool relu, bool valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = t0[r], b = t1[r];
        swap32(a, b);
        t0[r] = relu ? fmaxf(a, 0.f) : a;
        t1[r] = relu ? fmaxf(b, 0.f) : b;
    }
    if (!valid) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        st4(row, col0 + 8 * q, t0[4 * q], t0[4 * q + 1], t0[4 * q + 2], t0[4 * q + 3]);
        st4(row, col0 + 8 * q + 4, t1[4 * q], t1[4 * q + 1], t1[4 * q + 2], t1[4 * q + 3]);
    }
}

__device__ __forceinline__ unsigned positive_mask(const f32x16& a, const f32x16& b) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        m |= (a[r] > 0.f ? 1u : 0u) << r;
        m |= (b[r] > 0.f ? 1u : 0u) << (16 + r);
    }
    return m;
}

// ---------------------------------------------------------------- selection: points with a non-zero loss gradient first
__device__ __forceinline__ bool sel_flag(const float* __restrict__ draw4, int64_t p, int64_t n) {
    if (p >= n) return false;
    const float4 d = reinterpret_cast<const float4*>(draw4)[p];
    return d.x != 0.f || d.y != 0.f || d.z != 0.f || d.w != 0.f;
}

__global__ __launch_bounds__(256) void sel_count_kernel(const float* __restrict__ draw4, int64_t n, int* __restrict__ counts) {
    __shared__ int part[4];
    const int64_t p0 = (int64_t)blockIdx.x * SEL_PPB;
    int c = 0;
#pragma unroll
    for (int r = 0; r < SEL_PPB / 256; ++r) c += __popcll(__ballot(sel_flag(draw4, p0 + r * 256 + threadIdx.x, n)));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// stable partition: block b's non-zero points go to rows [sum of earlier blocks' counts, ...), its zero points to the rows
// after all non-zero ones, both in point order (so the result does not depend on scheduling -- nor on how the points are
// cut into blocks).  A block takes `ppb` consecutive points (768 < ppb <= SEL_PPB) = `upb` consecutive count units:
// counts[] holds the non-zero points per unit, either sel_count_kernel's (unit = block, ppb = SEL_PPB) or the loss
// backward's per ray (unit = ray of S points, upb rays per block: no counting launch).
// fin (optional): one more block, which has nothing to do with the selection: it adds up a ray batch's loss partial sums and
// writes the four losses and their coefficients (loss_finalize, rfx_common.h) -- a one-block job that would otherwise be a
// launch of its own between the loss kernel and this one.
struct LossFinJob { const double* partial; int n_partials; int64_t n_rays; int S; float* lc8; };

__global__ __launch_bounds__(256) void sel_scatter_kernel(const float* __restrict__ draw4, int64_t n, const int* __restrict__ counts,
                                                          int n_counts, int upb, int ppb, int* __restrict__ perm, int* __restrict__ hdr,
                                                          int nb_sel, LossFinJob fin) {
    if ((int)blockIdx.x >= nb_sel) {
        __shared__ float lc[8];
        loss_finalize(fin.partial, fin.n_partials, fin.n_rays, fin.S, lc);
        if (threadIdx.x < 8) fin.lc8[threadIdx.x] = lc[threadIdx.x];
        return;
    }
    __shared__ int red[2][4];
    __shared__ int wave_cnt[SEL_PPB / 256][4];
    int before = 0, total = 0;
    const int first_unit = (int)blockIdx.x * upb;
    for (int i = threadIdx.x; i < n_counts; i += 256) {
        const int c = counts[i];
        total += c;
        if (i < first_unit) before += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); total += __shfl_xor(total, o); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = before; red[1][wv] = total; }
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p_end = min(n, p0 + ppb);
    bool f[SEL_PPB / 256];
    unsigned long long m[SEL_PPB / 256];
#pragma unroll
    for (int r = 0; r < SEL_PPB / 256; ++r) {
        f[r] = sel_flag(draw4, p0 + r * 256 + threadIdx.x, p_end);
        m[r] = __ballot(f[r]);
        if (lane == 0) wave_cnt[r][wv] = __popcll(m[r]);
    }
    __syncthreads();
    before = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = total;
    int nz_run = before;                                         // non-zero rows placed so far (this block's share included)
    int64_t seen = p0;                                           // points placed so far (only the block's last lanes can lie past p_end)
#pragma unroll
    for (int r = 0; r < SEL_PPB / 256; ++r) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w == wv) {
                const int64_t p = p0 + r * 256 + threadIdx.x;
                const int rank = __popcll(m[r] & ((1ull << lane) - 1ull));
                if (p < p_end) perm[f[r] ? nz_run + rank : total + (int)(seen - nz_run) + (lane - rank)] = (int)p;
            }
            nz_run += wave_cnt[r][w];
            seen += 64;
        }
    }
}

// ---------------------------------------------------------------- backward kernel A (MFMA chain)
// ROWS: also stage what the weight-gradient kernel reads (X1, G, dY2; it recomputes H1, H3 and their gradients); without them
// only dX1 is written, which is all the input-gradient stages (_scatter with dx01, _dx) need.
// DXFULL: all of dX1 (d_emb, d_pos, d_cin, d_ex_rgb); without it only d_emb (columns 0..31, what the table scatter
// reads) is computed and stored: two of the three dX1 M-tiles and two dX3 M-tiles of matrix work and 256 B/point less.
__device__ __forceinline__ float dwr_image(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                           const float* __restrict__ w4, int slot, int l) {
    const int lo = l & 31, h = l >> 5;
    if (slot < DWR_OFF2) { const int k = 2 * slot + h; return k < N_IN1 ? w1[lo * N_IN1 + k] : 0.f; }           // B[k=in][n=hid] = W1[hid][in]
    if (slot < DWR_OFF3) { const int k = 2 * (slot - DWR_OFF2) + h; return w2[k * N_H + lo]; }                    // B[k=out][n=hid] = W2[out][hid]
    if (slot < DWR_OFF4) { const int k = 2 * (slot - DWR_OFF3) + h; return k < N_IN3 ? w3[lo * N_IN3 + k] : 0.f; }
    const int k = 2 * (slot - DWR_OFF4) + h;
    return k < N_OUT4 ? w4[k * N_H + lo] : 0.f;                                                                    // B[k=c][n=hid] = W4[c][hid]
}

// STASHED: the hash features of these points are already in ws.emb (rfx_field_forward_stash ran on the same points, table and
// workspace): the forward recompute skips its 128 gathers per point.
template <bool POS16, bool ROWS, bool DXFULL, bool STASHED>
__global__ __launch_bounds__(256, BWD_WAVES) void field_backward_kernel(FieldK f, const float* __restrict__ x01, int64_t n_all,
                                                             const float* __restrict__ draw4, BwdWs ws, Sel sel) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    stage_weights(f, wl, ALL_SLOTS);
    if (ROWS && blockIdx.x == 0) {        // the weight-gradient kernel recomputes H1 / H3 from the staged inputs: it needs the weights
        for (int i = threadIdx.x; i < DWR_SLOTS * 64; i += blockDim.x) ws.wcopy[i] = dwr_image(f.w1, f.w2, f.w3, f.w4, i >> 6, i & 63);
    }
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t n = sel.n_sel ? (int64_t)*sel.n_sel : n_all;       // rows to process: the selected points (see Sel)
    for (int64_t base = wave * 64; base < n; base += n_waves * 64) {
        const int64_t p = base + lane;                                // row of the staged arrays
        const bool valid = p < n;
        const int64_t src = sel.perm ? (int64_t)sel.perm[valid ? p : n - 1] : p;      // the point it holds
        float x[3];
        load_point(x01, src, valid ? n_all : 0, x);
        float4 dr = valid ? reinterpret_cast<const float4*>(draw4)[src] : make_float4(0.f, 0.f, 0.f, 0.f);

        Enc e;
        encode_point(f, x, e);
        // ---- stage X1 = [emb | pos, cin] (inside the forward), piece-major: see the workspace layout
        // the stash is in the caller's point order (src); what this kernel stages is in its own row order (p)
        float* embc_row = ws.embc + (p >> 6) * (PC_EMB * ROW_PIECE) + (p & 63) * 4;
        float* stash_row = ws.emb + (src >> 6) * (PC_EMB * ROW_PIECE) + (src & 63) * 4;
        float* x1row = ws.x1 + (p >> 6) * (PC_X1 * ROW_PIECE) + (p & 63) * 4;
        Mlp m;
        mlp_forward_123<STASHED ? 2 : (ROWS ? 1 : 0), ROWS, POS16>(f, x, wl, lane, e, m, STASHED ? stash_row : embc_row, x1row, valid,
                                                                   (STASHED && ROWS) ? embc_row : nullptr);
        if (ROWS && valid) st4(row_piece(x1row, 48), 0, e.cin, dr.x, dr.y, dr.z);      // [cin | dY4]: the weight gradients' inputs
        const unsigned mask1 = positive_mask(m.h1[0], m.h1[1]);
        const unsigned mask3 = positive_mask(m.h3[0], m.h3[1]);
        if (ROWS) {   // G = [geo15 | ex_rgb]: h2 rows 0..15 = (sdf, geo0..14)
            float o[16];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a = m.h2[0][r], b = m.h2[1][r];
                swap32(a, b);
                o[krow(r, 0)] = a; o[krow(r, 1)] = b;
            }
            if (valid) {
                float* grow = ws.g + (p >> 6) * (PC_G * ROW_PIECE) + (p & 63) * 4;
                st4(grow, 0, o[1], o[2], o[3], o[4]); st4(grow, ROW_PIECE, o[5], o[6], o[7], o[8]);
                st4(grow, 2 * ROW_PIECE, o[9], o[10], o[11], o[12]); st4(grow, 3 * ROW_PIECE, o[13], o[14], o[15], e.ex[1]);
                st4(grow, 4 * ROW_PIECE, e.ex[2], e.ex[3], 0.f, 0.f);
            }
        }

        // ---- dH3pre = relu'(h3) . (W4^T dY4)
        f32x16 d3[2] = {zero16(), zero16()};
        {
            float a = dr.x, b = dr.y;
            swap32(a, b);
            float w = wl[(OFFB4 + 0) * 64 + lane];
            d3[0] = mfma32(w, a, d3[0]); d3[1] = mfma32(w, b, d3[1]);
            a = dr.z; b = 0.f;
            swap32(a, b);
            w = wl[(OFFB4 + 1) * 64 + lane];
            d3[0] = mfma32(w, a, d3[0]); d3[1] = mfma32(w, b, d3[1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d3[0][r] = ((mask3 >> r) & 1u) ? d3[0][r] : 0.f;
            d3[1][r] = ((mask3 >> (16 + r)) & 1u) ? d3[1][r] : 0.f;
        }

        // ---- dX3 = W3^T dH3pre, M-tile 1 first (rows 32..63: d_pos[32..47], d_geo[0..14], d_ex_r)
        f32x16 gx1[2] = {zero16(), zero16()};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float w = wl[(OFFB3 + 16 + r) * 64 + lane];
            gx1[0] = mfma32(w, d3[0][r], gx1[0]); gx1[1] = mfma32(w, d3[1][r], gx1[1]);
        }
        // ---- dH1pre = relu'(h1) . (W2^T dY2), dY2 = (d_sdf, d_geo)
        f32x16 d1[2] = {zero16(), zero16()};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float w = wl[(OFFB2 + s) * 64 + lane];
            d1[0] = mfma32(w, gx1[0][8 + s], d1[0]); d1[1] = mfma32(w, gx1[1][8 + s], d1[1]);
        }
        {
            float a = dr.w, b = 0.f;
            swap32(a, b);
            const float w = wl[(OFFB2 + 8) * 64 + lane];
            d1[0] = mfma32(w, a, d1[0]); d1[1] = mfma32(w, b, d1[1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d1[0][r] = ((mask1 >> r) & 1u) ? d1[0][r] : 0.f;
            d1[1][r] = ((mask1 >> (16 + r)) & 1u) ? d1[1][r] : 0.f;
        }
        // own-point rows of gx1: rows q=0..15 -> d_pos[32..47] (colour path), q=16..30 -> d_geo, q=31 -> d_ex_r
        float gq[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = gx1[0][r], b = gx1[1][r];
            swap32(a, b);
            gq[krow(r, 0)] = a; gq[krow(r, 1)] = b;
        }
        if (ROWS && valid) {
            float* yrow = ws.dy2 + (p >> 6) * (PC_DY2 * ROW_PIECE) + (p & 63) * 4;
            st4(yrow, 0, dr.w, gq[16], gq[17], gq[18]); st4(yrow, ROW_PIECE, gq[19], gq[20], gq[21], gq[22]);
            st4(yrow, 2 * ROW_PIECE, gq[23], gq[24], gq[25], gq[26]); st4(yrow, 3 * ROW_PIECE, gq[27], gq[28], gq[29], gq[30]);
        }
        float* dxrow = ws.dx1 + p * LD_DX1;
        // ---- dX1 = W1^T dH1pre: M-tile 0 = d_emb
        {
            f32x16 t[2] = {zero16(), zero16()};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float w = wl[(OFFB1 + r) * 64 + lane];
                t[0] = mfma32(w, d1[0][r], t[0]); t[1] = mfma32(w, d1[1][r], t[1]);
            }
            store_tiles_as_rows(t[0], t[1], dxrow, 0, false, valid);
        }
        // ---- d_pos[0..31] = dX1 M-tile 1 + dX3 M-tile 0
        if (DXFULL) {
            f32x16 t[2] = {zero16(), zero16()};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float w1s = wl[(OFFB1 + 16 + r) * 64 + lane];
                t[0] = mfma32(w1s, d1[0][r], t[0]); t[1] = mfma32(w1s, d1[1][r], t[1]);
                const float w3s = wl[(OFFB3 + r) * 64 + lane];
                t[0] = mfma32(w3s, d3[0][r], t[0]); t[1] = mfma32(w3s, d3[1][r], t[1]);
            }
            store_tiles_as_rows(t[0], t[1], dxrow, 32, false, valid);
        }
        // ---- dX1 M-tile 2: rows 64..79 = d_pos[32..47] (+ colour path gq[0..15]), row 80 = d_cin
        //      dX3 M-tile 2: rows 64,65 = d_ex_g, d_ex_b
        if (DXFULL) {
            f32x16 t[2] = {zero16(), zero16()}, c[2] = {zero16(), zero16()};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float w1s = wl[(OFFB1 + 32 + r) * 64 + lane];
                t[0] = mfma32(w1s, d1[0][r], t[0]); t[1] = mfma32(w1s, d1[1][r], t[1]);
                const float w3s = wl[(OFFB3 + 32 + r) * 64 + lane];
                c[0] = mfma32(w3s, d3[0][r], c[0]); c[1] = mfma32(w3s, d3[1][r], c[1]);
            }
            float tq[32], cq[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = t[0][r], b = t[1][r];
                swap32(a, b);
                tq[krow(r, 0)] = a; tq[krow(r, 1)] = b;
            }
            {
                float a = c[0][0], b = c[1][0];
                swap32(a, b);
                cq[0] = a;                      // row 0 of M-tile 2 = X3 index 64 = ex_g
                a = c[0][1]; b = c[1][1];
                swap32(a, b);
                cq[1] = a;                      // X3 index 65 = ex_b
            }
            if (valid) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    st4(dxrow, 64 + 4 * i, tq[4 * i] + gq[4 * i], tq[4 * i + 1] + gq[4 * i + 1], tq[4 * i + 2] + gq[4 * i + 2],
                        tq[4 * i + 3] + gq[4 * i + 3]);
                // [80] d_cin, [81..83] d_ex_rgb from the colour net (the residual-add part is added by the dx kernel)
                st4(dxrow, 80, tq[16], gq[31], cq[0], cq[1]);
            }
        }
    }
}

// ---------------------------------------------------------------- backward kernel D: dW with H1 / H3 recomputed
// The chain stages only what cannot be recomputed from a point's row: X1 (81), G = [geo15 | ex_rgb] (18), dY2 (16) -- 480 B
// per point where round 1 also staged H1, dH1pre, H3, dH3pre (1 088 B) -- and this kernel rebuilds the hidden activations
// and their gradients on the matrix cores, 32 points at a time, in the orientation the weight gradients need:
//     H1pre [pt x hid] = X1 [pt x 81] . W1^T        A = X1 (lane = point), B = W1 (k-step image in LDS)
//     dH1pre           = (H1pre > 0) . (dY2 [pt x 16] . W2)
//     H3pre  [pt x hid] = X3 [pt x 66] . W3^T,  dH3pre = (H3pre > 0) . (dY4 [pt x 3] . W4)
// A D tile [pt x hid] holds (register r, lane half h) -> point krow(r, h), lane & 31 -> hidden unit: as it stands it is the
// A operand dY^T [hid x pt-pair] of dW = dY^T X and the B operand H [pt-pair x hid] of dW2 / dW4 -- no transposes.  The other
// operand of each product is a lane = feature read of the batch image.
//
// One 32-point batch as this kernel keeps it in LDS, in 16-byte pieces, piece-major like the staged rows:
//   X1 [21 pieces][33] | G [5][33] | dY2 [4][33]          (slot 32 of each piece is padding; X1 = emb 32 | pos 48 | cin, dY4)
// lane = point reads (ds_read_b128, recompute phase) are consecutive pieces; lane = feature reads (ds_read_b32, dW phase)
// step 33 * 4 floats per four lanes, i.e. 4 banks: both conflict-free.  The image is filled by LDS-DMA
// (global_load_lds_dwordx4: the 64 pieces of one wave-instruction land at consecutive LDS addresses, each from its own
// lane's source address -- 32 consecutive pieces of the staged tile, then the pad), 16 instructions per batch and no
// VGPRs, one batch ahead of the arithmetic.
constexpr int DWR_PX = 0, DWR_PG = (PC_EMB + PC_X1) * 33, DWR_PY = DWR_PG + PC_G * 33, DWR_PIECES = DWR_PY + PC_DY2 * 33;
constexpr int DWR_COL_DY4 = 81;        // X1 image columns 81..83: the colour part of d_raw (the chain stages it next to cin)
constexpr int DWR_BUF = 4096;                       // floats per staging buffer (16 wave-instructions x 1 KiB)
constexpr size_t DWR_LDS = (size_t)(DWR_SLOTS * 64 + 4 * 2 * DWR_BUF) * sizeof(float);
static_assert(DWR_PIECES <= 16 * 64 && 2 * DW_TOTAL <= 4 * 2 * DWR_BUF, "dW staging layout");

// float offset of column `col` of point row 0 in a piece-major image section that starts at piece `sec`
__device__ __forceinline__ int dwr_col(int sec, int col) { return (sec + (col >> 2) * 33) * 4 + (col & 3); }

// pieces [half * 512, half * 512 + 512) of a batch image: each of the two waves that share a batch fetches half of it
__device__ __forceinline__ void dwr_fetch(const BwdWs& ws, int64_t p0, int64_t n, int lane, float* buf, int half) {
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)buf + half * 8192);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (half * 8 + i) * 64 + lane;
        int q, row, pcs;
        const float* base;
        if (c < DWR_PG) { q = c / 33; row = c - 33 * q; base = q < PC_EMB ? ws.embc : ws.x1; pcs = q < PC_EMB ? PC_EMB : PC_X1; q -= q < PC_EMB ? 0 : PC_EMB; }
        else if (c < DWR_PY) { const int d = c - DWR_PG; q = d / 33; row = d - 33 * q; base = ws.g; pcs = PC_G; }
        else if (c < DWR_PIECES) { const int d = c - DWR_PY; q = d / 33; row = d - 33 * q; base = ws.dy2; pcs = PC_DY2; }
        else { q = 0; row = 0; base = ws.dy2; pcs = PC_DY2; }        // past the image: any valid piece
        // the pad slot and a ragged last batch re-read a valid point (finite values; masked or dropped below)
        const int64_t pt = std::min<int64_t>(p0 + (row < 32 ? row : 0), n - 1);
        const float* src = base + (pt >> 6) * (pcs * ROW_PIECE) + q * ROW_PIECE + (pt & 63) * 4;
        // as inline asm: behind the builtin hipcc drains the DMA (vmcnt(0)) at the next ds_read of any LDS address, which
        // would serialise the prefetch with the batch it is meant to overlap
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds0 + (unsigned)(i * 1024)) : "memory");
    }
}

// One batch, one side.  SIDE 0: H1pre, dH1pre -> dW1 (acc[0..2]), dW2 (acc[3]).  SIDE 1: H3pre, dH3pre -> dW3, dW4.
template <int SIDE>
__device__ __forceinline__ void dwr_batch(const float* __restrict__ buf, const float* __restrict__ wl, bool vl, int lane, f32x16 (&acc)[4]) {
    const int lo = lane & 31, h = lane >> 5;
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(buf);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x16 hh = zero16(), tt = zero16();
    if (SIDE == 0) {
#pragma unroll
        for (int q = 0; q < 21; ++q) {                              // H1pre = X1 . W1^T, 41 k-steps
            float4 v = b4[DWR_PX + q * 33 + lo]; if (!vl) v = z4;   // columns 4q .. 4q+3 of the lane's point
            hh = mfma32(h ? v.y : v.x, wl[(DWR_OFF1 + 2 * q) * 64 + lane], hh);
            if (2 * q + 1 < 41) hh = mfma32(h ? v.w : v.z, wl[(DWR_OFF1 + 2 * q + 1) * 64 + lane], hh);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                               // T = dY2 . W2, 8
            float4 v = b4[DWR_PY + q * 33 + lo]; if (!vl) v = z4;
            tt = mfma32(h ? v.y : v.x, wl[(DWR_OFF2 + 2 * q) * 64 + lane], tt);
            tt = mfma32(h ? v.w : v.z, wl[(DWR_OFF2 + 2 * q + 1) * 64 + lane], tt);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) {                              // H3pre = [pos48 | geo15 | ex3] . W3^T, 33
            float4 v = b4[DWR_PX + (8 + q) * 33 + lo]; if (!vl) v = z4;   // X1 columns 32..79: the OneBlob part
            hh = mfma32(h ? v.y : v.x, wl[(DWR_OFF3 + 2 * q) * 64 + lane], hh);
            hh = mfma32(h ? v.w : v.z, wl[(DWR_OFF3 + 2 * q + 1) * 64 + lane], hh);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            float4 v = b4[DWR_PG + q * 33 + lo]; if (!vl) v = z4;
            hh = mfma32(h ? v.y : v.x, wl[(DWR_OFF3 + 24 + 2 * q) * 64 + lane], hh);
            if (q < 4) hh = mfma32(h ? v.w : v.z, wl[(DWR_OFF3 + 24 + 2 * q + 1) * 64 + lane], hh);
        }
        float4 v = b4[DWR_PX + (DWR_COL_DY4 / 4) * 33 + lo]; if (!vl) v = z4;      // (cin, dY4): T3 = dY4 . W4, 2
        tt = mfma32(h ? v.z : v.y, wl[(DWR_OFF4 + 0) * 64 + lane], tt);
        tt = mfma32(h ? 0.f : v.w, wl[(DWR_OFF4 + 1) * 64 + lane], tt);
    }
    // masks: dHpre, H.  A point past n has zero rows here, so it adds nothing below
#pragma unroll
    for (int r = 0; r < 16; ++r) { tt[r] = hh[r] > 0.f ? tt[r] : 0.f; hh[r] = fmaxf(hh[r], 0.f); }
    // weight gradients: register r <-> the point pair (krow(r,0), krow(r,1)) of this batch.  Columns past a row's end read
    // other finite parts of the image: they only reach output columns / rows the epilogue drops
    const int oa = dwr_col(DWR_PX, lo), ob = dwr_col(DWR_PX, 32 + lo), oc = dwr_col(DWR_PX, 64 + lo);
    const int oy = dwr_col(DWR_PY, lo & 15), og = dwr_col(DWR_PG, lo >= 16 ? lo - 16 : 0), oh = dwr_col(DWR_PG, 16 + (lo & 3));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* prow = buf + krow(r, h) * 4;
        if (SIDE == 0) {
            const float x1a = prow[oa], x1b = prow[ob], x1c = prow[oc];
            const float dy2 = prow[oy];
            acc[0] = mfma32(tt[r], x1a, acc[0]); acc[1] = mfma32(tt[r], x1b, acc[1]); acc[2] = mfma32(tt[r], x1c, acc[2]);
            acc[3] = mfma32(dy2, hh[r], acc[3]);
        } else {
            const float x1b = prow[ob], x1c = prow[oc];
            const float gb = prow[og], gc = prow[oh];
            const float dy4 = lo < 3 ? prow[dwr_col(DWR_PX, DWR_COL_DY4 + (lo & 3))] : 0.f;
            const float x3b = lo < 16 ? x1c : gb;                   // X3 columns 32..63 = [pos 32..47 | geo 0..14, ex_r]
            acc[0] = mfma32(tt[r], x1b, acc[0]); acc[1] = mfma32(tt[r], x3b, acc[1]); acc[2] = mfma32(tt[r], gc, acc[2]);
            acc[3] = mfma32(dy4, hh[r], acc[3]);
        }
    }
}

// 8 waves: wave w and wave w + 4 (scheduled onto the same SIMD) share the batches of slot w & 3 -- the first takes the
// H1 side, the second the H3 side -- so that each SIMD has two waves' worth of independent MFMA chains and LDS reads.
__global__ __launch_bounds__(512) void field_dw_recompute_kernel(BwdWs ws, int64_t n_all, const int* __restrict__ n_sel,
                                                              float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float dwr_lds[];
    float* wl = dwr_lds;                                     // [DWR_SLOTS][64] weight images
    float* stage = dwr_lds + DWR_SLOTS * 64;                 // [slot][2][DWR_BUF]; after the loop: acc2 [2][DW_TOTAL]
    const int lane = threadIdx.x & 63, lo = lane & 31, h = lane >> 5;
    const int wv = threadIdx.x >> 6, slot = wv & 3, side = wv >> 2;
    float* mine = stage + slot * 2 * DWR_BUF;
    const int64_t n = n_sel ? (int64_t)*n_sel : n_all;         // the selected rows (see Sel)
    const int64_t n_b = (n + 31) / 32, n_s = (int64_t)gridDim.x * 4;
    const int64_t n_it = (n_b + n_s - 1) / n_s;              // every wave of the block runs the same number of rounds (barriers inside)
    int64_t bt = (int64_t)blockIdx.x * 4 + slot;             // batches bt, bt + n_s, ...
    if (bt < n_b) dwr_fetch(ws, bt * 32, n, lane, mine, side);
    for (int i = threadIdx.x; i < DWR_SLOTS * 16; i += 512)
        reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(ws.wcopy)[i];
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = zero16();
    int cur = 0;
    for (int64_t it = 0; it < n_it; ++it, bt += n_s, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my half of this round's batch has landed ...
        __syncthreads();                                      // ... and so has everybody's; the other buffer is free again
        if (bt + n_s < n_b) dwr_fetch(ws, (bt + n_s) * 32, n, lane, mine + (cur ^ 1) * DWR_BUF, side);
        if (bt < n_b) {
            const bool vl = bt * 32 + lo < n;
            if (side == 0) dwr_batch<0>(mine + cur * DWR_BUF, wl, vl, lane, acc);
            else dwr_batch<1>(mine + cur * DWR_BUF, wl, vl, lane, acc);
        }
    }
    // D tile: lane (col = in index lo, half h), reg r -> out row krow(r,h).  The four slots of the block are summed in
    // LDS in a fixed order (over the staging buffers, which every wave is done with), then the block writes ONE partial.
    __syncthreads();
    float* acc2 = stage;
    for (int turn = 0; turn < 2; ++turn) {
        if ((slot >> 1) == turn) {
            float* a = acc2 + (slot & 1) * DW_TOTAL;
            const int in_n = side ? N_IN3 : N_IN1, out_n = side ? N_OUT4 : N_OUT2;
            float* big = a + (side ? N_H * N_IN1 + N_OUT2 * N_H : 0);                       // dW1 | dW3  [N_H][in_n]
            float* small = a + (side ? N_H * N_IN1 + N_OUT2 * N_H + N_H * N_IN3 : N_H * N_IN1);   // dW2 | dW4  [out_n][N_H]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = krow(r, h);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int i1 = 32 * j + lo;
                    if (i1 < in_n) { float& d = big[o * in_n + i1]; d = turn ? d + acc[j][r] : acc[j][r]; }
                }
                if (o < out_n) { float& d = small[o * N_H + lo]; d = turn ? d + acc[3][r] : acc[3][r]; }
            }
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * DW_TOTAL;
    for (int i = threadIdx.x; i < DW_TOTAL; i += 512) out[i] = acc2[i] + acc2[DW_TOTAL + i];
}

// deterministic second stage: block = 64 consecutive outputs x 16 slices of the partial list (a wave reads 256
// contiguous bytes of one partial row); slices are added in a fixed order
template <bool OVERWRITE>
__global__ __launch_bounds__(1024) void field_dw_reduce_kernel(const float* __restrict__ partial, int n_partials,
                                                               float* __restrict__ dw1, float* __restrict__ dw2,
                                                               float* __restrict__ dw3, float* __restrict__ dw4) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < DW_TOTAL) {
#pragma unroll 8
        for (int k = slice; k < n_partials; k += 16) s += partial[(size_t)k * DW_TOTAL + i];
    }
    red[slice][col] = s;
    __syncthreads();
    if (slice != 0 || i >= DW_TOTAL) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][col];
    const int o1 = N_H * N_IN1, o2 = o1 + N_OUT2 * N_H, o3 = o2 + N_H * N_IN3;
    float* d = i < o1 ? (dw1 ? dw1 + i : nullptr) : i < o2 ? (dw2 ? dw2 + (i - o1) : nullptr)
             : i < o3 ? (dw3 ? dw3 + (i - o2) : nullptr) : (dw4 ? dw4 + (i - o3) : nullptr);
    if (d) *d = OVERWRITE ? s : *d + s;
}

// ---------------------------------------------------------------- backward kernel C: dx through OneBlob + GBV
// adds to dx01 (after the hash part wrote it): needs d_pos[48], d_cin, d_ex_rgb(colour net) from the
// dX1 rows and draw4 (residual adds: d ex_rgb += draw.rgb, d tres += draw.sdf).
__global__ __launch_bounds__(256) void field_dx_kernel(FieldK f, const float* __restrict__ x01, int64_t n,
                                                       const float* __restrict__ draw4, const float* __restrict__ dx1,
                                                       float* __restrict__ dx01, Sel sel) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // row of dX1
    if (q >= n || (sel.n_sel && q >= *sel.n_sel)) return;                    // rows past n_sel: no gradient, dx01 stays as it is
    const int64_t p = sel.perm ? (int64_t)sel.perm[q] : q;                   // the point it belongs to
    const float x[3] = {x01[p * 3], x01[p * 3 + 1], x01[p * 3 + 2]};
    const float* row = dx1 + q * LD_DX1;
    float dx[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float g[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {           // rows are 384-byte aligned: 16-byte loads
            const float4 v = reinterpret_cast<const float4*>(row + 32 + 16 * d)[q];
            g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        }
        dx[d] = oneblob_dim_dx<16>(x[d], g);
    }
    const float4 dr = reinterpret_cast<const float4*>(draw4)[p];
    // tsdf path: t = clamp(ex0*c_trunc/trunc); raw.sdf += t; cin = clamp(t,-1,1) (clamp mode) or t
    const float4 ex = lookup4(f.gbv, f.gbv_level, x);
    float t = ex.x * f.c_trunc;
    t = t / f.trunc;
    const float4 tailv = reinterpret_cast<const float4*>(row + 80)[0];      // d_cin, d_ex_rgb
    const float d_cin = tailv.x;
    float d_t;
    if (f.clamp_mode) {
        const bool in_hi = (t >= -f.clamp_hi) && (t <= f.clamp_hi);
        const float tc = fminf(fmaxf(t, -f.clamp_hi), f.clamp_hi);
        const bool in_one = (tc >= -1.0f) && (tc <= 1.0f);
        d_t = in_hi ? (dr.w + (in_one ? d_cin : 0.f)) : 0.f;
    } else {
        const bool in_one = (t >= -1.0f) && (t <= 1.0f);
        d_t = in_one ? (dr.w + d_cin) : 0.f;
    }
    const float gex[4] = {d_t * f.c_trunc / f.trunc, tailv.y + dr.x, tailv.z + dr.y, tailv.w + dr.z};
    lookup_dx<4>(f.gbv, f.gbv_level, x, gex, dx);
    dx01[p * 3] += dx[0]; dx01[p * 3 + 1] += dx[1]; dx01[p * 3 + 2] += dx[2];
}

// ---------------------------------------------------------------- level-partitioned table: features in, feature gradients out
// One scene on several GPUs (mp_slam/sharded.py): rank q keeps the hash levels [level_start[q], level_start[q+1]) and looks them
// up for every sample point of the iteration; the ranks that render the points receive the features as row-major blocks
// [points, 2 k_q] (one block per owning rank) and put them where the forward of a single GPU leaves them: the stash.
struct LevelRowsK { const float* src[RFX_MAX_LEVELS]; int ld[RFX_MAX_LEVELS]; int col[RFX_MAX_LEVELS]; };

// thread = (tile of 64 points, float4 piece q = levels 2q and 2q+1, lane = point): a wave's store is one contiguous 1 KiB run
__global__ __launch_bounds__(256) void stash_put_kernel(LevelRowsK r, int64_t n, float* __restrict__ emb) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tile = gid >> 9;
    const int q = (int)((gid >> 6) & 7), lane = (int)(gid & 63);
    const int64_t p = tile * 64 + lane;
    if (tile * 64 >= n) return;
    float2 a = make_float2(0.f, 0.f), b = a;          // the tail of the last tile is read by the forward's idle lanes: zeros
    if (p < n) {
        a = *reinterpret_cast<const float2*>(r.src[2 * q] + p * (int64_t)r.ld[2 * q] + r.col[2 * q]);
        b = *reinterpret_cast<const float2*>(r.src[2 * q + 1] + p * (int64_t)r.ld[2 * q + 1] + r.col[2 * q + 1]);
    }
    *reinterpret_cast<float4*>(emb + tile * (8 * ROW_PIECE) + q * ROW_PIECE + lane * 4) = make_float4(a.x, a.y, b.x, b.y);
}

// The way back: the chain's d_emb (columns 0..31 of its dX1 rows, in the selection's order) as row-major blocks per owning rank,
// in the caller's point order, zeros for the points without a gradient.  thread = (row, level): a row's 16 threads read its
// 128 bytes contiguously.  One more block adds the loss kernel's partial sums up (total8: what the ranks all-reduce).
struct LevelRowsOutK { float* dst[RFX_MAX_LEVELS]; int ld[RFX_MAX_LEVELS]; int col[RFX_MAX_LEVELS]; };
struct LossSumJob { const double* partial; int n_partials; double* total8; };

__global__ __launch_bounds__(256) void demb_rows_kernel(const float* __restrict__ dx1, const int* __restrict__ perm,
                                                        const int* __restrict__ n_sel, int64_t n, LevelRowsOutK o, int nb_rows,
                                                        LossSumJob job) {
    if ((int)blockIdx.x >= nb_rows) {
        __shared__ double part[32][8];
        const int v = threadIdx.x & 7, q = threadIdx.x >> 3;
        double a = 0.0;
        for (int k = q; k < job.n_partials; k += 32) a += job.partial[k * 8 + v];
        part[q][v] = a;
        __syncthreads();
        if (threadIdx.x < 8) {
            double t = 0.0;
            for (int qq = 0; qq < 32; ++qq) t += part[qq][threadIdx.x];
            job.total8[threadIdx.x] = t;
        }
        return;
    }
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid >> 4;
    const int l = (int)(gid & 15);
    if (i >= n) return;
    const int64_t p = perm ? (int64_t)perm[i] : i;
    const bool live = !n_sel || i < *n_sel;
    const float2 v = live ? *reinterpret_cast<const float2*>(dx1 + i * LD_DX1 + 2 * l) : make_float2(0.f, 0.f);
    *reinterpret_cast<float2*>(o.dst[l] + p * (int64_t)o.ld[l] + o.col[l]) = v;
}

// ---------------------------------------------------------------- host side
// the lookups address an entry of a level as (level base, 32-bit byte offset) -- at32() in rfx_field_device.h: a level of
// 4 GiB or more would wrap silently, so the entry points refuse it
static bool grid_levels_addressable(const rfx_grid_desc& g) {
    for (int l = 0; l < g.n_levels && l < RFX_MAX_LEVELS; ++l)
        if ((uint64_t)g.size[l] * (uint64_t)g.n_feat * sizeof(float) >= (1ull << 32)) return false;
    return true;
}

int make_fieldk(const rfx_field_desc* d, FieldK* k) {
    if (!d || !d->hash_table || !d->gbv || !d->w1 || !d->w2 || !d->w3 || !d->w4) return RFX_ERR_ARG;
    if (d->hash.n_levels != 16 || d->hash.n_feat != 2) return RFX_ERR_UNSUPPORTED;   // decoder input is 32+48+1
    if (!grid_levels_addressable(d->hash)) return RFX_ERR_UNSUPPORTED;
    if ((uint64_t)d->gbv_res * d->gbv_res * d->gbv_res * 16ull >= (1ull << 32)) return RFX_ERR_UNSUPPORTED;   // float4 entries, 32-bit byte offsets
    if (d->gbv_res <= 1 || !(d->trunc > 0.f)) return RFX_ERR_ARG;
    k->hash = d->hash;
    k->table = d->hash_table;
    k->gbv = d->gbv;
    // tcnn dense Grid, n_levels=1, per_level_scale=1: scale = base-1, res = base, size = round_up(res^3, 8)
    k->gbv_level.scale = (float)d->gbv_res - 1.0f;
    k->gbv_level.res = (unsigned)d->gbv_res;
    const uint64_t r3 = (uint64_t)d->gbv_res * d->gbv_res * d->gbv_res;
    if (r3 >= (1ull << 31)) return RFX_ERR_UNSUPPORTED;
    k->gbv_level.size = (unsigned)((r3 + 7) / 8 * 8);
    k->gbv_level.offset = 0;
    k->gbv_level.hashed = 0;
    k->w1 = d->w1; k->w2 = d->w2; k->w3 = d->w3; k->w4 = d->w4;
    k->c_trunc = d->c_trunc; k->trunc = d->trunc; k->clamp_hi = d->clamp_hi;
    k->clamp_mode = d->clamp_mode; k->pos_fp16 = d->pos_fp16;
    k->staged = d->staged;
    if (k->staged && ((uintptr_t)k->staged & 15)) return RFX_ERR_ARG;
    return RFX_OK;
}

__global__ __launch_bounds__(256) void stage_weights_kernel(FieldK f, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ALL_SLOTS * 64) out[i] = staged_weight(f, i >> 6, i & 63);
}

constexpr int FWD_RESIDENT_BLOCKS = 256 * FWD_WAVES;      // blocks of 4 waves the chip holds at FWD_WAVES waves per SIMD (256 CUs)

static inline int wave_grid(int64_t n, int max_blocks) {
    int64_t b = (n + 255) / 256;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, max_blocks));
}

}  // namespace rfx

using namespace rfx;

extern "C" {

size_t rfx_field_staged_floats(void) { return (size_t)ALL_SLOTS * 64; }

int rfx_field_stage_weights(const rfx_field_desc* f, float* staged, rfx_stream stream) {
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (!staged || ((uintptr_t)staged & 15)) return RFX_ERR_ARG;
    k.staged = nullptr;
    hipLaunchKernelGGL(stage_weights_kernel, dim3((ALL_SLOTS * 64 + 255) / 256), dim3(256), 0, as_stream(stream), k, staged);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_grid_encode_forward(const rfx_grid_desc* g, const float* table, const float* x01, int64_t n, float* feat,
                            rfx_stream stream) {
    if (n == 0) return RFX_OK;
    if (!g || !table || !x01 || !feat || n < 0) return RFX_ERR_ARG;
    if (g->n_levels < 1 || g->n_levels > RFX_MAX_LEVELS) return RFX_ERR_ARG;
    if (g->n_feat != 1 && g->n_feat != 2 && g->n_feat != 4) return RFX_ERR_UNSUPPORTED;
    if (!grid_levels_addressable(*g)) return RFX_ERR_UNSUPPORTED;
    if (n == 0) return RFX_OK;
    if (g->n_feat == 2 && (n << 4) < (int64_t)0x7fffffff * 256) {
        const int sh = lp_shift_of(g->n_levels);
        hipLaunchKernelGGL(grid_encode_forward_lp_kernel, dim3((unsigned)(((n << sh) + 255) / 256)), dim3(256), 0, as_stream(stream), *g,
                           table, x01, n, sh, feat);
    } else {
        hipLaunchKernelGGL(grid_encode_forward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), *g,
                           table, x01, n, feat);
    }
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

size_t rfx_grid_encode_backward_workspace_bytes(int64_t n, int n_levels) {
    if (n <= 0 || n_levels <= 0) return 0;
    return scatter_scratch_floats(n, n_levels) * sizeof(float);
}

size_t rfx_grid_encode_backward_workspace_bytes_for(const rfx_grid_desc* g, int64_t n) {
    if (!g || n <= 0 || g->n_levels < 1 || g->n_levels > RFX_MAX_LEVELS) return 0;
    size_t need = scatter_scratch_floats(n, g->n_levels), all = 0;
    if (n >= SCATTER_MIN_POINTS)
        for (int l = 0; l < g->n_levels; ++l)
            if (level_is_binned(*g, l)) all += (binned_level_floats(*g, l, n) + 3) & ~(size_t)3;
    // (+ 128 floats: the callers that carve this region out of a larger workspace round it down to whole 256-byte units, and the
    //  record base is aligned up to 16 bytes: without the margin the last level missed the group by a few floats -- round 6)
    return (std::max(need, all) + 128) * sizeof(float);
}

int rfx_grid_encode_backward(const rfx_grid_desc* g, const float* table, const float* x01, int64_t n,
                             const float* dfeat, float* dtable, float* dx01, void* workspace, size_t workspace_bytes,
                             rfx_stream stream) {
    if (n == 0) return RFX_OK;
    if (!g || !table || !x01 || !dfeat || n < 0) return RFX_ERR_ARG;
    if (g->n_feat != 2 || g->n_levels < 1 || g->n_levels > RFX_MAX_LEVELS) return RFX_ERR_UNSUPPORTED;
    if (!grid_levels_addressable(*g)) return RFX_ERR_UNSUPPORTED;
    if (n == 0 || (!dtable && !dx01)) return RFX_OK;
    if (dtable) {
        if (workspace && (workspace_bytes < rfx_grid_encode_backward_workspace_bytes(n, g->n_levels) || ((uintptr_t)workspace & 7)))
            return RFX_ERR_WORKSPACE;
        int rc = launch_grid_scatter(*g, table, x01, n, dfeat, g->n_levels * 2, dtable, reinterpret_cast<float*>(workspace),
                                     as_stream(stream), nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr,
                                     workspace ? workspace_bytes / sizeof(float) : 0);
        if (rc) return rc;
        RFX_LAUNCH_CHECK();
    }
    if (dx01) {
        launch_encode_dx(*g, table, x01, n, dfeat, g->n_levels * 2, dx01, as_stream(stream));
        RFX_LAUNCH_CHECK();
    }
    return RFX_OK;
}


#ifdef SCATTER_PROF
extern "C" int rfx_debug_scatter_prof(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scatter_prof), sizeof(unsigned long long) * (size_t)std::min(n, 2 * 8192)) == hipSuccess ? 0 : -2;
}
// which: 0 = bin_sort (block = blockIdx.y * gridDim.x + blockIdx.x), 1 = bin_reduce
extern "C" int rfx_debug_bin_prof(int which, unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bin_prof), sizeof(unsigned long long) * (size_t)std::min(n, 2 * 8192),
                               sizeof(unsigned long long) * 2 * 8192 * (size_t)(which ? 1 : 0)) == hipSuccess ? 0 : -2;
}
#endif

int rfx_oneblob_forward(const float* x01, int64_t n, int n_bins, int pos_fp16, float* out, rfx_stream stream) {
    if (n == 0) return RFX_OK;
    if (!x01 || !out || n < 0) return RFX_ERR_ARG;
    if (n_bins != 16) return RFX_ERR_UNSUPPORTED;     // pos.n_bins = 16 in every reference config
    if (n == 0) return RFX_OK;
    hipLaunchKernelGGL(oneblob_forward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x01, n,
                       pos_fp16, out);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

static int launch_forward(const rfx_field_desc* f, const float* x01, int64_t n, float* raw4, float* emb, rfx_stream stream,
                          bool from_stash = false) {
    if (n == 0) return RFX_OK;
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (!x01 || !raw4 || n < 0 || (from_stash && !emb)) return RFX_ERR_ARG;
    dim3 grid(wave_grid(n, 256 * 4));
    // The tail in HALF passes (round 6).  The kernel holds FWD_WAVES waves per SIMD: FWD_SLOTS_RESIDENT wave slots on the chip.
    // With more 64-point passes than slots, the passes behind the last full round used to be a third wave-pass on some SIMDs
    // (2 124 passes: 55.7 us against 43.9 us for 1 888); they go out as 32-point half passes instead, one per wave of the first
    // blocks (one per SIMD while there are at most 4 x CUs of them).  fp32 OneBlob, table lookups (not the stash-fed form) only.
    int64_t n_full = n;
    int n_half = 0;
    static const bool half_off = getenv("RFX_NO_HALF_PASS") != nullptr;          // (A/B runs)
    if (!half_off && !k.pos_fp16 && !from_stash) {
        constexpr int64_t slots = (int64_t)FWD_RESIDENT_BLOCKS * 4;
        const int64_t passes = (n + 63) / 64, rounds = passes / slots;
        const int64_t rest = n - rounds * slots * 64;                              // points behind the last full round
        const int64_t halves = (rest + 31) / 32;
        if (rounds >= 1 && rest > 0 && halves <= FWD_RESIDENT_BLOCKS * 2) {        // one half per SIMD: 4 waves of the first half of the blocks
            n_full = rounds * slots * 64; n_half = (int)halves;
            grid = dim3(FWD_RESIDENT_BLOCKS);
        }
    }
    hipStream_t st = as_stream(stream);
    if (from_stash) {
        if (k.pos_fp16) hipLaunchKernelGGL((field_forward_kernel<true, 2>), grid, dim3(256), 0, st, k, x01, n, raw4, emb, n_full, n_half);
        else hipLaunchKernelGGL((field_forward_kernel<false, 2>), grid, dim3(256), 0, st, k, x01, n, raw4, emb, n_full, n_half);
    } else if (emb) {
        if (k.pos_fp16) hipLaunchKernelGGL((field_forward_kernel<true, 1>), grid, dim3(256), 0, st, k, x01, n, raw4, emb, n_full, n_half);
        else hipLaunchKernelGGL((field_forward_kernel<false, 1>), grid, dim3(256), 0, st, k, x01, n, raw4, emb, n_full, n_half);
    } else {
        if (k.pos_fp16) hipLaunchKernelGGL((field_forward_kernel<true, 0>), grid, dim3(256), 0, st, k, x01, n, raw4, emb, n_full, n_half);
        else hipLaunchKernelGGL((field_forward_kernel<false, 0>), grid, dim3(256), 0, st, k, x01, n, raw4, emb, n_full, n_half);
    }
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_field_forward(const rfx_field_desc* f, const float* x01, int64_t n, float* raw4, rfx_stream stream) {
    return launch_forward(f, x01, n, raw4, nullptr, stream);
}

int rfx_field_query_sdf(const rfx_field_desc* f, const float* x01, int64_t n, float* sdf, rfx_stream stream) {
    if (n == 0) return RFX_OK;
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (!x01 || !sdf || n < 0) return RFX_ERR_ARG;
    if (n == 0) return RFX_OK;
    k.clamp_mode = 0;   // query_sdf_res clamps to +-1 regardless of self.clamp (scene_rep.py:233)
    if (k.pos_fp16) hipLaunchKernelGGL((field_query_kernel<0, true>), dim3(wave_grid(n, 256 * 4)), dim3(256), 0, as_stream(stream), k, x01, n, sdf);
    else hipLaunchKernelGGL((field_query_kernel<0, false>), dim3(wave_grid(n, 256 * 4)), dim3(256), 0, as_stream(stream), k, x01, n, sdf);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_field_query_color(const rfx_field_desc* f, const float* x01, int64_t n, float* rgb3, rfx_stream stream) {
    if (n == 0) return RFX_OK;
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (!x01 || !rgb3 || n < 0) return RFX_ERR_ARG;
    if (n == 0) return RFX_OK;
    if (k.pos_fp16) hipLaunchKernelGGL((field_query_kernel<1, true>), dim3(wave_grid(n, 256 * 4)), dim3(256), 0, as_stream(stream), k, x01, n, rgb3);
    else hipLaunchKernelGGL((field_query_kernel<1, false>), dim3(wave_grid(n, 256 * 4)), dim3(256), 0, as_stream(stream), k, x01, n, rgb3);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

size_t rfx_field_backward_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    const size_t np = align_up((size_t)n, 64);
    return (np * ws_floats_per_point() + (size_t)DW_BLOCKS * DW_TOTAL + DWR_SLOTS * 64 + 16 + align_up(np / SEL_PPB + 1, 4) + scatter_scratch_floats(n, RFX_MAX_LEVELS)) * sizeof(float);
}

// ---- the four stages of the Q1 backward as separate entry points (rfx_field_backward chains them)
int rfx_field_forward_stash(const rfx_field_desc* f, const float* x01, int64_t n, float* raw4, void* workspace, size_t workspace_bytes,
                            rfx_stream stream) {
    if (n == 0) return RFX_OK;
    if (n < 0) return RFX_ERR_ARG;
    if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
    if ((uintptr_t)workspace & 15) return RFX_ERR_ARG;
    return launch_forward(f, x01, n, raw4, carve(workspace, n).emb, stream);
}

static int level_rows_ok(const rfx_level_rows* r) {
    if (!r) return RFX_ERR_ARG;
    for (int l = 0; l < RFX_MAX_LEVELS; ++l)
        if (!r->rows[l] || r->ld[l] < 2 || (r->ld[l] & 1) || r->col[l] < 0 || (r->col[l] & 1) || r->col[l] + 2 > r->ld[l] ||
            ((uintptr_t)r->rows[l] & 7))
            return RFX_ERR_ARG;
    return RFX_OK;
}

int rfx_field_stash_put(const rfx_level_rows* rows, int64_t n, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (n == 0) return RFX_OK;
    if (n < 0) return RFX_ERR_ARG;
    int rc = level_rows_ok(rows);
    if (rc) return rc;
    if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
    if ((uintptr_t)workspace & 15) return RFX_ERR_ARG;
    LevelRowsK k;
    for (int l = 0; l < RFX_MAX_LEVELS; ++l) { k.src[l] = rows->rows[l]; k.ld[l] = rows->ld[l]; k.col[l] = rows->col[l]; }
    const int64_t threads = (n + 63) / 64 * 512;
    hipLaunchKernelGGL(stash_put_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), k, n,
                       carve(workspace, n).emb);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_field_forward_stashed(const rfx_field_desc* f, const float* x01, int64_t n, float* raw4, void* workspace,
                              size_t workspace_bytes, rfx_stream stream) {
    if (n == 0) return RFX_OK;
    if (n < 0) return RFX_ERR_ARG;
    if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
    if ((uintptr_t)workspace & 15) return RFX_ERR_ARG;
    return launch_forward(f, x01, n, raw4, carve(workspace, n).emb, stream, true);
}

int rfx_field_backward_demb_rows(int64_t n, const rfx_level_rows* rows, const double* loss_partials, int n_loss_partials,
                                 double* loss_total8, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    const bool sum_job = loss_partials && loss_total8 && n_loss_partials > 0;
    if (n == 0 && !sum_job) return RFX_OK;
    if (n < 0) return RFX_ERR_ARG;
    LevelRowsOutK o = {};
    BwdWs ws = {};
    if (n > 0) {
        int rc = level_rows_ok(rows);
        if (rc) return rc;
        if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
        if ((uintptr_t)workspace & 15) return RFX_ERR_ARG;
        for (int l = 0; l < RFX_MAX_LEVELS; ++l) { o.dst[l] = const_cast<float*>(rows->rows[l]); o.ld[l] = rows->ld[l]; o.col[l] = rows->col[l]; }
        ws = carve(workspace, n);
    }
    const int nb_rows = (int)((n * 16 + 255) / 256);
    hipLaunchKernelGGL(demb_rows_kernel, dim3((unsigned)(nb_rows + (sum_job ? 1 : 0))), dim3(256), 0, as_stream(stream), ws.dx1,
                       sel_on(n) ? ws.perm : nullptr, sel_on(n) ? ws.sel_hdr : nullptr, n, o, nb_rows,
                       LossSumJob{loss_partials, sum_job ? n_loss_partials : 0, loss_total8});
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_grid_encode_backward_merged(const rfx_grid_desc* g, const float* table, const float* x01_a, int64_t n_a,
                                    const float* dfeat_a, const float* x01_b, int64_t n_b, const float* dfeat_b, float* dtable,
                                    void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return grid_encode_backward_merged_from(g, table, x01_a, n_a, dfeat_a, x01_b, n_b, dfeat_b, dtable, workspace, workspace_bytes, stream,
                                            RFX_MAX_LEVELS + 1);
}

}  // extern "C"

namespace rfx {
int grid_encode_backward_merged_from(const rfx_grid_desc* g, const float* table, const float* x01_a, int64_t n_a,
                                     const float* dfeat_a, const float* x01_b, int64_t n_b, const float* dfeat_b, float* dtable,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream, int overwrite_from_level) {
    if (n_a < 0 || n_b < 0) return RFX_ERR_ARG;
    if (n_a + n_b == 0) return RFX_OK;
    if (!g || !table || !dtable || (n_a > 0 && (!x01_a || !dfeat_a)) || (n_b > 0 && (!x01_b || !dfeat_b))) return RFX_ERR_ARG;
    if (g->n_feat != 2 || g->n_levels < 1 || g->n_levels > RFX_MAX_LEVELS) return RFX_ERR_UNSUPPORTED;
    if (!grid_levels_addressable(*g)) return RFX_ERR_UNSUPPORTED;
    if (workspace && (workspace_bytes < rfx_grid_encode_backward_workspace_bytes(n_a + n_b, g->n_levels) || ((uintptr_t)workspace & 7)))
        return RFX_ERR_WORKSPACE;
    if (n_a == 0) {           // the first source carries the selection in the fused callers: keep it the non-empty one
        return launch_grid_scatter(*g, table, x01_b, n_b, dfeat_b, g->n_levels * 2, dtable, reinterpret_cast<float*>(workspace),
                                   as_stream(stream), nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr,
                                   workspace ? workspace_bytes / sizeof(float) : 0, overwrite_from_level);
    }
    return launch_grid_scatter(*g, table, x01_a, n_a, dfeat_a, g->n_levels * 2, dtable, reinterpret_cast<float*>(workspace),
                               as_stream(stream), x01_b, dfeat_b, g->n_levels * 2, n_b, nullptr, nullptr, nullptr, nullptr,
                               workspace ? workspace_bytes / sizeof(float) : 0, overwrite_from_level);
}
}  // namespace rfx

extern "C" {

static int launch_backward_chain(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, void* workspace,
                                 size_t workspace_bytes, rfx_stream stream, bool rows, bool dxfull, bool stashed = false,
                                 const int* ray_counts = nullptr, int ray_S = 0, const LossFinJob* fin = nullptr, int* finalized = nullptr) {
    if (finalized) *finalized = 0;
    if (n == 0) return RFX_OK;
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (!x01 || !draw4 || n < 0) return RFX_ERR_ARG;
    if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
    if ((uintptr_t)workspace & 15) return RFX_ERR_ARG;
    BwdWs ws = carve(workspace, n);
    const size_t lds = (size_t)ALL_SLOTS * 64 * sizeof(float);
    using Kern = void (*)(FieldK, const float*, int64_t, const float*, BwdWs, Sel);
    // [pos_fp16 | stashed][variant]: 0 = rows + full dX1 (_chain), 1 = full dX1 only (_chain_inputs), 2 = rows + d_emb (_chain_weights)
    static const Kern kern[4][3] = {
        {field_backward_kernel<false, true, true, false>, field_backward_kernel<false, false, true, false>, field_backward_kernel<false, true, false, false>},
        {field_backward_kernel<true, true, true, false>, field_backward_kernel<true, false, true, false>, field_backward_kernel<true, true, false, false>},
        {field_backward_kernel<false, true, true, true>, field_backward_kernel<false, false, true, true>, field_backward_kernel<false, true, false, true>},
        {field_backward_kernel<true, true, true, true>, field_backward_kernel<true, false, true, true>, field_backward_kernel<true, true, false, true>}};
    static bool attr_set[64] = {};   // per device; raising the dynamic-LDS limit is idempotent, so a race is benign
    int dev = 0;
    RFX_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        for (int a = 0; a < 4; ++a)
            for (int v = 0; v < 3; ++v)
                RFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern[a][v]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int variant = rows ? (dxfull ? 0 : 2) : 1;
    Sel sel{nullptr, nullptr};
    if (sel_on(n)) {            // points with a non-zero loss gradient first; the later stages read the same perm / count
        if (ray_counts && ray_S > 0 && ray_S < 256 && n % ray_S == 0) {
            // the non-zero rows were counted per ray by the kernel that wrote draw4 (the loss backward): blocks of whole rays
            const int upb = SEL_PPB / ray_S, ppb = upb * ray_S;
            const int n_rays = (int)(n / ray_S);
            const int nb_sel = (n_rays + upb - 1) / upb;
            hipLaunchKernelGGL(sel_scatter_kernel, dim3(nb_sel + (fin ? 1 : 0)), dim3(256), 0, as_stream(stream), draw4, n, ray_counts, n_rays,
                               upb, ppb, ws.perm, ws.sel_hdr, nb_sel, fin ? *fin : LossFinJob{});
            if (fin && finalized) *finalized = 1;
        } else {
            const int nb = (int)((n + SEL_PPB - 1) / SEL_PPB);
            hipLaunchKernelGGL(sel_count_kernel, dim3(nb), dim3(256), 0, as_stream(stream), draw4, n, ws.sel_counts);
            hipLaunchKernelGGL(sel_scatter_kernel, dim3(nb), dim3(256), 0, as_stream(stream), draw4, n, ws.sel_counts, nb, 1, SEL_PPB, ws.perm,
                               ws.sel_hdr, nb, LossFinJob{});
        }
        RFX_LAUNCH_CHECK();
        sel = Sel{ws.perm, ws.sel_hdr};
    }
    hipLaunchKernelGGL(kern[(k.pos_fp16 ? 1 : 0) + (stashed ? 2 : 0)][variant], dim3(wave_grid(n, 256 * 2)), dim3(256), lds, as_stream(stream), k, x01, n, draw4, ws, sel);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_field_backward_chain(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                             void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream, true, true);
}

int rfx_field_backward_chain_inputs(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                    void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream, false, true);
}

int rfx_field_backward_chain_weights(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream, true, false);
}

int rfx_field_backward_chain_stashed(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream, true, true, true);
}

int rfx_field_backward_chain_inputs_stashed(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                            void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream, false, true, true);
}

int rfx_field_backward_chain_weights_stashed(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                             void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream, true, false, true);
}

static int launch_backward_weights(int64_t n, const float* draw4, float* dw1, float* dw2, float* dw3, float* dw4,
                                   void* workspace, size_t workspace_bytes, rfx_stream stream, bool overwrite,
                                   DwJob* defer = nullptr) {
    if (defer) *defer = DwJob{};
    if (!dw1 && !dw2 && !dw3 && !dw4) return RFX_OK;
    hipStream_t st = as_stream(stream);
    if (n == 0) {           // nothing to add; the overwriting form still has to leave zeros
        if (overwrite) {
            if (dw1) RFX_HIP_TRY(hipMemsetAsync(dw1, 0, sizeof(float) * N_H * N_IN1, st));
            if (dw2) RFX_HIP_TRY(hipMemsetAsync(dw2, 0, sizeof(float) * N_OUT2 * N_H, st));
            if (dw3) RFX_HIP_TRY(hipMemsetAsync(dw3, 0, sizeof(float) * N_H * N_IN3, st));
            if (dw4) RFX_HIP_TRY(hipMemsetAsync(dw4, 0, sizeof(float) * N_OUT4 * N_H, st));
        }
        return RFX_OK;
    }
    if (!draw4 || n < 0) return RFX_ERR_ARG;
    if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
    BwdWs ws = carve(workspace, n);
    // persistent: one block per CU (150 KB of LDS each), whole 32-point batches dealt round-robin to the wave pairs
    static bool attr_set[64] = {};       // the attribute is per device; benign if raced
    int dev = 0;
    RFX_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        RFX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(field_dw_recompute_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)DWR_LDS));
        attr_set[dev] = true;
    }
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(DW_BLOCKS, ((n + 31) / 32 + 3) / 4));
    hipLaunchKernelGGL(field_dw_recompute_kernel, dim3(blocks), dim3(512), DWR_LDS, st, ws, n, sel_on(n) ? ws.sel_hdr : nullptr, ws.partial);
    RFX_LAUNCH_CHECK();
    if (defer && overwrite) {             // the caller issues the second stage (beside the scatter's staging pass)
        *defer = DwJob{ws.partial, blocks, dw1, dw2, dw3, dw4};
        return RFX_OK;
    }
    if (overwrite)
        hipLaunchKernelGGL(field_dw_reduce_kernel<true>, dim3((DW_TOTAL + 63) / 64), dim3(1024), 0, st, ws.partial, blocks, dw1, dw2, dw3, dw4);
    else
        hipLaunchKernelGGL(field_dw_reduce_kernel<false>, dim3((DW_TOTAL + 63) / 64), dim3(1024), 0, st, ws.partial, blocks, dw1, dw2, dw3, dw4);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_field_backward_weights(int64_t n, const float* draw4, float* dw1, float* dw2, float* dw3, float* dw4,
                               void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_weights(n, draw4, dw1, dw2, dw3, dw4, workspace, workspace_bytes, stream, false);
}

int rfx_field_backward_scatter(const rfx_field_desc* f, const float* x01, int64_t n, float* d_hash, float* dx01,
                               void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (n == 0 || (!d_hash && !dx01)) return RFX_OK;
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (!x01 || n < 0) return RFX_ERR_ARG;
    if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
    BwdWs ws = carve(workspace, n);
    if (d_hash) {
        // the scatter's region is the last of the workspace: whatever the caller hands over beyond the minimum lets more binned
        // levels (T >= 2^19) share a group of launches (rfx_grid_encode_backward_workspace_bytes_for says how much all of them need)
        const size_t avail = (workspace_bytes - (size_t)((char*)ws.demb_t - (char*)workspace)) / sizeof(float);
        rc = launch_grid_scatter(k.hash, k.table, x01, n, ws.dx1, LD_DX1, d_hash, ws.demb_t, as_stream(stream), nullptr, nullptr, 0, 0,
                                 sel_on(n) ? ws.perm : nullptr, sel_on(n) ? ws.sel_hdr : nullptr, nullptr, nullptr, avail);
        if (rc) return rc;
        RFX_LAUNCH_CHECK();
    }
    if (dx01) {
        launch_encode_dx(k.hash, k.table, x01, n, ws.dx1, LD_DX1, dx01, as_stream(stream), sel_on(n) ? ws.perm : nullptr,
                         sel_on(n) ? ws.sel_hdr : nullptr);
        RFX_LAUNCH_CHECK();
    }
    return RFX_OK;
}

int rfx_field_backward_scatter_merged(const rfx_field_desc* f, const float* x01, int64_t n, const float* extra_x01,
                                      const float* extra_dfeat, int64_t extra_n, float* d_hash, void* workspace,
                                      size_t workspace_bytes, void* scatter_ws, size_t scatter_bytes, rfx_stream stream) {
    if ((n == 0 && extra_n == 0) || !d_hash) return RFX_OK;
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (n < 0 || extra_n < 0 || (n > 0 && !x01) || (extra_n > 0 && (!extra_x01 || !extra_dfeat))) return RFX_ERR_ARG;
    if (n > 0 && (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n))) return RFX_ERR_WORKSPACE;
    if (scatter_ws && (scatter_bytes < rfx_grid_encode_backward_workspace_bytes(n + extra_n, k.hash.n_levels) || ((uintptr_t)scatter_ws & 7)))
        return RFX_ERR_WORKSPACE;
    BwdWs ws{};
    if (n > 0) ws = carve(workspace, n);
    rc = launch_grid_scatter(k.hash, k.table, x01, n, ws.dx1, LD_DX1, d_hash, reinterpret_cast<float*>(scatter_ws), as_stream(stream),
                             extra_x01, extra_dfeat, k.hash.n_levels * 2, extra_n, sel_on(n) ? ws.perm : nullptr,
                             sel_on(n) ? ws.sel_hdr : nullptr, nullptr, nullptr, scatter_ws ? scatter_bytes / sizeof(float) : 0);
    if (rc) return rc;
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_field_backward_dx(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, float* dx01,
                          void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (n == 0 || !dx01) return RFX_OK;
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (!x01 || !draw4 || n < 0) return RFX_ERR_ARG;
    if (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n)) return RFX_ERR_WORKSPACE;
    BwdWs ws = carve(workspace, n);
    hipLaunchKernelGGL(field_dx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), k, x01, n, draw4,
                       ws.dx1, dx01, sel_on(n) ? Sel{ws.perm, ws.sel_hdr} : Sel{nullptr, nullptr});
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_field_backward(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                       float* d_hash, float* dw1, float* dw2, float* dw3, float* dw4, float* dx01,
                       void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (n == 0) return RFX_OK;
    int rc = rfx_field_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream);
    if (rc) return rc;
    rc = rfx_field_backward_weights(n, draw4, dw1, dw2, dw3, dw4, workspace, workspace_bytes, stream);
    if (rc) return rc;
    rc = rfx_field_backward_scatter(f, x01, n, d_hash, dx01, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return rfx_field_backward_dx(f, x01, n, draw4, dx01, workspace, workspace_bytes, stream);
}

}  // extern "C"

namespace rfx {
// field_backward_weights_overwrite followed by rfx_field_backward_scatter_merged, with the weight gradients' second stage
// moved into the scatter's staging launch when there is one (same sums in the same order: bit-identical results)
int field_backward_weights_scatter(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, float* dw1, float* dw2,
                                   float* dw3, float* dw4, const float* extra_x01, const float* extra_dfeat, int64_t extra_n,
                                   float* d_hash, void* workspace, size_t workspace_bytes, void* scatter_ws, size_t scatter_bytes,
                                   rfx_stream stream, int overwrite_from_level, void* weights_done_event) {
    // weights_done_event (optional, a hipEvent_t): recorded between the weight-gradient launch and the scatter's launches
    // (rfx_ba_desc.stage_events[RFX_BA_EV_WEIGHTS])
    if (!d_hash || (n == 0 && extra_n == 0)) {
        if (d_hash && overwrite_from_level <= RFX_MAX_LEVELS) return RFX_ERR_ARG;
        int rc = launch_backward_weights(n, draw4, dw1, dw2, dw3, dw4, workspace, workspace_bytes, stream, true);
        if (rc) return rc;
        if (weights_done_event) RFX_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(weights_done_event), as_stream(stream)));
        return rfx_field_backward_scatter_merged(f, x01, n, extra_x01, extra_dfeat, extra_n, d_hash, workspace, workspace_bytes, scatter_ws,
                                                 scatter_bytes, stream);
    }
    FieldK k;
    int rc = make_fieldk(f, &k);
    if (rc) return rc;
    if (n < 0 || extra_n < 0 || (n > 0 && !x01) || (extra_n > 0 && (!extra_x01 || !extra_dfeat))) return RFX_ERR_ARG;
    if (n > 0 && (!workspace || workspace_bytes < rfx_field_backward_workspace_bytes(n))) return RFX_ERR_WORKSPACE;
    if (scatter_ws && (scatter_bytes < rfx_grid_encode_backward_workspace_bytes(n + extra_n, k.hash.n_levels) || ((uintptr_t)scatter_ws & 7)))
        return RFX_ERR_WORKSPACE;
    DwJob job;
    rc = launch_backward_weights(n, draw4, dw1, dw2, dw3, dw4, workspace, workspace_bytes, stream, true, &job);
    if (rc) return rc;
    if (weights_done_event) RFX_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(weights_done_event), as_stream(stream)));
    BwdWs ws{};
    if (n > 0) ws = carve(workspace, n);
    bool taken = false;
    rc = launch_grid_scatter(k.hash, k.table, x01, n, ws.dx1, LD_DX1, d_hash, reinterpret_cast<float*>(scatter_ws), as_stream(stream),
                             extra_x01, extra_dfeat, k.hash.n_levels * 2, extra_n, sel_on(n) ? ws.perm : nullptr,
                             sel_on(n) ? ws.sel_hdr : nullptr, job.partial ? &job : nullptr, &taken,
                             scatter_ws ? scatter_bytes / sizeof(float) : 0, overwrite_from_level);
    if (rc) return rc;
    if (job.partial && !taken) {          // no staging launch on this path: the stand-alone second stage
        hipLaunchKernelGGL(field_dw_reduce_kernel<true>, dim3((DW_TOTAL + 63) / 64), dim3(1024), 0, as_stream(stream), job.partial,
                           job.n_partials, job.dw1, job.dw2, job.dw3, job.dw4);
    }
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// the three stashed chain stages for a caller that already knows how many rows of every ray have a gradient
// (ray_counts[n / S], from the kernel that wrote draw4): variant 0 = _chain_stashed, 1 = _chain_inputs_stashed,
// 2 = _chain_weights_stashed.  Same result: the selection is a stable partition, whoever counts.
// loss_partials (optional): a ray batch's loss partial sums still waiting for their finalize (composite_loss_grad): taken along by
// the selection's launch when there is one (*finalized = 1), else left to the caller.
int field_backward_chain_stashed_counted(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, void* workspace,
                                         size_t workspace_bytes, int variant, const int* ray_counts, int S, const double* loss_partials,
                                         int n_loss_partials, float* lc8, int* finalized, rfx_stream stream) {
    LossFinJob fin{loss_partials, n_loss_partials, S > 0 ? n / S : 0, S, lc8};
    return launch_backward_chain(f, x01, n, draw4, workspace, workspace_bytes, stream, variant != 1, variant != 2, true, ray_counts, S,
                                 loss_partials && lc8 && n_loss_partials > 0 ? &fin : nullptr, finalized);
}

// rfx_field_backward_weights that OVERWRITES dw1..dw4 (no zero-fill needed before it); used by rfx_ba_forward_backward
int field_backward_weights_overwrite(int64_t n, const float* draw4, float* dw1, float* dw2, float* dw3, float* dw4,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return launch_backward_weights(n, draw4, dw1, dw2, dw3, dw4, workspace, workspace_bytes, stream, true);
}
}  // namespace rfx
