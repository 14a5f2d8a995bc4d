// rfx_tsdf.hip -- moving TSDF volume (MV) and global explicit volume (GBV) kernels for gfx950.
//
// Replaces the PyCUDA kernels of the reference (model/Volume.py:127-611, mp_slam/mapper.py:36-185).
// The reference launches one thread per voxel of the whole box; for the integrate kernel ~97 %
// of those threads exit at the frustum / depth tests.  Here V1 is a frustum-culled walk in three launches
// ("queue form", documented at its section below):
//   mv_frame_kernel  : one pass over the H*W frame -> packed colour, {depth, 1/lambda} image, the {F, G}
//                      classification image of the fast path (8 B/pixel each, L2-resident) and coarse max-depth tiles.
//   mv_rows_kernel   : one thread per (x,y) voxel row of the frustum's footprint.  Each row is a line in camera
//                      space, so frustum /\ row is one z-interval; its 64-voxel chunks go to a work queue.
//   mv_chunks_kernel : resident waves take chunks off the queue (segments of it dealt to the XCDs, a wave's items gathered
//                      into registers 32 at a time), lanes along z (the contiguous axis): every volume access is a
//                      coalesced 256-B run.  Free-space voxels (85 % of a frame's updates) are classified by two compares
//                      against the {F, G} image and only move their weight; the truncation band is compacted through LDS
//                      and evaluated after the item loop, the block's records dealt over its waves in rounds of 64, by the
//                      reference's full expression tree.
// mv_integrate_kernel (one wave per tile of rows, round 1) remains as the fallback for volumes whose every voxel decodes
// literally (dy*dz >= 2^24).
// Per-voxel arithmetic is evaluated exactly as the reference kernel text does (same operation
// order, fmaf where nvcc -fmad=true contracts, IEEE div/sqrt), so results are bit-identical to
// oracle/tsdf_oracle.c; the cull and the fast path only remove work whose outcome is provably the reference's.
#include "rfx_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#ifndef MV_TX
#define MV_TX 2
#endif
#ifndef MV_TY
#define MV_TY 4
#endif
#ifndef MV_U
#define MV_U 2          // chunks in flight per wave (queue form: 2 -> 8 waves per SIMD; measured 72 us per call vs 81 at 4)
#endif
#ifndef MV_TD
#define MV_TD 16      // pixels per side of a coarse max-depth tile
#endif
#ifndef MV_ROWS_THREADS
#define MV_ROWS_THREADS 1024     // mv_rows_kernel block: one queue atomic per block that has work
#endif
#ifndef MV_XCD_DEAL
#define MV_XCD_DEAL 1  // mv_chunks_kernel: segments of the queue dealt to the XCDs (see the kernel)
#endif
// (Round 5 built a far-end walk of every row through the max-depth tiles: -14 % items at office0, mv_chunks -1.3 us, mv_rows
// +8 us, not kept.  Round 6 tried it as a run-time choice for the 1.5e9-voxel apartment volume, where it cut the call from 0.36 to
// 0.25 ms -- and found it WRONG there: 1.75 M voxels differed from the unclipped walk after 8 frames (surfaces left at their
// initial value; caught by tests/test_sharded_configs_gpu.py, whose slabs and whole volume chose differently).  Removed.)
#ifndef MV_XCD_SEGS
#define MV_XCD_SEGS 32 // segments per XCD (rounded to a power-of-two segment length); 8 / 32 measured: 40.0 / 38.9 us
#endif
#ifndef MV_DBG_SKIP
#define MV_DBG_SKIP 0
#endif
#if (defined(MV_DBG_SKIP) && MV_DBG_SKIP) || (defined(MV_DBG_NEAR) && MV_DBG_NEAR)
#ifndef RFX_DEV_BUILD
#error "MV_DBG_SKIP / MV_DBG_NEAR produce WRONG results on purpose (timing builds): they need -DRFX_DEV_BUILD as well"
#endif
#endif
#ifndef MV_DBG_NEAR
#define MV_DBG_NEAR 0  // timing builds only: 1 = the exact path without its stores, 2 = without its gathers
#endif
#define MV_HDR 4       // workspace header words before the coarse tiles: [0] queue count, [1] risky-queue count

namespace rfx {

thread_local int g_last_hip_error = 0;

#ifdef MV_STATS      // dev builds only (tools/v1_probe.py): work counters of the V1 kernel
__device__ unsigned long long g_mv_stats[16];   // 0 waves, 1 waves with work, 2 chunk items, 3 lanes in z-range, 4 lanes projected
                                               // in-image, 5 lanes updated, 6 colour-band lanes, 7 risky-row chunk items
#define MV_STAT(i, v) do { unsigned long long _v = (v); if (_v) atomicAdd(&g_mv_stats[i], _v); } while (0)
#else
#define MV_STAT(i, v) do { } while (0)
#endif

#ifdef MV_TIMING     // dev builds only (tools/v1_wave_times.py): start / end of every wave of mv_chunks_kernel, 100 MHz ticks
__device__ unsigned long long g_mv_times[2 * 8192 * 2];
#endif

struct MvParams {
    float K[9];
    float c2w[16];
    float origin[3];      // (float)(int)origin  -- the reference truncates (Volume.py:230-232)
    float voxel;
    int   dx, dy, dz;
    int   H, W;
    float trunc, obs_weight;
    int   weight_clamp, reintegrate;
    float old_bnd[6];
    int   risky_rows;     // rows with y < risky_rows or y >= dy - risky_rows decode literally
    int   literal_all;    // 1: every voxel decodes literally (conditions for the split not met)
    float ratio_eps;      // bound on |cam_norm/(lambda*cam_z) - 1| from pixel rounding
    int   dimg_colmajor;  // 1: the packed {depth,1/lambda} image is stored [x][y] (see prepass)
    // up to three disjoint windows of (x,y) tiles: [0] the frustum footprint (host AABB), [1],[2] the
    // first / last tile rows in y, which hold the literally-decoded boundary rows and are always walked
    int   win_x0[3], win_y0[3], win_wx[3], win_wy[3];
    int   alias_margin;   // voxels within this index distance of an x-slab boundary may decode to another cell (decode_split)
    float edge_eps;       // fast path: an approximately projected coordinate closer than this to a rounding boundary (k + 0.5)
                          // may round to another pixel than the reference's: such lanes take the exact projection
};

// ---------------------------------------------------------------------------- per-voxel math
struct RowConst {   // everything that is constant along z for exactly-decoded rows
    float px, py;         // world x,y of the row
    float ax, ay, az;     // fma(R0c, tx, R1c*ty) for c = 0,1,2
};

__device__ __forceinline__ RowConst make_row(const MvParams& P, float vx, float vy) {
    RowConst r;
    r.px = madd(vx, P.voxel, P.origin[0]);
    r.py = madd(vy, P.voxel, P.origin[1]);
    float tx = r.px - P.c2w[3];
    float ty = r.py - P.c2w[7];
    r.ax = madd(P.c2w[0], tx, P.c2w[4] * ty);
    r.ay = madd(P.c2w[1], tx, P.c2w[5] * ty);
    r.az = madd(P.c2w[2], tx, P.c2w[6] * ty);
    return r;
}

// literal fp32 index decode of Volume.py:224-226 (32-bit int arithmetic like the reference)
__device__ __forceinline__ void decode_literal(int idx, int dy, int dz, float& vx, float& vy, float& vz) {
    vx = floorf(((float)idx) / ((float)(dy * dz)));
    vy = floorf(((float)(idx - ((int)vx) * dy * dz)) / ((float)dz));
    vz = (float)(idx - ((int)vx) * dy * dz - ((int)vy) * dz);
}

template <int U>
struct Lanes {
    float cx[U], cy[U], cz[U];
    int   pix[U];
    int   dix[U];
    bool  ok[U];
    int64_t idx[U];
};

// stage A: camera point + pixel; stage B: depth/lambda fetch + sdf; stage C: volume RMW.
template <int U>
__device__ __forceinline__ void integrate_lanes(const MvParams& P, Lanes<U>& L,
                                                const float2* __restrict__ dimg,
                                                const float* __restrict__ cpk,
                                                float* __restrict__ tsdf, float* __restrict__ weight,
                                                float* __restrict__ color) {
    float2 dl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        bool ok = L.ok[u] && (L.cz[u] > 0.0f);
        float czs = ok ? L.cz[u] : 1.0f;
        int px = f2i_rn(madd(P.K[0], (L.cx[u] / czs), P.K[2]));
        int py = f2i_rn(madd(P.K[4], (L.cy[u] / czs), P.K[5]));
        ok = ok && px >= 0 && px < P.W && py >= 0 && py < P.H;
        L.ok[u] = ok;
        MV_STAT(4, __popcll(__ballot(ok)) * ((threadIdx.x & 63) == 0));
        L.pix[u] = ok ? py * P.W + px : 0;
        L.dix[u] = ok ? (P.dimg_colmajor ? px * P.H + py : py * P.W + px) : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) dl[u] = L.ok[u] ? dimg[L.dix[u]] : make_float2(0.0f, 0.0f);

    float sdf[U], cur[U], wold[U], oc[U], ncl[U];
    bool band[U];
#if defined(MV_DEBUG_STAGE) && MV_DEBUG_STAGE == 2
    {
        float acc = 0.f;
        for (int u = 0; u < U; ++u) acc += dl[u].x + dl[u].y;
        if (acc == 12345.678f) tsdf[0] = acc;
        return;
    }
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float d = dl[u].x;
        float norm = sqrtf(madd(L.cz[u], L.cz[u], madd(L.cx[u], L.cx[u], L.cy[u] * L.cy[u])));
        sdf[u] = -madd(dl[u].y, norm, -d);
        bool upd = L.ok[u] && (d > 0.0f) && (sdf[u] >= -P.trunc);
        L.ok[u] = upd;
        band[u] = upd && (sdf[u] <= P.trunc);
        MV_STAT(5, __popcll(__ballot(upd)) * ((threadIdx.x & 63) == 0));
        MV_STAT(6, __popcll(__ballot(band[u])) * ((threadIdx.x & 63) == 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        cur[u] = 0.f; wold[u] = 0.f; oc[u] = 0.f; ncl[u] = 0.f;
        if (L.ok[u]) { cur[u] = tsdf[L.idx[u]]; wold[u] = weight[L.idx[u]]; }
        if (band[u]) { oc[u] = color[L.idx[u]]; ncl[u] = cpk[L.pix[u]]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!L.ok[u]) continue;
        float dist = fminf(1.0f, sdf[u] / P.trunc);
        float w_old = wold[u];
        float w_new = w_old + P.obs_weight;
        float new_t = madd(cur[u], w_old, P.obs_weight * dist) / w_new;
        float new_w = w_new;
        if (P.weight_clamp == 1) {
            new_w = fminf(w_new, 128.0f);
            if (new_w > 40.0f) new_w = 40.0f;
        }
        float new_c = 0.0f;
        if (band[u]) {
            float nc = ncl[u];
            float nb = floorf(nc / 65536.0f);
            float ng = floorf((nc - nb * 65536.0f) / 256.0f);
            float nr = nc - nb * 65536.0f - ng * 256.0f;
            float ob = floorf(oc[u] / 65536.0f);
            float og = floorf((oc[u] - ob * 65536.0f) / 256.0f);
            float orr = oc[u] - ob * 65536.0f - og * 256.0f;
            nb = fminf(roundf(madd(ob, w_old, P.obs_weight * nb) / w_new), 255.0f);
            ng = fminf(roundf(madd(og, w_old, P.obs_weight * ng) / w_new), 255.0f);
            nr = fminf(roundf(madd(orr, w_old, P.obs_weight * nr) / w_new), 255.0f);
            new_c = nb * 65536.0f + ng * 256.0f + nr;
        }
        const bool reset = (P.obs_weight == -1.0f) && (w_old <= 1.0f) && (P.reintegrate == 1);
        if (reset) { new_t = 1.0f; new_w = 0.0f; new_c = 0.0f; }
        tsdf[L.idx[u]] = new_t;
        weight[L.idx[u]] = new_w;
        if (band[u] || reset) color[L.idx[u]] = new_c;
    }
}

__device__ __forceinline__ bool outside_old(const MvParams& P, float px, float py, float pz) {
    return px < P.old_bnd[0] || px >= P.old_bnd[1] || py < P.old_bnd[2] || py >= P.old_bnd[3] ||
           pz < P.old_bnd[4] || pz >= P.old_bnd[5];
}

// ---------------------------------------------------------------------------- V1 main kernel
template <int TX, int TY, int U>
__global__ __launch_bounds__(256) void mv_integrate_kernel(MvParams P, const float2* __restrict__ dimg,
                                                           const unsigned* __restrict__ dmax_bits,
                                                           const float* __restrict__ cpk,
                                                           float* __restrict__ tsdf,
                                                           float* __restrict__ weight,
                                                           float* __restrict__ color) {
    constexpr int ROWS = TX * TY;
    static_assert(ROWS <= 64, "tile must fit one wave");
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int t = wave, wi = 0;
    for (; wi < 3; ++wi) {
        const int cnt = P.win_wx[wi] * P.win_wy[wi];
        if (t < cnt) break;
        t -= cnt;
    }
    if (wi == 3) return;
    const int x0 = (P.win_x0[wi] + t / P.win_wy[wi]) * TX;
    const int y0 = (P.win_y0[wi] + t % P.win_wy[wi]) * TY;

    // ---- per-row conservative z interval (lane r < ROWS owns row r of the tile)
    //      phase A (row lanes): frustum clip -> [lo, hi] and the coarse-tile box of the row's projection
    //      phase B (all lanes, 64 / ROWS per row): max depth over that box -- the tile reads are dependent L2 round
    //               trips, up to ~30 per row when one lane walks them alone
    //      phase C (row lanes): far clip from that depth, integer interval
    int z0 = 0, z1 = 0;
    float lo = 0.0f, hi = -1.0f, Az = 0.0f, Bz = 0.0f, eps = 0.0f;
    bool row_live = false, want_tiles = false;
    int tu0 = 0, tu1 = -1, tv0 = 0, tv1 = -1;
    const int tw = (P.W + MV_TD - 1) / MV_TD, th = (P.H + MV_TD - 1) / MV_TD;
    {
        const int rx = x0 + lane / TY, ry = y0 + lane % TY;
        if (lane < ROWS && rx < P.dx && ry < P.dy) {
            const bool risky = P.literal_all || ry < P.risky_rows || ry >= P.dy - P.risky_rows;
            if (risky) {
                z0 = 0; z1 = P.dz;      // literal-decode rows are not culled (they may alias)
            } else {
                const float wx = P.origin[0] + (float)rx * P.voxel - P.c2w[3];
                const float wy = P.origin[1] + (float)ry * P.voxel - P.c2w[7];
                const float wz = P.origin[2] - P.c2w[11];
                const float Ax = P.c2w[0] * wx + P.c2w[4] * wy + P.c2w[8] * wz;
                const float Ay = P.c2w[1] * wx + P.c2w[5] * wy + P.c2w[9] * wz;
                Az = P.c2w[2] * wx + P.c2w[6] * wy + P.c2w[10] * wz;
                const float Bx = P.c2w[8] * P.voxel, By = P.c2w[9] * P.voxel;
                Bz = P.c2w[10] * P.voxel;
                const float fx = P.K[0], fy = P.K[4], cx = P.K[2], cy = P.K[5];
                const float m = 0.05f;   // pixel margin
                const float mag = fabsf(Ax) + fabsf(Ay) + fabsf(Az) +
                                  (float)P.dz * (fabsf(Bx) + fabsf(By) + fabsf(Bz));
                eps = 1e-4f * fmaxf(fx, fy) * mag + 1e-4f;
                lo = 0.0f; hi = (float)(P.dz - 1);
                bool empty = false;
                auto clip = [&](float a, float b) {   // keep z with a + b z >= -eps
                    a += eps;
                    const float q = -a * __builtin_amdgcn_rcpf(b);   // ~1 ulp: absorbed by eps and the +-1 voxel margin
                    if (b > 0.0f)      lo = fmaxf(lo, q);
                    else if (b < 0.0f) hi = fminf(hi, q);
                    else if (a < 0.0f) empty = true;
                };
                clip(Az, Bz);                                                            // cam_z > 0
                clip(fx * Ax + (cx + 0.5f + m) * Az, fx * Bx + (cx + 0.5f + m) * Bz);    // px >= 0
                clip(((float)P.W - 0.5f + m - cx) * Az - fx * Ax,
                     ((float)P.W - 0.5f + m - cx) * Bz - fx * Bx);                       // px < W
                clip(fy * Ay + (cy + 0.5f + m) * Az, fy * By + (cy + 0.5f + m) * Bz);    // py >= 0
                clip(((float)P.H - 0.5f + m - cy) * Az - fy * Ay,
                     ((float)P.H - 0.5f + m - cy) * Bz - fy * By);                       // py < H
                if (!empty && lo <= hi) {
                    // The row is a straight line in the image: its pixels lie in the bounding box of the two
                    // end points.  No voxel of the row can update if it is deeper than the deepest pixel of the
                    // coarse tiles that box touches (+ trunc): the far end is clipped again below, per row.
                    const float za = fmaxf(Az + lo * Bz, 1e-6f), zb = fmaxf(Az + hi * Bz, 1e-6f);
                    const float ra = __builtin_amdgcn_rcpf(za), rb = __builtin_amdgcn_rcpf(zb);
                    const float u0 = fx * (Ax + lo * Bx) * ra + cx, v0 = fy * (Ay + lo * By) * ra + cy;
                    const float u1 = fx * (Ax + hi * Bx) * rb + cx, v1 = fy * (Ay + hi * By) * rb + cy;
                    tu0 = max(0, (int)floorf((fminf(u0, u1) - 2.0f) / MV_TD)); tu1 = min(tw - 1, (int)floorf((fmaxf(u0, u1) + 2.0f) / MV_TD));
                    tv0 = max(0, (int)floorf((fminf(v0, v1) - 2.0f) / MV_TD)); tv1 = min(th - 1, (int)floorf((fmaxf(v0, v1) + 2.0f) / MV_TD));
                    row_live = true;
                    want_tiles = tu1 >= tu0 && tv1 >= tv0;
                }
            }
        }
    }
    float tm = 0.0f;
    {
        constexpr int HELPERS = 64 / ROWS;                   // lanes that share one row's tile walk
        static_assert(64 % ROWS == 0 && (HELPERS & (HELPERS - 1)) == 0, "tile shape");
        const int row = lane / HELPERS, sub = lane % HELPERS;
        const int bu0 = __shfl(tu0, row), bu1 = __shfl(tu1, row), bv0 = __shfl(tv0, row), bv1 = __shfl(tv1, row);
        const bool need = __shfl((int)want_tiles, row) != 0;
        float part = 0.0f;
        if (need) {
            const int ntx = bu1 - bu0 + 1, nt = ntx * (bv1 - bv0 + 1);
            for (int t = sub; t < nt; t += HELPERS)
                part = fmaxf(part, __uint_as_float(dmax_bits[MV_HDR + (bv0 + t / ntx) * tw + bu0 + t % ntx]));
        }
#pragma unroll
        for (int o = 1; o < HELPERS; o <<= 1) part = fmaxf(part, __shfl_xor(part, o));
        tm = __shfl(part, (lane % ROWS) * HELPERS);         // row lane r takes the result of its helper group
    }
    if (row_live) {
        bool empty = false;
        if (!(tm > 0.0f)) empty = true;
        else {
            // the same clip as in phase A (keep z with a + b z >= -eps), on the far side: cam_z <= deepest pixel + trunc
            const float a = (tm + P.trunc) / (1.0f - P.ratio_eps) * 1.0001f + 1e-3f - Az + eps;
            const float bb = -Bz;
            const float q = -a * __builtin_amdgcn_rcpf(bb);
            if (bb > 0.0f)      lo = fmaxf(lo, q);
            else if (bb < 0.0f) hi = fminf(hi, q);
            else if (a < 0.0f)  empty = true;
        }
        if (!empty && lo <= hi) {
            z0 = max(0, (int)floorf(lo) - 1);
            z1 = min(P.dz, (int)ceilf(hi) + 2);
            if (z1 < z0) z1 = z0;
        }
    }

    // ---- exactly-decoded rows: flatten (row, 64-voxel chunk) work items over the tile and keep U of
    //      them in flight, so one wave overlaps the dependent load chains of several rows.
#if defined(MV_DEBUG_STAGE) && MV_DEBUG_STAGE == 1
    if (z1 == 12345678) tsdf[0] = 0.f;   // keep the cull alive
    return;
#endif
    const int ry_l = y0 + lane % TY;
    const bool risky_l = lane < ROWS && (P.literal_all || ry_l < P.risky_rows || ry_l >= P.dy - P.risky_rows);
    const int nchunk = (!risky_l && z1 > z0) ? (z1 - z0 + 63) >> 6 : 0;
    int incl = nchunk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int excl = incl - nchunk;
    const int total = __shfl(incl, 63);
    if (lane == 0) { MV_STAT(0, 1); MV_STAT(1, total > 0 ? 1 : 0); MV_STAT(2, total); }
    for (int base = 0; base < total; base += U) {
        Lanes<U> L;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = base + u;
            int r = __popcll(__ballot(incl <= q));          // row whose chunk range contains q
            const bool live = q < total;
            r = live ? r : 0;
            const int rz0 = __shfl(z0, r), rz1 = __shfl(z1, r), rex = __shfl(excl, r);
            const int rx = x0 + r / TY, ry = y0 + r % TY;
            const RowConst rc = make_row(P, (float)rx, (float)ry);
            const int z = rz0 + ((q - rex) << 6) + lane;
            bool ok = live && z < rz1;
            const float pz = madd((float)z, P.voxel, P.origin[2]);
            if (P.reintegrate == 1) ok = ok && !outside_old(P, rc.px, rc.py, pz);
            const float tz = pz - P.c2w[11];
            L.cx[u] = madd(P.c2w[8], tz, rc.ax);
            L.cy[u] = madd(P.c2w[9], tz, rc.ay);
            L.cz[u] = madd(P.c2w[10], tz, rc.az);
            L.ok[u] = ok;
            L.idx[u] = ((int64_t)rx * P.dy + ry) * P.dz + z;
            MV_STAT(3, __popcll(__ballot(ok)) * (lane == 0));
        }
        integrate_lanes<U>(P, L, dimg, cpk, tsdf, weight, color);
    }
    // ---- rows that decode literally (reference fp32 index decode may alias): whole row, per-voxel decode
    unsigned long long risky_rows_mask = __ballot(risky_l && z1 > z0);
    while (risky_rows_mask) {
        const int r = __ffsll((long long)risky_rows_mask) - 1;
        risky_rows_mask &= risky_rows_mask - 1;
        const int rx = x0 + r / TY, ry = y0 + r % TY;
        const int64_t row_base = ((int64_t)rx * P.dy + ry) * P.dz;
        for (int zb = 0; zb < P.dz; zb += 64) {
            if (lane == 0) MV_STAT(7, 1);
            Lanes<1> L;
            const int z = zb + lane;
            bool ok = z < P.dz;
            const int idx = (int)(row_base + z);
            float vx, vy, vz;
            decode_literal(ok ? idx : 0, P.dy, P.dz, vx, vy, vz);
            const RowConst rc = make_row(P, vx, vy);
            const float pz = madd(vz, P.voxel, P.origin[2]);
            if (P.reintegrate == 1) ok = ok && !outside_old(P, rc.px, rc.py, pz);
            const float tz = pz - P.c2w[11];
            L.cx[0] = madd(P.c2w[8], tz, rc.ax);
            L.cy[0] = madd(P.c2w[9], tz, rc.ay);
            L.cz[0] = madd(P.c2w[10], tz, rc.az);
            L.ok[0] = ok;
            L.idx[0] = idx;
            integrate_lanes<1>(P, L, dimg, cpk, tsdf, weight, color);
        }
    }
}


// ============================================================================ V1, queue form (round 3)
// Three launches per frame:
//   mv_frame_kernel  : one block per 32x32-pixel tile.  Packs the colour image (when the caller hands over rgb), the
//                      {depth, 1/lambda} image the exact path reads, the {F, G} CLASSIFICATION image the fast path reads
//                      (two thresholds on the squared camera-space norm of a voxel, see below) and the tile's max depth.
//   mv_rows_kernel   : one THREAD per (x,y) row of the frustum's footprint window.  Frustum clip -> z interval, far clip
//                      from a max-depth pyramid over the coarse tiles (LDS; at most four reads per row), then the row's
//                      64-voxel chunks (aligned to 256 B of the volume arrays) are appended to a work queue as 32-byte
//                      items that carry everything that is constant along the row (camera-space row constants, byte
//                      offset of the row): one returning atomic per 1 024-thread block.
//   mv_chunks_kernel : a grid of resident waves pulls U items per trip (scalar loads), lanes along z.
// Fast path of the chunk kernel (obs_weight > 0, no re-integration: every mapping frame), all of it branch-free: every
// conditional memory operation is a raw buffer load / store whose per-lane offset is pushed out of range for the lanes
// that must not take part (the range check drops them: no exec-mask regions, no scalar bookkeeping).  Per voxel:
//   * camera point exactly as the reference forms it (5 operations);
//   * APPROXIMATE projection (one v_rcp_f32 instead of two IEEE divisions).  The rounded pixel is the reference's unless
//     the approximate coordinate lies within `edge_eps` of a rounding boundary; such lanes (~0.5 %) take the exact path;
//   * one 8-byte gather of {F, G} at that pixel: with n2 = |cam|^2 (the very argument of the reference's square root)
//         n2 < F  <=>  sdf > trunc for sure   (free space: dist = 1, no colour; F = 0 where that can never be decided)
//         n2 > G  <=>  sdf < -trunc for sure, or the pixel has no depth (the reference returns)
//     anything else is "near" and takes the exact path.  A free-space voxel that still holds tsdf = 1 keeps it
//     ((1*w + obs*1) / (w + obs) is x / x), so only its weight moves -- and not even that at the clamp.
//   * near lanes (truncation band, boundary pixels, voxels that were near a surface before) are compacted through LDS and
//     evaluated 64 at a time by the reference's full expression tree (exact projection included).
// Everything else (re-integration, de-integration, rows next to x-slab boundaries whose fp32 index decode may alias)
// goes through the generic body: unconditional loads, exact projection, the same LDS compaction.
// Bit-identical to oracle/tsdf_oracle.c: the fast path only ever decides what the exact expressions would decide, with
// margins three orders above the rounding errors involved (derivation next to mv_frame_kernel).
struct __attribute__((aligned(16))) MvItem {
    unsigned xy;          // rx | ry << 15
    unsigned cw;          // chunk | lo << 16 | hi << 22: lanes [lo, hi) of the chunk lie inside the row's z interval
    float ax, ay, az;     // make_row(): everything of the camera point that is constant along the row
    unsigned off_lo, off_hi;   // byte offset of the row's first voxel in a volume array (global index * 4)
    unsigned pad;
};
static_assert(sizeof(MvItem) == 32, "queue item");

constexpr int MV_ROWS_LDS_TILES = 4096;         // coarse tiles kept in LDS by mv_rows_kernel (16 KB); more -> global reads
constexpr int MV_Q_MAX_DIM = 32767;
constexpr int MV_PYR_FLOATS = MV_ROWS_LDS_TILES + 64;    // every level above the base together (a 1-D strip of tiles: ~n)

__device__ __forceinline__ bool alias_zone(const MvParams& P, int ry, int z) {
    const int64_t off = (int64_t)ry * P.dz + z;                      // index distance from the slab's first voxel
    return off < P.alias_margin || (int64_t)P.dy * P.dz - off <= P.alias_margin;
}

// ---- frame kernel.  Thresholds of the classification image, for a pixel with depth d > 0 and rl = 1/lambda (the value
// the reference multiplies the norm by):  sdf = rn(d - rl * norm), norm = sqrt_rn(n2).
//   free  (sdf > trunc)  is implied by  rl * sqrt(n2) * (1 + 2^-23) <= d - trunc - delta, i.e. by
//         n2 < F := ((d - trunc)(1 - 1e-5) - 1e-6)^2 / rl^2 * (1 - 1e-5)         (F = 0 when the bracket is not positive)
//   untouched (sdf < -trunc) is implied by  n2 > G := ((d + trunc)(1 + 1e-5) + 1e-6)^2 / rl^2 * (1 + 1e-5)
// The relative margins (1e-5) are ~100 fp32 ulps; the roundings in forming F, G, n2 and sdf are a handful of ulps each.
// d <= 0 or NaN: F = 0, G = -1 (n2 > -1 always: untouched, the reference returns at `depth <= 0`).
// Queue form (nimg != nullptr): writes fg and the exact path's image nimg = {depth, 1/lambda, packed colour, 0} (round 5: the
// three values the exact path samples at a pixel in ONE 16-byte gather; the packed colour is formed from rgb or copied from
// the caller's color_packed).  Tile form (nimg == nullptr): writes dimg = {depth, 1/lambda} and, from rgb, cpk.
__global__ __launch_bounds__(256) void mv_frame_kernel(const float* __restrict__ depth, const float* __restrict__ rgb,
                                                       const float* __restrict__ color_packed,
                                                       float2* __restrict__ dimg, float2* __restrict__ fg, float4* __restrict__ nimg,
                                                       float* __restrict__ cpk, unsigned* __restrict__ hdr, int H, int W,
                                                       float fx, float fy, float cx, float cy, float trunc, int colmajor) {
    // one block per MV_TD x MV_TD pixel tile, one pixel per thread (the divisions and square roots of a pixel are one
    // dependent chain: four pixels per thread, as in round 2, took four chains' time on a chip that had only 300 blocks)
    static_assert(MV_TD == 16, "tile indexing below assumes 16x16 pixels, 256 threads");
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = 0u; hdr[1] = 0u; }
    const int tw = (W + MV_TD - 1) / MV_TD;
    const int ty = blockIdx.x / tw, tx = blockIdx.x - ty * tw;
    const int px = tx * MV_TD + (threadIdx.x & 15), py = ty * MV_TD + (threadIdx.x >> 4);
    const bool in = px < W && py < H;
    const int i = in ? py * W + px : 0;
    const float d = depth[i];
    float r = 0.f, g = 0.f, b = 0.f;
    if (rgb) { r = rgb[(int64_t)i * 3]; g = rgb[(int64_t)i * 3 + 1]; b = rgb[(int64_t)i * 3 + 2]; }
    float m = 0.0f;
    if (in) {
        const float vx = (((float)px) - cx) / fx;
        const float vy = (((float)py) - cy) / fy;
        const float lambda = sqrtf(madd(vx, vx, vy * vy) + 1.0f);
        const float rl = 1.0f / lambda;
        float F = 0.0f, G = -1.0f;
        if (d > 0.0f) {
            const float inv = 1.0f / rl;
            const float tf = (d - trunc) * (1.0f - 1e-5f) - 1e-6f;
            if (tf > 0.0f) { const float a = tf * inv; F = a * a * (1.0f - 1e-5f); }
            const float tg = (d + trunc) * (1.0f + 1e-5f) + 1e-6f;
            const float bb = tg * inv;
            G = bb * bb * (1.0f + 1e-5f);
            m = d;             // NaN never gets here
        }
        // A voxel row (world z) projects to a near-straight pixel walk.  Storing the images with the walk direction
        // contiguous turns the per-lane gather of a 64-voxel chunk from 64 cache lines into ~12: column-major when the
        // walk is mostly along image y.
        const int di = colmajor ? px * H + py : i;
        const float packed = rgb ? floorf(b * 65536.0f + g * 256.0f + r)      // np.floor(B*65536 + G*256 + R), left to right
                                 : (color_packed ? color_packed[i] : 0.0f);
        if (nimg) {
            fg[di] = make_float2(F, G);
            nimg[di] = make_float4(d, rl, packed, 0.0f);
        } else {
            dimg[di] = make_float2(d, rl);
            if (rgb) cpk[i] = packed;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        hdr[MV_HDR + blockIdx.x] = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
}

// ---- rows kernel
__global__ __launch_bounds__(MV_ROWS_THREADS) void mv_rows_kernel(MvParams P, const unsigned* __restrict__ dmax_bits,
                                                       unsigned* __restrict__ q_counts, MvItem* __restrict__ queue,
                                                       unsigned q_cap, MvItem* __restrict__ queue_risky, unsigned q_cap_risky,
                                                       int blocks_main, int64_t slab_skip) {
    // max-depth pyramid over the coarse tiles: level 0 = the tiles, level l+1 = 2x2 maxima of level l.  A row's
    // projection is a straight segment; its bounding box is looked up at the level where it spans at most 2x2 cells.
    __shared__ float pyr[MV_ROWS_LDS_TILES + MV_PYR_FLOATS];
    __shared__ int lvl_off[16], lvl_w[16], lvl_h[16];
    const int tw = (P.W + MV_TD - 1) / MV_TD, th = (P.H + MV_TD - 1) / MV_TD;
    const int n_tiles = tw * th;
    const bool in_lds = n_tiles <= MV_ROWS_LDS_TILES;
    int n_lvl = 1;
    if (in_lds) {
        for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) pyr[i] = __uint_as_float(dmax_bits[MV_HDR + i]);
        if (threadIdx.x == 0) {
            int o = 0, w = tw, h = th, l = 0;
            for (;; ++l) {
                lvl_off[l] = o; lvl_w[l] = w; lvl_h[l] = h;
                if (w == 1 && h == 1) break;
                o += w * h; w = (w + 1) >> 1; h = (h + 1) >> 1;
            }
            lvl_off[15] = l + 1;
        }
        __syncthreads();
        n_lvl = lvl_off[15];
        for (int l = 1; l < n_lvl; ++l) {
            const int w = lvl_w[l], h = lvl_h[l], pw = lvl_w[l - 1], ph = lvl_h[l - 1];
            const float* __restrict__ src = pyr + lvl_off[l - 1];
            for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
                const int y = i / w, x = i - y * w;
                const int x1 = min(2 * x + 1, pw - 1), y1 = min(2 * y + 1, ph - 1);
                pyr[lvl_off[l] + i] = fmaxf(fmaxf(src[2 * y * pw + 2 * x], src[2 * y * pw + x1]),
                                            fmaxf(src[y1 * pw + 2 * x], src[y1 * pw + x1]));
            }
            __syncthreads();
        }
    }
    const int lane = threadIdx.x & 63;
    // a block works on ONE window: [0, blocks_main) the frustum footprint, the rest the risky boundary rows (windows 1, 2
    // back to back); so `risky` is block-uniform and each block appends to one queue with one atomic
    const bool risky = (int)blockIdx.x >= blocks_main;
    int wi = 0, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (risky) {
        t -= blocks_main * (int)blockDim.x;
        wi = 1;
        if (t >= P.win_wx[1] * P.win_wy[1]) { t -= P.win_wx[1] * P.win_wy[1]; wi = 2; }
    }
    int z0 = 0, z1 = 0, rx = 0, ry = 0;
    bool row = false;
    if (t < P.win_wx[wi] * P.win_wy[wi]) {
        rx = P.win_x0[wi] + t / P.win_wy[wi];
        ry = P.win_y0[wi] + t % P.win_wy[wi];
        row = rx < P.dx && ry < P.dy;
    }
    if (row) {
        const float wx = P.origin[0] + (float)rx * P.voxel - P.c2w[3];
        const float wy = P.origin[1] + (float)ry * P.voxel - P.c2w[7];
        const float wz = P.origin[2] - P.c2w[11];
        const float Ax = P.c2w[0] * wx + P.c2w[4] * wy + P.c2w[8] * wz;
        const float Ay = P.c2w[1] * wx + P.c2w[5] * wy + P.c2w[9] * wz;
        const float Az = P.c2w[2] * wx + P.c2w[6] * wy + P.c2w[10] * wz;
        const float Bx = P.c2w[8] * P.voxel, By = P.c2w[9] * P.voxel, Bz = P.c2w[10] * P.voxel;
        const float fx = P.K[0], fy = P.K[4], cx = P.K[2], cy = P.K[5];
        const float m = 0.05f;   // pixel margin
        const float mag = fabsf(Ax) + fabsf(Ay) + fabsf(Az) + (float)P.dz * (fabsf(Bx) + fabsf(By) + fabsf(Bz));
        const float eps = 1e-4f * fmaxf(fx, fy) * mag + 1e-4f;      // for the image-plane clips (pixel units x metres)
        const float eps_z = 1e-5f * mag + 1e-5f;                    // for the depth clips (metres)
        float lo = 0.0f, hi = (float)(P.dz - 1);
        bool empty = false;
        auto clip = [&](float a, float b, float e) {   // keep z with a + b z >= -e
            a += e;
            const float q = -a * __builtin_amdgcn_rcpf(b);   // ~1 ulp: absorbed by e and the +-1 voxel margin
            if (b > 0.0f)      lo = fmaxf(lo, q);
            else if (b < 0.0f) hi = fminf(hi, q);
            else if (a < 0.0f) empty = true;
        };
        clip(Az, Bz, eps_z);                                                          // cam_z > 0
        clip(fx * Ax + (cx + 0.5f + m) * Az, fx * Bx + (cx + 0.5f + m) * Bz, eps);    // px >= 0
        clip(((float)P.W - 0.5f + m - cx) * Az - fx * Ax, ((float)P.W - 0.5f + m - cx) * Bz - fx * Bx, eps);   // px < W
        clip(fy * Ay + (cy + 0.5f + m) * Az, fy * By + (cy + 0.5f + m) * Bz, eps);    // py >= 0
        clip(((float)P.H - 0.5f + m - cy) * Az - fy * Ay, ((float)P.H - 0.5f + m - cy) * Bz - fy * By, eps);   // py < H
        if (!empty && lo <= hi) {
            // The row is a straight line in the image: its pixels lie in the bounding box of the two end points.  No
            // voxel of the row can update if it is deeper than the deepest pixel of the coarse tiles that box touches
            // (+ trunc).
            const float za = fmaxf(Az + lo * Bz, 1e-6f), zb = fmaxf(Az + hi * Bz, 1e-6f);
            const float ra = __builtin_amdgcn_rcpf(za), rb = __builtin_amdgcn_rcpf(zb);
            const float u0 = fx * (Ax + lo * Bx) * ra + cx, v0 = fy * (Ay + lo * By) * ra + cy;
            const float u1 = fx * (Ax + hi * Bx) * rb + cx, v1 = fy * (Ay + hi * By) * rb + cy;
            int tu0 = max(0, (int)floorf((fminf(u0, u1) - 2.0f) / MV_TD)), tu1 = min(tw - 1, (int)floorf((fmaxf(u0, u1) + 2.0f) / MV_TD));
            int tv0 = max(0, (int)floorf((fminf(v0, v1) - 2.0f) / MV_TD)), tv1 = min(th - 1, (int)floorf((fmaxf(v0, v1) + 2.0f) / MV_TD));
            float tm = 0.0f;
            if (tu1 >= tu0 && tv1 >= tv0) {
                if (in_lds) {
                    int l = 0;
                    while (l + 1 < n_lvl && (tu1 - tu0 > 1 || tv1 - tv0 > 1)) { tu0 >>= 1; tu1 >>= 1; tv0 >>= 1; tv1 >>= 1; ++l; }
                    const float* __restrict__ lv = pyr + lvl_off[l];
                    const int w = lvl_w[l];
                    tm = fmaxf(fmaxf(lv[tv0 * w + tu0], lv[tv0 * w + tu1]), fmaxf(lv[tv1 * w + tu0], lv[tv1 * w + tu1]));
                } else {
                    for (int tv = tv0; tv <= tv1; ++tv)
                        for (int tu = tu0; tu <= tu1; ++tu) tm = fmaxf(tm, __uint_as_float(dmax_bits[MV_HDR + tv * tw + tu]));
                }
            }
            if (tm > 0.0f) {
                // cam_z <= (deepest pixel + trunc) / (1 - ratio_eps), on the far side
                clip((tm + P.trunc) / (1.0f - P.ratio_eps) * 1.0001f + 1e-3f - Az, -Bz, eps_z);
                if (!empty && lo <= hi) {
                    z0 = max(0, (int)floorf(lo) - 1);
                    z1 = min(P.dz, (int)ceilf(hi) + 2);
                    if (z1 < z0) z1 = z0;
                }
            }
        }
    }
    // ---- chunks of this row: [c0, c1) from the interval; a risky row also walks the chunks its alias zones touch
    const int nch_row = (P.dz + 63) >> 6;
    const int c0 = z0 >> 6, c1 = z1 > z0 ? (z1 + 63) >> 6 : c0;
    auto alias_chunk = [&](int c) {      // alias zones are a prefix and a suffix of the slab in index order: test the chunk's ends
        return alias_zone(P, ry, c << 6) || alias_zone(P, ry, min(P.dz, (c << 6) + 64) - 1);
    };
    int n_items = row ? c1 - c0 : 0;
    if (row && risky) {
        n_items = 0;
        for (int c = 0; c < nch_row; ++c) n_items += ((c >= c0 && c < c1) || alias_chunk(c)) ? 1 : 0;
    }
    // ---- append: wave prefix sums, then ONE returning atomic per block (a single counter word retires ~90 such atomics
    //      per microsecond; one per wave, ~900 a frame, would cost as much as the rest of this kernel).
    //      Where this kernel's 14.5 us go (round 5, timing builds that return early): launch + pyramid 4.8, + the rows'
    //      intervals 2.8, + scans, barriers and the atomic 5.8 (2 of them waiting for the atomic), + the item stores 1.
    //      A block-wide, coalesced form of the stores (item i by thread i, its row found by bisection) was built: 14.4 us.
    int incl = n_items;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    const int total = __shfl(incl, 63);
    if (lane == 0) { MV_STAT(0, 1); MV_STAT(1, total > 0 ? 1 : 0); }
    __shared__ int wave_total[16];
    __shared__ unsigned block_base;
    const int wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    if (lane == 63) wave_total[wv] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int w = 0; w < nwv; ++w) { const int tt = wave_total[w]; wave_total[w] = sum; sum += tt; }
        block_base = sum ? atomicAdd(q_counts + (risky ? 1 : 0), (unsigned)sum) : 0u;
    }
    __syncthreads();
    if (n_items == 0) return;
    unsigned pos = block_base + (unsigned)wave_total[wv] + (unsigned)(incl - n_items);
    MvItem it;
    it.xy = (unsigned)rx | ((unsigned)ry << 15);
    {
        const RowConst rc = make_row(P, (float)rx, (float)ry);
        it.ax = rc.ax; it.ay = rc.ay; it.az = rc.az;
        const uint64_t off = (uint64_t)((((int64_t)rx * P.dy + ry) * P.dz - slab_skip) * 4);     // slab-local byte offset
        it.off_lo = (unsigned)off; it.off_hi = (unsigned)(off >> 32);
        it.pad = 0u;
    }
    MvItem* __restrict__ q = risky ? queue_risky : queue;
    const unsigned cap = risky ? q_cap_risky : q_cap;
    for (int c = risky ? 0 : c0; c < (risky ? nch_row : c1); ++c) {
        const bool in_iv = c >= c0 && c < c1;
        if (!in_iv && !(risky && alias_chunk(c))) continue;
        const int lo_l = in_iv ? max(z0 - (c << 6), 0) : 0, hi_l = in_iv ? min(z1 - (c << 6), 64) : 0;
        it.cw = (unsigned)c | ((unsigned)lo_l << 16) | ((unsigned)hi_l << 22);
        if (pos < cap) q[pos] = it;
        ++pos;
    }
}

// Correctly rounded fp32 quotients a0/b and a1/b from ONE reciprocal: the instruction sequence hipcc emits for an
// IEEE division (v_rcp_f32, one Newton step on the reciprocal, two residual corrections of the quotient) without its
// range scaling (v_div_scale / v_div_fixup), which is the identity for the operands used here (|a| < 1e3, 1e-6 < b < 1e3,
// no denormal intermediate): bit-identical to a0 / b and a1 / b, 13 instructions instead of 22.
__device__ __forceinline__ void div2_shared(float a0, float a1, float b, float& q0, float& q1) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r = fmaf(e, r, r);
    float x = a0 * r, t = fmaf(-b, x, a0);
    x = fmaf(t, r, x); t = fmaf(-b, x, a0);
    q0 = fmaf(t, r, x);
    x = a1 * r; t = fmaf(-b, x, a1);
    x = fmaf(t, r, x); t = fmaf(-b, x, a1);
    q1 = fmaf(t, r, x);
}

__device__ __forceinline__ bool div2_in_range(float cxv, float cyv, float czv) {
    return czv > 1e-6f && czv < 1e3f && fabsf(cxv) < 1e3f && fabsf(cyv) < 1e3f;
}

// The voxel update proper (integrate_lanes' expression tree, Volume.py:285-334) for ONE voxel whose depth sample, stored
// values and colour sample are known (oc / ncl are only used inside the colour band).
__device__ __forceinline__ void update_voxel(const MvParams& P, int64_t idx, float cxv, float cyv, float czv, float d, float rl,
                                             float cur, float w_old, float oc, float ncl,
                                             float* __restrict__ tsdf, float* __restrict__ weight, float* __restrict__ color) {
    const float norm = sqrtf(madd(czv, czv, madd(cxv, cxv, cyv * cyv)));
    const float sdf = -madd(rl, norm, -d);
    const bool upd = (d > 0.0f) && (sdf >= -P.trunc);
    if (!upd) return;
    const bool band = sdf <= P.trunc;
    const float dist = fminf(1.0f, sdf / P.trunc);
    const float w_new = w_old + P.obs_weight;
    float new_t = madd(cur, w_old, P.obs_weight * dist) / w_new;
    float new_w = w_new;
    if (P.weight_clamp == 1) {
        new_w = fminf(w_new, 128.0f);
        if (new_w > 40.0f) new_w = 40.0f;
    }
    float new_c = 0.0f;
    if (band) {
        const float nc = ncl;
        float nb = floorf(nc / 65536.0f);
        float ng = floorf((nc - nb * 65536.0f) / 256.0f);
        float nr = nc - nb * 65536.0f - ng * 256.0f;
        const float ob = floorf(oc / 65536.0f);
        const float og = floorf((oc - ob * 65536.0f) / 256.0f);
        const float orr = oc - ob * 65536.0f - og * 256.0f;
        nb = fminf(roundf(madd(ob, w_old, P.obs_weight * nb) / w_new), 255.0f);
        ng = fminf(roundf(madd(og, w_old, P.obs_weight * ng) / w_new), 255.0f);
        nr = fminf(roundf(madd(orr, w_old, P.obs_weight * nr) / w_new), 255.0f);
        new_c = nb * 65536.0f + ng * 256.0f + nr;
    }
    const bool reset = (P.obs_weight == -1.0f) && (w_old <= 1.0f) && (P.reintegrate == 1);
    if (reset) { new_t = 1.0f; new_w = 0.0f; new_c = 0.0f; }
#if MV_DBG_NEAR & 1
    if (new_t == 12345.678f && new_w == 3.f && new_c == 7.f) tsdf[idx] = new_t;
#else
    tsdf[idx] = new_t;
    weight[idx] = new_w;
    if (band || reset) color[idx] = new_c;
#endif
}

// ---- the exact path ("near" voxels: truncation band, boundary pixels, voxels that held a surface before).
// Pending records of one wave, field-major in LDS: voxel index, camera point, the stored tsdf / weight as the fast path
// loaded them (24 B per record: re-reading them in the round instead was measured, the lines have left the L2 by then and
// the kernel fetched 75 MB more per frame).  Round 5: the records are
// NOT evaluated where they arise.  Up to round 4 a wave drained its list as soon as it held 64 records -- two dependent
// memory round trips (depth sample, then colour) in the middle of the item loop, 0 to 5 times per wave depending on how
// much surface its rows cross: 16 of the kernel's 44 us, most of it waves waiting behind their own drains while the rest
// of the chip had finished (tools/v1_wave_times.py: wave lifetimes 11-38 us around a mean of 27).  Now a wave only appends
// (a drain inside the loop happens when a list is about to overflow: MV_NEAR_CAP records); after the item loop the
// block's four lists are dealt out again in rounds of 64 records over the four waves (near_finish), and a round issues
// everything it reads -- depth sample, stored colour, colour sample -- at once: one round trip.
constexpr int MV_NEAR_FIELDS = 6;      // voxel index, camera point, tsdf and weight as the fast path loaded them
constexpr int MV_NEAR_CAP = 192;       // records per wave (4.5 KB; 8 blocks per CU: 147 of 160 KB); a list is drained inside
                                       // the loop above CAP - 64 (a frame of the bench leaves ~80 per wave)
// the append / drain invariant: a drain inside the loop leaves at most CAP - 64 records, one trip then appends at most 64
static_assert(MV_NEAR_CAP >= 128 && MV_NEAR_CAP % 64 == 0, "near lists: drained above CAP - 64, appended 64 at a time");
struct NearList {
    float (*nb)[MV_NEAR_CAP];
    int n;                 // wave-uniform fill
};

// the reference's full update on one record per lane: exact projection, depth sample, stored values, update
__device__ __forceinline__ void near_round(const MvParams& P, bool act, int64_t idx, float cxv, float cyv, float czv,
                                           float cur, float wold, const float4* __restrict__ nimg,
                                           float* __restrict__ tsdf, float* __restrict__ weight, float* __restrict__ color) {
    bool v = act && czv > 0.0f;
    const bool generic = v && !div2_in_range(cxv, cyv, czv);
    const float czs = v ? czv : 1.0f;
    float qx, qy;
    if (!__any(generic)) div2_shared(v ? cxv : 0.0f, v ? cyv : 0.0f, czs, qx, qy);
    else { qx = cxv / czs; qy = cyv / czs; }                       // operands outside div2_shared's range: the compiler's division
    const int px = f2i_rn(madd(P.K[0], qx, P.K[2]));
    const int py = f2i_rn(madd(P.K[4], qy, P.K[5]));
    v = v && px >= 0 && px < P.W && py >= 0 && py < P.H;
    if (v) {
        // both loads of the round are issued here (the stored colour speculatively: it is only used inside the band)
#if MV_DBG_NEAR & 2
        const float4 dl = make_float4(cxv + 2.0f, 0.5f, czv, 0.f);
        const float oc = cyv;
#else
#if MV_DBG_SKIP & 256
        const float4 dl = nimg[(P.dimg_colmajor ? px * P.H + py : py * P.W + px) & 0x1fff];
#else
        const float4 dl = nimg[P.dimg_colmajor ? px * P.H + py : py * P.W + px];
#endif
        const float oc = color[idx];
#endif
        update_voxel(P, idx, cxv, cyv, czv, dl.x, dl.y, cur, wold, oc, dl.z, tsdf, weight, color);
    }
}

// the top min(64, n) pending records of the wave's own list
__device__ __forceinline__ void near_drain(const MvParams& P, NearList& L, int lane, const float4* __restrict__ nimg,
                                           float* __restrict__ tsdf,
                                           float* __restrict__ weight, float* __restrict__ color) {
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's own LDS stores have landed
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const int take = min(64, L.n), base = L.n - take;
    const bool act = lane < take;
    const int s = base + (act ? lane : 0);
    const int64_t idx = (int64_t)__float_as_int(L.nb[0][s]);        // < 2^31 voxels (checked by the host)
    const float cxv = L.nb[1][s], cyv = L.nb[2][s], czv = L.nb[3][s], cur = L.nb[4][s], wold = L.nb[5][s];
    near_round(P, act, idx, cxv, cyv, czv, cur, wold, nimg, tsdf, weight, color);
    MV_STAT(6, (unsigned long long)take * (lane == 0));
    L.n = base;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();             // reads done before the buffer is written again
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void near_append(NearList& L, bool near_l, int lane, int idx, float cxv, float cyv, float czv,
                                            float cur, float wold) {
    const unsigned long long m = __ballot(near_l);
    if (!m) return;
    if (near_l) {
        const int s = L.n + __popcll(m & ((1ull << lane) - 1ull));
        L.nb[0][s] = __int_as_float(idx);
        L.nb[1][s] = cxv; L.nb[2][s] = cyv; L.nb[3][s] = czv; L.nb[4][s] = cur; L.nb[5][s] = wold;
    }
    L.n += __popcll(m);
}

// after the item loop (every wave of the block gets here): the block's pending records in rounds of 64, round r to wave r % 4
__device__ __forceinline__ void near_finish(const MvParams& P, float (*lists)[MV_NEAR_FIELDS][MV_NEAR_CAP], int* counts, int n_own,
                                            int wv, int lane, const float4* __restrict__ nimg,
                                            float* __restrict__ tsdf, float* __restrict__ weight, float* __restrict__ color) {
    if (lane == 0) counts[wv] = n_own;
    __syncthreads();
    const int n0 = counts[0], n1 = counts[1], n2 = counts[2], n3 = counts[3];
    const int total = n0 + n1 + n2 + n3;
    for (int r = wv; r * 64 < total; r += 4) {
        const int g = r * 64 + lane;
        const bool act = g < total;
        const int gc = act ? g : 0;
        const int w = (gc >= n0) + (gc >= n0 + n1) + (gc >= n0 + n1 + n2);
        const int s = gc - (w > 0 ? n0 : 0) - (w > 1 ? n1 : 0) - (w > 2 ? n2 : 0);
        const int64_t idx = (int64_t)__float_as_int(lists[w][0][s]);
        const float cxv = lists[w][1][s], cyv = lists[w][2][s], czv = lists[w][3][s], cur = lists[w][4][s], wold = lists[w][5][s];
        near_round(P, act, idx, cxv, cyv, czv, cur, wold, nimg, tsdf, weight, color);
        MV_STAT(6, (unsigned long long)min(64, total - r * 64) * (lane == 0));
    }
}

constexpr unsigned MV_OOB = 0x80000000u;      // a buffer offset beyond every descriptor's range: the access is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mv_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0, (int)bytes, 0x00020000);
}

// ---- fast body (obs_weight > 0, no re-integration, exactly decoded rows)
// What bounds it, as measured in round 5 (profiles/r5_notes.md; timing builds -DMV_DBG_SKIP / -DMV_DBG_NEAR remove one stream at a
// time, -DMV_BLOCKS_PER_CU the occupancy, tools/v1_wave_times.py gives the waves' lifetimes):
//   * 19 us with no volume access and no exact path: vector-instruction issue (62 VALU + 50 SALU per item; the classification
//     gather itself costs < 1 us, whatever its addresses);
//   * + 4.0 / 3.4 / 4.5 us for the tsdf load / weight load / weight store, + ~10 us for the exact path (3.7 us of it arithmetic,
//     the rest its gathers and its three scattered stores);
//   * by occupancy T = 31 us + 60 us / (waves per SIMD): at 8 waves 7.5 us of exposed latency on a 31 us throughput floor.
//   The fabric is NOT the bound although FETCH_SIZE x 2 (calibrated on this access shape: the L2 fetches whole 128-byte lines,
//   the counter tallies them at 64) reads 170 MB for 85 MB of algorithmic work: with every image gather confined to an
//   L2-resident window the counter drops by 80 MB and the kernel by 2 us -- the surplus is 8 L2s each pulling its own copy of
//   the images out of the Infinity Cache.  Software pipelining (the loads of trip t + 1 issued before trip t is classified)
//   was built: 74 registers, 6 waves per SIMD, 43.7 us against 38.3.
// Structure: the volume reads are plain loads from a scalar base with a lane offset clamped into the row (no descriptor to
// build per item), the weight store is the only operation under a lane mask, the classification gather is a buffer load
// through ONE loop-invariant descriptor whose range check drops the lanes that have no pixel.
// `n` items in the caller's LOCAL numbering; local item j is queue[deal.global(j)] (see mv_chunks_kernel)
struct MvDeal {
    unsigned shift, part, parts;         // segments of 2^shift items, segment s of the queue belongs to part s % parts
    __device__ __forceinline__ unsigned global(unsigned j) const { return (((j >> shift) * parts + part) << shift) | (j & ((1u << shift) - 1u)); }
};
template <int U>
__device__ __forceinline__ void mv_chunks_fast(const MvParams& P, MvDeal deal, unsigned n, const MvItem* __restrict__ queue,
                                               const float4* __restrict__ nimg, const float2* __restrict__ fg,
                                               float* __restrict__ tsdf,
                                               float* __restrict__ weight, float* __restrict__ color, unsigned wave,
                                               unsigned n_waves, NearList& L) {
    const int lane = threadIdx.x & 63;
    const float lane_f = (float)lane;
    const __amdgpu_buffer_rsrc_t r_fg = mv_rsrc(fg, (unsigned)P.H * (unsigned)P.W * 8u);
    // image byte offset = px * sx8 + py * sy8, formed in fp32 (exact below 2^24: the host sends H * W * 8 >= 2^24 elsewhere)
    const float sx8 = P.dimg_colmajor ? 8.0f * (float)P.H : 8.0f, sy8 = P.dimg_colmajor ? 8.0f : 8.0f * (float)P.W;
    const float edge = 0.5f - P.edge_eps;
    const float hw = 0.5f * (float)(P.W - 1), hh = 0.5f * (float)(P.H - 1);      // pixel px is in the image <=> |px - hw| <= hw
    // ---- the wave's items.  Up to round 4 every trip began with scalar loads of its U items: one dependent round trip to
    //      memory per trip in front of the round trip of the volume loads (~10 trips per wave, the waves 62 % of their
    //      lifetime in s_waitcnt).  Now ONE vector gather per 16 trips puts 16 * U items into four registers -- lane 2k and
    //      2k + 1 hold the two 16-byte halves of the batch's k-th item, the next batch is in flight while this one is
    //      worked on -- and a trip takes its items out with v_readlane: no memory access between two trips.
    constexpr unsigned TRIPS = 32u / (unsigned)U;                              // trips per batch: 32 items in 64 lanes
    const unsigned first = wave * (unsigned)U, stride = n_waves * (unsigned)U;
    const unsigned n_trips = first < n ? (n - first + stride - 1u) / stride : 0u;
    const unsigned last_item = n ? n - 1u : 0u;
    auto fetch = [&](unsigned trip0) -> uint4 {      // items of trips [trip0, trip0 + TRIPS): item k of the batch -> lanes 2k, 2k + 1
        const unsigned k = (unsigned)lane >> 1;
        const unsigned idx = min(first + (trip0 + k / (unsigned)U) * stride + k % (unsigned)U, last_item);
        return reinterpret_cast<const uint4*>(queue)[(size_t)deal.global(idx) * 2u + ((unsigned)lane & 1u)];
    };
    uint4 batch = make_uint4(0u, 0u, 0u, 0u), batch_next = batch;
    if (n_trips) batch = fetch(0u);
    for (unsigned trip = 0; trip < n_trips; ++trip) {
        const unsigned it = first + trip * stride;
        const unsigned tb = trip % TRIPS;
        if (tb == 0u) {
            if (trip) batch = batch_next;
            if (trip + TRIPS < n_trips) batch_next = fetch(trip + TRIPS);
        }
        float cxv[U], cyv[U], czv[U], cur[U], wold[U];
        float2 fgv[U];
        bool inside[U], exact[U];
        int idx[U];
        float* wp[U];
        MvItem items[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l0 = (int)(tb * (unsigned)U + (unsigned)u) * 2;
            items[u].cw = (unsigned)__builtin_amdgcn_readlane((int)batch.y, l0);
            items[u].ax = __int_as_float(__builtin_amdgcn_readlane((int)batch.z, l0));
            items[u].ay = __int_as_float(__builtin_amdgcn_readlane((int)batch.w, l0));
            items[u].az = __int_as_float(__builtin_amdgcn_readlane((int)batch.x, l0 + 1));
            items[u].off_lo = (unsigned)__builtin_amdgcn_readlane((int)batch.y, l0 + 1);
            items[u].off_hi = (unsigned)__builtin_amdgcn_readlane((int)batch.z, l0 + 1);
        }
        // ---- stage 0: volume loads, camera point, approximate projection, classification gather.  Lane predicates are
        //      combined with & (no short-circuit: a && here becomes a divergent branch with its scalar bookkeeping)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = it + u < n;
            const MvItem& I = items[u];
            const unsigned cw = (unsigned)__builtin_amdgcn_readfirstlane((int)I.cw);
            const int c = (int)(cw & 0xffffu);
            const int lo_l = (int)((cw >> 16) & 63u), len_l = live ? (int)((cw >> 22) & 127u) - lo_l : 0;     // hi >= lo
            const uint64_t off = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)I.off_hi) << 32 |
                                  (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)I.off_lo)) + (uint64_t)c * 256u;
            const float* __restrict__ tp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(tsdf) + off);
            wp[u] = reinterpret_cast<float*>(reinterpret_cast<char*>(weight) + off);
            // lanes outside the row's interval read the nearest voxel inside it: the same 64-byte sectors as their
            // neighbours, so a partial chunk costs the memory system only the sectors its interval touches
            const unsigned lane_c = (unsigned)min(max(lane, lo_l), max(lo_l + len_l - 1, lo_l));
#if MV_DBG_SKIP & 1        // timing builds only (results wrong): one memory stream of the fast body removed at a time
            cur[u] = 1.0f; (void)tp;
#else
            cur[u] = tp[lane_c];
#endif
#if MV_DBG_SKIP & 2
            wold[u] = (float)(lane_c & 1u);
#else
            wold[u] = wp[u][lane_c];
#endif
            inside[u] = (unsigned)(lane - lo_l) < (unsigned)len_l;
            idx[u] = (int)(off >> 2) + lane;
            // camera point, exactly as the reference forms it
            const float zf = lane_f + (float)(c << 6);
            const float pz = madd(zf, P.voxel, P.origin[2]);
            const float tz = pz - P.c2w[11];
            const float ax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, I.ax)));
            const float ay = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, I.ay)));
            const float az = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, I.az)));
            cxv[u] = madd(P.c2w[8], tz, ax);
            cyv[u] = madd(P.c2w[9], tz, ay);
            czv[u] = madd(P.c2w[10], tz, az);
            // approximate projection.  Its rounded pixel is the reference's unless a coordinate lies within edge_eps of a
            // rounding boundary (bound: see edge_eps; it holds for every cz >= 1e-6, whatever cx and cy -- a quotient that
            // overflows does so in the reference too and lands outside the image either way); those lanes, and lanes with
            // 0 < cz < 1e-6, take the exact path
            const float rz = __builtin_amdgcn_rcpf(czv[u]);
            const float uu = madd(P.K[0], cxv[u] * rz, P.K[2]);
            const float vv = madd(P.K[4], cyv[u] * rz, P.K[5]);
            const float ur = rintf(uu), vr = rintf(vv);
            const bool approx_ok = (fmaxf(fabsf(uu - ur), fabsf(vv - vr)) <= edge) & (czv[u] >= 1e-6f);
            const bool in_img = (fabsf(ur - hw) <= hw) & (fabsf(vr - hh) <= hh);
            exact[u] = inside[u] & (czv[u] > 0.0f) & !approx_ok;
            const bool gather = inside[u] & approx_ok & in_img;
#if MV_DBG_SKIP & 256      // timing builds: every gather inside one 128 KB window of the image (no image traffic beyond the L2)
            const unsigned goff = gather ? ((unsigned)madd(ur, sx8, vr * sy8) & 0x1fff8u) : MV_OOB;
#else
            const unsigned goff = gather ? (unsigned)madd(ur, sx8, vr * sy8) : MV_OOB;
#endif
            const auto g2 = __builtin_amdgcn_raw_buffer_load_b64(r_fg, goff, 0, 0);
            const unsigned g2x = g2[0], g2y = g2[1];        // (bit_cast of a vector element reads element 0 twice: clang 22)
            fgv[u] = make_float2(__uint_as_float(g2x), __uint_as_float(g2y));
            MV_STAT(2, live && lane == 0 ? 1 : 0);
            MV_STAT(3, __popcll(__ballot(inside[u])) * (lane == 0));
            MV_STAT(4, __popcll(__ballot(gather)) * (lane == 0));
        }
        // ---- stage 1: classify.  Lanes without a gather read F = G = 0: never free, always untouched.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float n2 = madd(czv[u], czv[u], madd(cxv[u], cxv[u], cyv[u] * cyv[u]));
            const bool free_l = n2 < fgv[u].x, one = cur[u] == 1.0f;
            const bool free_1 = free_l & one;                              // free space and still 1: only the weight moves
            const bool band = (n2 >= fgv[u].x) & (n2 <= fgv[u].y);         // neither provably free nor provably untouched
            // fminf(w, 128) then "> 40 -> 40" of the reference is min(w, 40) for every input, NaN included
            const float w_sum = wold[u] + P.obs_weight;
            const float new_w = P.weight_clamp == 1 ? fminf(w_sum, 40.0f) : w_sum;
#if !(MV_DBG_SKIP & 8)
            if (free_1 & (new_w != wold[u])) wp[u][(unsigned)lane] = new_w;           // tsdf stays 1 (see the header)
#endif
            MV_STAT(5, __popcll(__ballot(free_1)) * (lane == 0));
            const bool near_l = exact[u] | band | (free_l & !one);
            // 8: items none of whose lanes is touched; 9: their in-interval lanes; 10: items with a weight store; 11: items with near lanes
            MV_STAT(8, (it + u < n) && !__any(free_1 | near_l) && lane == 0 ? 1 : 0);
            MV_STAT(9, !__any(free_1 | near_l) ? __popcll(__ballot(inside[u])) * (lane == 0) : 0);
            MV_STAT(10, __any(free_1 & (new_w != wold[u])) && lane == 0 ? 1 : 0);
            MV_STAT(11, __any(near_l) && lane == 0 ? 1 : 0);
#if MV_DBG_SKIP & 16
            if (__any(near_l) && cxv[u] == 12345.678f) tsdf[0] = 0.f;
#else
            near_append(L, near_l, lane, idx[u], cxv[u], cyv[u], czv[u], cur[u], wold[u]);
            if (L.n > MV_NEAR_CAP - 64) near_drain(P, L, lane, nimg, tsdf, weight, color);
#endif
        }
    }
}

// ---- generic body: every lane of an item through the exact projection; free-space shortcut only when `fast`
template <int U, bool RISKY, bool REINT>
__device__ __forceinline__ void mv_chunks_generic(const MvParams& P, unsigned begin, unsigned n, const MvItem* __restrict__ queue,
                                                  const float4* __restrict__ nimg,
                                                  float* __restrict__ tsdf, float* __restrict__ weight,
                                                  float* __restrict__ color, unsigned wave, unsigned n_waves, NearList& L,
                                                  int64_t slab_skip) {
    const int lane = threadIdx.x & 63;
    const bool fast = !REINT && P.obs_weight > 0.0f;
    for (unsigned it = begin + wave * U; it < n; it += n_waves * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // every load below is UNCONDITIONAL with an in-bounds address (a load under a lane mask gets its own basic block
            // and s_waitcnt, which strings the round trips together): an item past the end is replaced by the trip's first
            const bool live = it + u < n;
            const MvItem I = queue[live ? it + u : it];
            const unsigned ix = (unsigned)__builtin_amdgcn_readfirstlane((int)I.xy), cw = (unsigned)__builtin_amdgcn_readfirstlane((int)I.cw);
            const int rx = ix & 0x7fff, ry = (ix >> 15) & 0x7fff;
            const int c = cw & 0xffff, lo_l = (cw >> 16) & 63, hi_l = (cw >> 22) & 127;
            const int z = (c << 6) + lane;
            bool v = live && lane >= lo_l && lane < hi_l;
            if (RISKY) v = live && z < P.dz && (v || alias_zone(P, ry, z));
            const int64_t row0 = ((int64_t)rx * P.dy + ry) * P.dz - slab_skip;        // slab-local index of the row
            const int64_t gidx = row0 + slab_skip + z;                                // global index: what the literal decode sees
            const int zc = min(z, P.dz - 1);
            const float cur = (tsdf + row0)[zc], wold = (weight + row0)[zc];
            float fvx = (float)rx, fvy = (float)ry, fvz = (float)z;
            if (RISKY) decode_literal(v ? (int)gidx : 0, P.dy, P.dz, fvx, fvy, fvz);
            const RowConst rc = make_row(P, fvx, fvy);
            const float pz = madd(fvz, P.voxel, P.origin[2]);
            if (REINT) v = v && !outside_old(P, rc.px, rc.py, pz);
            const float tz = pz - P.c2w[11];
            const float cxv = madd(P.c2w[8], tz, rc.ax);
            const float cyv = madd(P.c2w[9], tz, rc.ay);
            const float czv = madd(P.c2w[10], tz, rc.az);
            MV_STAT(2, live && lane == 0 ? 1 : 0);
            MV_STAT(7, live && RISKY && lane == 0 ? 1 : 0);
            MV_STAT(3, __popcll(__ballot(v)) * (lane == 0));
            bool near_l = v && czv > 0.0f;
            if (fast) {
                // exact pixel, then the free-space / untouched tests with a 1-ulp square root and a margin far above its error
                const bool generic = near_l && !div2_in_range(cxv, cyv, czv);
                const float czs = near_l ? czv : 1.0f;
                float qx, qy;
                if (!__any(generic)) div2_shared(near_l ? cxv : 0.0f, near_l ? cyv : 0.0f, czs, qx, qy);
                else { qx = cxv / czs; qy = cyv / czs; }
                const int px = f2i_rn(madd(P.K[0], qx, P.K[2]));
                const int py = f2i_rn(madd(P.K[4], qy, P.K[5]));
                const bool ok = near_l && px >= 0 && px < P.W && py >= 0 && py < P.H;
                const float2 dl = *reinterpret_cast<const float2*>(nimg + (ok ? (P.dimg_colmajor ? px * P.H + py : py * P.W + px) : 0));
                const float d = dl.x;
                const float norm_a = __builtin_amdgcn_sqrtf(madd(czv, czv, madd(cxv, cxv, cyv * cyv)));
                const float sdf_a = d - dl.y * norm_a;
                // |sdf~ - sdf| <= ~2.5 ulp of max(norm, d) (1-ulp square root, two more roundings): margin = 5 ulp + 1e-7
                const float margin = fmaf(6e-7f, norm_a + fabsf(d), 1e-7f);
                const bool cand = ok && (d > 0.0f) && (sdf_a >= -P.trunc - margin);          // else: provably untouched
                const bool free_l = cand && (sdf_a > P.trunc + margin) && (cur == 1.0f);
                if (free_l) {
                    float new_w = wold + P.obs_weight;
                    if (P.weight_clamp == 1) {
                        new_w = fminf(new_w, 128.0f);
                        if (new_w > 40.0f) new_w = 40.0f;
                    }
                    if (new_w != wold) (weight + row0)[z] = new_w;       // tsdf stays 1 (see the header)
                }
                MV_STAT(5, __popcll(__ballot(free_l)) * (lane == 0));
                near_l = cand && !free_l;
            }
            near_append(L, near_l, lane, (int)(row0 + z), cxv, cyv, czv, cur, wold);
            if (L.n > MV_NEAR_CAP - 64) near_drain(P, L, lane, nimg, tsdf, weight, color);
        }
    }
}

// one launch for both queues: blocks [0, blocks_main) pull the frustum rows' chunks, the rest the chunks of the rows next to
// x-slab boundaries (literal index decode; a few thousand items -- a launch of their own cost 5 us behind the main one)
// Residency: the grid is sized to the blocks that are resident at once (the items are dealt statically over gridDim, so a
// block that has to wait for a slot doubles the kernel's time).  The hardware admits min(8, 800 / (ceil(sgpr / 16) * 16 + 16))
// 256-thread blocks per CU, which for 98+ SGPRs is one block fewer than hipOccupancyMaxActiveBlocksPerMultiprocessor
// answers (MI355X_MICROARCH.md, "Residency and cooperative launch"): the kernel is held to 80 SGPRs, where both say 8.
template <int U, bool FAST, bool REINT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void mv_chunks_kernel(MvParams P, const unsigned* __restrict__ q_counts,
                                                        const MvItem* __restrict__ queue, unsigned q_cap,
                                                        const MvItem* __restrict__ queue_risky, unsigned q_cap_risky,
                                                        unsigned blocks_main, int64_t slab_skip,
                                                        const float4* __restrict__ nimg, const float2* __restrict__ fg,
                                                        float* __restrict__ tsdf, float* __restrict__ weight,
                                                        float* __restrict__ color) {
    __shared__ float nbuf[4][MV_NEAR_FIELDS][MV_NEAR_CAP];
    __shared__ int near_counts[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef MV_TIMING
    const unsigned long long t_start = wall_clock64();
#endif
    NearList L;
    L.nb = nbuf[wv];
    L.n = 0;
    if (blockIdx.x < blocks_main) {
        const unsigned n = min(q_counts[0], q_cap);
        if (FAST) {
#if MV_XCD_DEAL
            // XCD-aware dealing.  Workgroups go to the 8 XCDs round-robin (blockIdx % 8; an affinity for speed, never for
            // correctness) and every XCD has an L2 of its own: with the items dealt cyclically over the whole grid each L2
            // sees rows from all over the frustum, pulls in its own copy of the whole {F, G} / depth / colour images and
            // shares the lines between neighbouring chunks of a row with another XCD (measured: 12 % of the kernel's
            // fetches).  The queue (rows in (x, y) order) is cut into segments of 2^shift items, ~MV_XCD_SEGS per XCD, and
            // segment s goes to the blocks with blockIdx % 8 == s % 8, whose waves sweep their segments side by side:
            // a strip of the footprint per L2 at any time, and the strips of one XCD spread over the frustum so that each
            // XCD gets its share of the surface-crossing rows (one contiguous eighth each left the XCDs 21-33 us apart).
            const unsigned parts = min(8u, blocks_main);                                   // a grid of fewer than 8 blocks: as many parts
            const unsigned part = blockIdx.x % parts, nb = (blocks_main - part + parts - 1u) / parts;      // main blocks of this part (>= 1)
            const unsigned want = max(64u, n / (parts * (unsigned)MV_XCD_SEGS));
            MvDeal deal; deal.shift = 31u - (unsigned)__builtin_clz(want); deal.part = part; deal.parts = parts;
            const unsigned full = n >> deal.shift, rem = n & ((1u << deal.shift) - 1u);
            const unsigned my_full = full > part ? (full - part + parts - 1u) / parts : 0u;
            const unsigned n_local = (my_full << deal.shift) + (full % parts == part ? rem : 0u);
            const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x / parts) * 4u + (unsigned)wv), n_waves = nb * 4u;
#else
            MvDeal deal; deal.shift = 31u; deal.part = 0u; deal.parts = 1u;
            const unsigned n_local = n;
            const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (unsigned)wv), n_waves = blocks_main * 4u;
#endif
            mv_chunks_fast<U>(P, deal, n_local, queue, nimg, fg, tsdf, weight, color, wave, n_waves, L);
        } else {
            const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (unsigned)wv), n_waves = blocks_main * 4u;
            mv_chunks_generic<U, false, REINT>(P, 0u, n, queue, nimg, tsdf, weight, color, wave, n_waves, L, slab_skip);
        }
    } else {
        const unsigned n = min(q_counts[1], q_cap_risky);
        const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x - blocks_main) * 4u + (unsigned)wv);
        mv_chunks_generic<U, true, REINT>(P, 0u, n, queue_risky, nimg, tsdf, weight, color, wave, (gridDim.x - blocks_main) * 4u, L,
                                          slab_skip);
    }
    near_finish(P, nbuf, near_counts, L.n, wv, lane, nimg, tsdf, weight, color);
#ifdef MV_TIMING
    if (lane == 0 && blockIdx.x * 4u + wv < 2u * 8192u) {
        __builtin_amdgcn_s_waitcnt(0);          // the wave's stores have left
        g_mv_times[2 * (blockIdx.x * 4 + wv)] = t_start;
        g_mv_times[2 * (blockIdx.x * 4 + wv) + 1] = wall_clock64();
    }
#endif
}

// ---------------------------------------------------------------------------- simple sweeps
__global__ __launch_bounds__(256) void pack_color_kernel(const float* __restrict__ rgb, float* __restrict__ out,
                                                         int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        out[i] = floorf(b * 65536.0f + g * 256.0f + r);   // np.floor(B*65536 + G*256 + R), left to right
    }
}

__global__ __launch_bounds__(256) void mv_fill_kernel(float* __restrict__ t, float* __restrict__ w,
                                                      float* __restrict__ c, int64_t n) {
    const int64_t n4 = n >> 2;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        reinterpret_cast<float4*>(t)[i] = one;
        reinterpret_cast<float4*>(w)[i] = zero;
        reinterpret_cast<float4*>(c)[i] = zero;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = (n4 << 2) + threadIdx.x;
        t[i] = 1.f; w[i] = 0.f; c[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void mv_copy_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                                      const float* __restrict__ c, float* __restrict__ tb,
                                                      float* __restrict__ wb, float* __restrict__ cb, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(t)[i];
        float4 b = reinterpret_cast<const float4*>(w)[i];
        float4 d = reinterpret_cast<const float4*>(c)[i];
        reinterpret_cast<float4*>(tb)[i] = a;
        reinterpret_cast<float4*>(wb)[i] = b;
        reinterpret_cast<float4*>(cb)[i] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = (n4 << 2) + threadIdx.x;
        tb[i] = t[i]; wb[i] = w[i]; cb[i] = c[i];
    }
}

struct ShiftParams {
    int dx, dy, dz, odx, ody, odz;
    float origin[3], old_origin[3], voxel;
    int risky_rows, literal_all;
    int x0;           // first x-plane of the destination slab (its arrays start there); 0 for a whole volume
    int ox_a, ox_b;   // x-planes [ox_a, ox_b) of the OLD volume that the source arrays hold (0, odx for a whole volume)
};

// V2: one block per (x,y) row of the NEW volume, threads along z.
__global__ __launch_bounds__(256) void mv_shift_kernel(ShiftParams P, float* __restrict__ t, float* __restrict__ w,
                                                       float* __restrict__ c, const float* __restrict__ ot,
                                                       const float* __restrict__ ow, const float* __restrict__ oc) {
    const int row_l = blockIdx.x;                    // row of the destination slab
    const int rx = P.x0 + row_l / P.dy, ry = row_l % P.dy;
    const bool risky = P.literal_all || ry < P.risky_rows || ry >= P.dy - P.risky_rows;
    const int64_t base = (int64_t)row_l * P.dz;                       // slab-local
    const int64_t gbase = ((int64_t)rx * P.dy + ry) * P.dz;          // global index: what the literal decode sees
    for (int z = threadIdx.x; z < P.dz; z += blockDim.x) {
        float vx = (float)rx, vy = (float)ry, vz = (float)z;
        if (risky) decode_literal((int)(gbase + z), P.dy, P.dz, vx, vy, vz);
        const float wx = madd(vx, P.voxel, P.origin[0]);
        const float wy = madd(vy, P.voxel, P.origin[1]);
        const float wz = madd(vz, P.voxel, P.origin[2]);
        const int ox = (int)roundf((wx - P.old_origin[0]) / P.voxel);
        const int oy = (int)roundf((wy - P.old_origin[1]) / P.voxel);
        const int oz = (int)roundf((wz - P.old_origin[2]) / P.voxel);
        float a = 1.0f, b = 0.0f, d = 0.0f;
        if (P.ox_a <= ox && ox < P.ox_b && 0 <= oy && oy < P.ody && 0 <= oz && oz < P.odz) {
            const int64_t o = (int64_t)oz + (int64_t)oy * P.odz + (int64_t)(ox - P.ox_a) * P.ody * P.odz;
            a = ot[o]; b = ow[o]; d = oc[o];
        }
        t[base + z] = a; w[base + z] = b; c[base + z] = d;
    }
}

struct VolView { int dx, dy, dz; float origin[3]; float voxel; };

// V3: one thread per query point; accumulators are double like the reference's `auto x = 0.0`.
// Slab form (one GPU of several holds the x-planes [x0, x1) of the volume): every coordinate is formed from the WHOLE
// volume's origin, so a point gets the very cell and weights it gets on one GPU; the point belongs to the rank that owns
// the plane of its lower corner, which reads the upper plane from its own slab or, at its last plane, from the one-plane
// halo its right neighbour sent (tsdf_halo / color_halo = plane x1).  inside[p] = 1 where this rank produced the result.
__global__ __launch_bounds__(256) void mv_trilerp_kernel(VolView V, int x0, int x1, const float* __restrict__ tsdf,
                                                         const float* __restrict__ color,
                                                         const float* __restrict__ tsdf_halo, const float* __restrict__ color_halo,
                                                         const float* __restrict__ pts, int64_t n,
                                                         float* __restrict__ out, unsigned char* __restrict__ inside) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float x = pts[p * 3], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
    const int lx = (int)floorf((x - V.origin[0]) / V.voxel);
    const int ly = (int)floorf((y - V.origin[1]) / V.voxel);
    const int lz = (int)floorf((z - V.origin[2]) / V.voxel);
    const float xo = madd((float)lx, V.voxel, V.origin[0]);
    const float yo = madd((float)ly, V.voxel, V.origin[1]);
    const float zo = madd((float)lz, V.voxel, V.origin[2]);
    float* o = out + p * 5;
    const bool whole = x0 == 0 && x1 == V.dx;
    if (lx < 0 || lx >= V.dx - 1 || ly < 0 || ly >= V.dy - 1 || lz < 0 || lz >= V.dz - 1) {
        // outside the volume: the reference's default record.  In slab form the first rank reports it (one owner per point)
        const bool mine = whole || x0 == 0;
        if (inside) inside[p] = mine ? 1 : 0;
        if (mine) { o[0] = 1.0f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
        return;
    }
    if (lx < x0 || lx >= x1) { if (inside) inside[p] = 0; return; }
    if (inside) inside[p] = 1;
    const float u = (x - xo) / V.voxel, v = (y - yo) / V.voxel, w = (z - zo) / V.voxel;
    double t = 0.0, cb = 0.0, cg = 0.0, cr = 0.0;
    float t_low = 0.f;
    const int64_t plane = (int64_t)V.dy * V.dz;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t in_plane = (int64_t)(lz + k) + (int64_t)(ly + j) * V.dz;
                const bool halo = lx + i >= x1;                       // only i = 1 at the slab's last plane
                const float tv = halo ? tsdf_halo[in_plane] : tsdf[in_plane + (int64_t)(lx + i - x0) * plane];
                if (!i && !j && !k) t_low = tv;
                const float cc = halo ? color_halo[in_plane] : color[in_plane + (int64_t)(lx + i - x0) * plane];
                const float b = floorf(cc / 65536.0f);
                const float g = floorf((cc - b * 65536.0f) / 256.0f);
                const float r = floorf(cc - b * 65536.0f - g * 256.0f);
                const float wu = madd((float)i, u, (float)(1 - i) * (1.0f - u));
                const float wv = madd((float)j, v, (float)(1 - j) * (1.0f - v));
                const float ww = madd((float)k, w, (float)(1 - k) * (1.0f - w));
                const float wt = wu * wv * ww;
                t += (double)(wt * tv); cb += (double)(wt * b); cg += (double)(wt * g); cr += (double)(wt * r);
            }
    o[0] = (float)t; o[1] = (float)floor(cr); o[2] = (float)floor(cg); o[3] = (float)floor(cb); o[4] = t_low;
}

__global__ __launch_bounds__(256) void mv_filter_kernel(float* __restrict__ t, float* __restrict__ w,
                                                        float* __restrict__ c, int64_t n, float thr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        if (wi >= thr || wi == 0.0f) continue;
        w[i] = 0.f; t[i] = 1.f; c[i] = 0.f;
    }
}

// V5: one thread per output slot; walks the voxels that map to the slot from the highest index
// down, so the survivor is deterministic ("last writer in index order").
// [i0, i1): the global voxel indices held by tsdf / color (one x-slab of the volume; the whole volume: [0, n)); hit (optional):
// hit[s] = 1 where this slab wrote slot s, for the merge over the slabs (the slab with the highest indices wins a slot).
__global__ __launch_bounds__(256) void mv_truncated_pc_kernel(VolView V, const float* __restrict__ tsdf,
                                                              const float* __restrict__ color, float trunc,
                                                              int pc_num, float tt, float* __restrict__ pc7,
                                                              unsigned* __restrict__ count, int risky_rows,
                                                              int literal_all, int64_t i0, int64_t i1,
                                                              unsigned char* __restrict__ hit) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned found = 0;
    if (s < pc_num) {
        int64_t kmax = i1 - 1 - s >= 0 ? (i1 - 1 - s) / pc_num : -1;
        bool written = false;
        for (int64_t k = kmax; k >= 0 && s + k * pc_num >= i0; --k) {
            const int64_t idx = s + k * pc_num;
            const float t = tsdf[idx - i0];
            if (t <= -tt || t >= tt) continue;
            ++found;
            if (written) continue;
            written = true;
            const float oc = color[idx - i0];
            const float ob = floorf(oc / 65536.0f);
            const float og = floorf((oc - ob * 65536.0f) / 256.0f);
            const float orr = oc - ob * 65536.0f - og * 256.0f;
            const int64_t x = idx / ((int64_t)V.dy * V.dz);
            const int64_t r = idx - x * V.dy * V.dz;
            const int64_t y = r / V.dz;
            float vx = (float)x, vy = (float)y, vz = (float)(r - y * V.dz);
            if (literal_all || y < risky_rows || y >= V.dy - risky_rows) decode_literal((int)idx, V.dy, V.dz, vx, vy, vz);
            float* o = pc7 + (int64_t)s * 7;
            o[0] = madd(vx + 0.5f, V.voxel, V.origin[0]);
            o[1] = madd(vy + 0.5f, V.voxel, V.origin[1]);
            o[2] = madd(vz + 0.5f, V.voxel, V.origin[2]);
            o[3] = t * trunc; o[4] = orr; o[5] = og; o[6] = ob;
        }
        if (hit) hit[s] = written ? 1 : 0;
    }
    // one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) found += __shfl_xor(found, o);
    if ((threadIdx.x & 63) == 0 && found) atomicAdd(count, found);
}

// ---------------------------------------------------------------------------- GBV
struct GbvParams {
    float K[9];
    float box[6];
    int   res, H, W;
    float voxel_size, trunc, obs_weight;
};

// G1: one thread per voxel, x fastest; literal decode (cheap at 8e6 voxels); trgb as float4.
__global__ __launch_bounds__(256) void gbv_integrate_kernel(GbvParams P, const float* __restrict__ c2w,
                                                            float4* __restrict__ trgb, float* __restrict__ w,
                                                            const float* __restrict__ rgb,
                                                            const float* __restrict__ depth) {
    const int R = P.res;
    const int n = R * R * R;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float vz = floorf(((float)idx) / ((float)(R * R)));
    const float vy = floorf(((float)(idx - ((int)vz) * R * R)) / ((float)R));
    const float vx = (float)(idx - ((int)vz) * R * R - ((int)vy) * R);
    const float px = madd(vx * P.voxel_size, P.box[1] - P.box[0], P.box[0]);
    const float py = madd(vy * P.voxel_size, P.box[3] - P.box[2], P.box[2]);
    const float pz = madd(vz * P.voxel_size, P.box[5] - P.box[4], P.box[4]);
    const float tx = px - c2w[3], ty = py - c2w[7], tz = pz - c2w[11];
    const float cx = madd(c2w[8], tz, madd(c2w[0], tx, c2w[4] * ty));
    const float cy = madd(c2w[9], tz, madd(c2w[1], tx, c2w[5] * ty));
    const float cz = madd(c2w[10], tz, madd(c2w[2], tx, c2w[6] * ty));
    if (cz <= 0.0f) return;
    const int ix = f2i_rn(madd(P.K[0], cx / cz, P.K[2]));
    const int iy = f2i_rn(madd(P.K[4], cy / cz, P.K[5]));
    if (ix < 0 || ix >= P.W || iy < 0 || iy >= P.H) return;
    const int pix = iy * P.W + ix;
    const float d = depth[pix];
    if (d <= 0.0f) return;
    const float vvx = (((float)ix) - P.K[2]) / P.K[0];
    const float vvy = (((float)iy) - P.K[5]) / P.K[4];
    const float lambda = sqrtf(madd(vvx, vvx, vvy * vvy) + 1.0f);
    const float norm = sqrtf(madd(cz, cz, madd(cx, cx, cy * cy)));
    const float diff = -madd(1.0f / lambda, norm, -d);
    if (diff < -1.0f * P.trunc) return;
    const float dist = fminf(1.0f, diff / P.trunc);
    const float w_old = w[idx];
    const float w_new = w_old + P.obs_weight;
    float4 v = trgb[idx];
    const float new_t = madd(v.x, w_old, P.obs_weight * dist) / w_new;
    if (P.obs_weight < 0.0f && w_old <= 1.0f) {
        trgb[idx] = make_float4(1.f, 0.f, 0.f, 0.f);
        w[idx] = 0.f;
        return;
    }
    if (new_t > 1.0f) return;
    const float nr = rgb[pix * 3], ng = rgb[pix * 3 + 1], nb = rgb[pix * 3 + 2];
    float4 o;
    o.x = new_t;
    o.y = fminf(madd(v.y, w_old, P.obs_weight * nr) / w_new, 1.0f);
    o.z = fminf(madd(v.z, w_old, P.obs_weight * ng) / w_new, 1.0f);
    o.w = fminf(madd(v.w, w_old, P.obs_weight * nb) / w_new, 1.0f);
    trgb[idx] = o;
    w[idx] = w_new;
}

__global__ __launch_bounds__(256) void gbv_clear_kernel(float4* __restrict__ trgb, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        trgb[i] = make_float4(1.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------- host helpers
static inline int sweep_blocks(int64_t n_items) {
    int64_t b = (n_items + 255) / 256;
    if (b > 256 * 8) b = 256 * 8;   // 8 blocks per CU, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

// rows whose fp32 index decode may differ from the exact one (see oracle/tsdf_oracle.c header).
static void decode_split(int dx, int dy, int dz, int index_decode, int* risky_rows, int* literal_all, int* alias_margin = nullptr) {
    *risky_rows = 0; *literal_all = 0;
    if (alias_margin) *alias_margin = 0;
    if (index_decode == 1) return;                   // exact everywhere
    const int64_t M = (int64_t)dy * dz, N = M * dx;
    if (M >= (1 << 24)) { *literal_all = 1; return; }
    const int64_t margin = (N >> 22) + 64;           // >= 2x the worst fp32 rounding of idx/(dy*dz)
    if (alias_margin) *alias_margin = (int)margin;
    int64_t rr = (margin + dz - 1) / dz;
    if (2 * rr >= dy) { *literal_all = 1; return; }
    *risky_rows = (int)rr;
}

}  // namespace rfx

using namespace rfx;

static inline size_t mv_header_bytes(size_t tiles) { return ((MV_HDR + tiles) * sizeof(unsigned) + 255) / 256 * 256; }

extern "C" {

#ifdef MV_STATS
int rfx_debug_mv_stats(unsigned long long out[16], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rfx::g_mv_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rfx::g_mv_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

#ifdef MV_TIMING
int rfx_debug_mv_times(unsigned long long* out, int n_waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rfx::g_mv_times), sizeof(unsigned long long) * 2 * (size_t)n_waves) == hipSuccess ? 0 : -1;
}
#endif

int rfx_abi_version(void) { return RFX_ABI_VERSION; }
int rfx_last_hip_error(void) { return g_last_hip_error; }

// timing events for a caller without a HIP binding (include/rfx.h, ABI 10)
int rfx_event_create(rfx_event* out) {
    if (!out) return RFX_ERR_ARG;
    hipEvent_t e = nullptr;
    // timing only: no system-scope fence (cache write-back + invalidate) when the event completes -- the fence is what an
    // event costs the work behind it, and nothing here reads device memory from the host after waiting on these events
    RFX_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    *out = e;
    return RFX_OK;
}
int rfx_event_destroy(rfx_event ev) {
    if (!ev) return RFX_ERR_ARG;
    RFX_HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return RFX_OK;
}
int rfx_event_elapsed_ms(rfx_event start, rfx_event stop, float* ms) {
    if (!start || !stop || !ms) return RFX_ERR_ARG;
    // an event that was never recorded is an expected answer here (a stage the phase does not run), not a fault of the process:
    // the runtime's sticky last-error is cleared, or the caller's next unrelated HIP call (torch's) would report it
    hipError_t e = hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop));
    if (e == hipSuccess) e = hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return ::rfx::hip_fail(e);
    }
    return RFX_OK;
}

// workspace = [queue counters + coarse max-depth tiles | A: 8 B per pixel | B: 16 B per pixel | work queue | queue of the rows
// next to x-slab boundaries].  Queue form: A = the {F, G} classification image, B = the exact path's image {depth, 1/lambda,
// packed colour, 0}.  Tile form: A = {depth, 1/lambda}, B = the packed colour (when the caller hands over rgb).  The queue
// is sized for the worst case (every 64-voxel chunk of the volume), so an append can never overflow: 32 B per chunk.
static inline size_t mv_queue_capacity(int dx, int dy, int dz) { return (size_t)dx * dy * (size_t)((dz + 63) / 64); }
constexpr size_t MV_QUEUE_PAD = 16;      // entries past the capacity that mv_chunks_kernel may read (never uses)
// queue of the rows next to x-slab boundaries (they decode the voxel index literally): 2 * risky_rows rows per x
// (risky_rows <= ceil(576 / dz) for any volume below 2^31 voxels: decode_split's margin is at most (2^31 >> 22) + 64;
//  the bound does not depend on the slab a call works on)
static inline size_t mv_risky_capacity(int dx, int dz) {
    const size_t rr_max = (size_t)((576 + dz - 1) / dz);
    return 2 * rr_max * (size_t)dx * (size_t)((dz + 63) / 64);
}
static inline size_t mv_image_bytes(int H, int W) { return ((size_t)H * W * sizeof(float2) + 255) / 256 * 256; }
static inline size_t mv_cpk_bytes(int H, int W) { return ((size_t)H * W * sizeof(float) + 255) / 256 * 256; }

size_t rfx_tsdf_integrate_workspace_bytes(int dx, int dy, int dz, int H, int W) {
    if (H <= 0 || W <= 0 || dx <= 0 || dy <= 0 || dz <= 0) return 0;
    const size_t tiles = (size_t)((H + MV_TD - 1) / MV_TD) * ((W + MV_TD - 1) / MV_TD);
    return mv_header_bytes(tiles) + 3 * mv_image_bytes(H, W) +
           (mv_queue_capacity(dx, dy, dz) + mv_risky_capacity(dx, dz) + 2 * MV_QUEUE_PAD) * sizeof(MvItem);
}

// V1 on the x-planes [x0, x1) of a dx*dy*dz volume; tsdf/weight/color hold ONLY those planes (slab-local arrays).  Every
// voxel is computed with its GLOBAL index and coordinates -- including the reference's fp32 index decode, which depends on
// the global linear index -- so the slabs of a volume, integrated separately (one per GPU), are bit-identical to the whole.
// Exactly one of color_packed / rgb255 is given: rgb255 ([H*W,3], 0..255 valued) is packed by the frame kernel.
static int integrate_slab(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1,
                          const float origin[3], float voxel, const float K[9], const float c2w[16],
                          const float* color_packed, const float* rgb255, const float* depth, int H, int W,
                          float trunc, float obs_weight, int weight_clamp, int reintegrate,
                          const float old_bnd[6], int index_decode,
                          void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (!tsdf || !weight || !color || !origin || !K || !c2w || !depth) return RFX_ERR_ARG;
    if ((color_packed == nullptr) == (rgb255 == nullptr)) return RFX_ERR_ARG;
    if (dx <= 0 || dy <= 0 || dz <= 0 || H <= 0 || W <= 0 || !(voxel > 0.0f)) return RFX_ERR_ARG;
    if ((int64_t)H * W >= (1LL << 28)) return RFX_ERR_UNSUPPORTED;          // image byte offsets are 32-bit
    if (x0 < 0 || x1 > dx || x1 <= x0) return RFX_ERR_ARG;
    if (reintegrate && !old_bnd) return RFX_ERR_ARG;
    if ((int64_t)dx * dy * dz >= (1LL << 31)) return RFX_ERR_UNSUPPORTED;   // the reference indexes with int32
    if (!workspace || workspace_bytes < rfx_tsdf_integrate_workspace_bytes(x1 - x0, dy, dz, H, W) || ((uintptr_t)workspace & 15)) return RFX_ERR_WORKSPACE;
    if (K[0] == 0.0f || K[4] == 0.0f) return RFX_ERR_ARG;
    const bool whole = x0 == 0 && x1 == dx;
    const int64_t skip = (int64_t)x0 * dy * dz;       // global index of the slab's first voxel

    MvParams P;
    for (int i = 0; i < 9; ++i) P.K[i] = K[i];
    for (int i = 0; i < 16; ++i) P.c2w[i] = c2w[i];
    for (int i = 0; i < 3; ++i) P.origin[i] = (float)(int)origin[i];
    P.voxel = voxel; P.dx = dx; P.dy = dy; P.dz = dz; P.H = H; P.W = W;
    P.trunc = trunc; P.obs_weight = obs_weight; P.weight_clamp = weight_clamp ? 1 : 0;
    P.reintegrate = reintegrate ? 1 : 0;
    for (int i = 0; i < 6; ++i) P.old_bnd[i] = old_bnd ? old_bnd[i] : 0.0f;
    decode_split(dx, dy, dz, index_decode, &P.risky_rows, &P.literal_all, &P.alias_margin);
    {   // |cam_norm/(lambda*cam_z) - 1| <= e: pixel rounding moves the ray by <= half a pixel
        const float mvx = fmaxf(fabsf(K[2]), fabsf((float)(W - 1) - K[2])) / fabsf(K[0]);
        const float mvy = fmaxf(fabsf(K[5]), fabsf((float)(H - 1) - K[5])) / fabsf(K[4]);
        const float hx = 0.5f / fabsf(K[0]), hy = 0.5f / fabsf(K[4]);
        P.ratio_eps = fminf(0.5f, (mvx + hx) * hx + (mvy + hy) * hy + 1e-5f);
    }
    {   // |u~ - u| of the fast path's projection u~ = fma(fx, cx * rcp(cz), cx0) against the reference's fma(fx, cx / cz, cx0):
        // the quotients differ by <= 2 ulp (1-ulp reciprocal, one rounding), i.e. by <= 2.4e-7 |q|, and |fx q| <= |u| + |cx0|;
        // the two fma roundings add <= one ulp of |u| each.  In the image (|u| <= W + 1) that is <= 5e-7 (W + |cx0| + 1);
        // taken 8x larger.  Coordinates far outside the image are not near any in-image rounding boundary that matters.
        const float su = (float)W + fabsf(K[2]) + 1.0f, sv = (float)H + fabsf(K[5]) + 1.0f;
        P.edge_eps = fminf(0.25f, 4e-6f * fmaxf(su, sv) + 1e-4f);
    }

    // world z-axis in camera coordinates = third row of R; its image-space direction picks the layout
    P.dimg_colmajor = (fabsf(K[4] * c2w[9]) >= fabsf(K[0] * c2w[8])) ? 1 : 0;
    hipStream_t st = as_stream(stream);
    unsigned* dmax_bits = reinterpret_cast<unsigned*>(workspace);
    const size_t n_tiles = (size_t)((H + MV_TD - 1) / MV_TD) * ((W + MV_TD - 1) / MV_TD);
    float2* img_a = reinterpret_cast<float2*>(reinterpret_cast<char*>(workspace) + mv_header_bytes(n_tiles));
    float4* img_b = reinterpret_cast<float4*>(reinterpret_cast<char*>(img_a) + mv_image_bytes(H, W));
    char* after_images = reinterpret_cast<char*>(img_b) + 2 * mv_image_bytes(H, W);
#ifndef MV_NO_QUEUE
    const bool queue_form = !P.literal_all && dx <= MV_Q_MAX_DIM && dy <= MV_Q_MAX_DIM && dz <= 64 * 65535;
#else
    const bool queue_form = false;
#endif
    float2* dimg = img_a;                                   // tile form
    float* cpk_ws = reinterpret_cast<float*>(img_b);
    const float* cpk = color_packed ? color_packed : cpk_ws;
    float2* fgimg = img_a;                                  // queue form
    float4* nimg = img_b;
    hipLaunchKernelGGL(mv_frame_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, depth, rgb255, color_packed, dimg, fgimg,
                       queue_form ? nimg : (float4*)nullptr, cpk_ws, dmax_bits, H, W, K[0], K[4], K[2], K[5], trunc, P.dimg_colmajor);
    RFX_LAUNCH_CHECK();
    constexpr int TX = MV_TX, TY = MV_TY, U = MV_U;
    // Window of tiles the view frustum can touch: the frustum is convex, so its (x,y) footprint lies in
    // the bounding box of the apex and the four corner rays pushed past the far side of the volume.
    // Rows that decode literally may alias to other cells, so the window is only used when none do.
    int tx0 = 0, ty0 = 0, tx1 = (dx + TX - 1) / TX, ty1 = (dy + TY - 1) / TY;
    if (!P.literal_all) {
        const float ext = voxel * sqrtf((float)dx * dx + (float)dy * dy + (float)dz * dz);
        const float far = 2.0f * ext + 2.0f * (fabsf(c2w[3] - origin[0]) + fabsf(c2w[7] - origin[1]) + fabsf(c2w[11] - origin[2]));
        float lo[2] = {c2w[3], c2w[7]}, hi[2] = {c2w[3], c2w[7]};
        for (int cidx = 0; cidx < 4; ++cidx) {
            const float u = (cidx & 1) ? (float)W + 1.0f : -2.0f, v = (cidx & 2) ? (float)H + 1.0f : -2.0f;
            const float rx = (u - K[2]) / K[0], ry = (v - K[5]) / K[4];
            for (int a = 0; a < 2; ++a) {
                const float wv = c2w[4 * a + 3] + far * (c2w[4 * a + 0] * rx + c2w[4 * a + 1] * ry + c2w[4 * a + 2]);
                lo[a] = fminf(lo[a], wv); hi[a] = fmaxf(hi[a], wv);
            }
        }
        const float pad = 2.0f * voxel;
        const int vx0 = (int)floorf((lo[0] - pad - P.origin[0]) / voxel), vx1 = (int)ceilf((hi[0] + pad - P.origin[0]) / voxel);
        const int vy0 = (int)floorf((lo[1] - pad - P.origin[1]) / voxel), vy1 = (int)ceilf((hi[1] + pad - P.origin[1]) / voxel);
        tx0 = std::max(0, std::min(tx1, vx0 / TX)); ty0 = std::max(0, std::min(ty1, vy0 / TY));
        tx1 = std::max(tx0, std::min(tx1, vx1 / TX + 1)); ty1 = std::max(ty0, std::min(ty1, vy1 / TY + 1));
    }
    const int all_tx = (dx + TX - 1) / TX, all_ty = (dy + TY - 1) / TY;
    for (int i = 0; i < 3; ++i) { P.win_x0[i] = P.win_y0[i] = P.win_wx[i] = P.win_wy[i] = 0; }
    if (queue_form) {
        // queue form: windows in ROWS.  [0] = the frustum footprint without the risky boundary rows, [1],[2] = those rows
        // (every x: their alias zones are walked whatever the view); they go to a queue of their own, whose body
        // carries the literal index decode.
        int rx0 = std::min(dx, tx0 * TX), rx1 = std::min(dx, tx1 * TX), ry0 = std::min(dy, ty0 * TY), ry1 = std::min(dy, ty1 * TY);
        rx0 = std::max(rx0, x0); rx1 = std::min(rx1, x1);              // the slab
        if (rx1 < rx0) rx1 = rx0;
        if (P.risky_rows > 0) {
            P.win_x0[1] = x0; P.win_y0[1] = 0; P.win_wx[1] = x1 - x0; P.win_wy[1] = P.risky_rows;
            P.win_x0[2] = x0; P.win_y0[2] = dy - P.risky_rows; P.win_wx[2] = x1 - x0; P.win_wy[2] = P.risky_rows;
            ry0 = std::max(ry0, P.risky_rows); ry1 = std::min(ry1, dy - P.risky_rows);
            if (ry1 < ry0) ry1 = ry0;
        }
        P.win_x0[0] = rx0; P.win_y0[0] = ry0; P.win_wx[0] = rx1 - rx0; P.win_wy[0] = ry1 - ry0;
        const int64_t rows_main = (int64_t)P.win_wx[0] * P.win_wy[0];
        const int64_t rows_risky = (int64_t)P.win_wx[1] * P.win_wy[1] + (int64_t)P.win_wx[2] * P.win_wy[2];
        if (rows_main + rows_risky == 0) return RFX_OK;
        const int nch = (dz + 63) / 64;
        const size_t cap = mv_queue_capacity(x1 - x0, dy, dz), cap_risky = mv_risky_capacity(x1 - x0, dz);
        MvItem* queue = reinterpret_cast<MvItem*>(after_images);
        MvItem* queue_risky = queue + cap + MV_QUEUE_PAD;
        const unsigned q_cap = (unsigned)std::min<size_t>(cap, 0xffffffffu), q_cap_risky = (unsigned)std::min<size_t>(cap_risky, 0xffffffffu);
        const int blocks_main = (int)((rows_main + MV_ROWS_THREADS - 1) / MV_ROWS_THREADS), blocks_risky = (int)((rows_risky + MV_ROWS_THREADS - 1) / MV_ROWS_THREADS);
        hipLaunchKernelGGL(mv_rows_kernel, dim3((unsigned)(blocks_main + blocks_risky)), dim3(MV_ROWS_THREADS), 0, st, P, dmax_bits, dmax_bits, queue,
                           q_cap, queue_risky, q_cap_risky, blocks_main, skip);
        RFX_LAUNCH_CHECK();
        // a grid of RESIDENT blocks pulls from the queue (a second, partial round of blocks would run at a fraction of the
        // chip); small volumes need fewer.  Three instances: the fast body (every mapping frame), the generic one, and
        // the generic one with the re-integration window test.
        using Kern = void (*)(MvParams, const unsigned*, const MvItem*, unsigned, const MvItem*, unsigned, unsigned, int64_t, const float4*,
                              const float2*, float*, float*, float*);
        const int variant = P.reintegrate ? 2 : (obs_weight > 0.0f && trunc > 0.0f && (int64_t)H * W * 8 < (1 << 24) ? 0 : 1);
        const Kern kern = variant == 2 ? (Kern)mv_chunks_kernel<U, false, true>
                        : variant == 1 ? (Kern)mv_chunks_kernel<U, false, false> : (Kern)mv_chunks_kernel<U, true, false>;
        static int resident[3] = {0, 0, 0};       // blocks per CU x CUs, per variant; benign if raced (same value)
        if (!resident[variant]) {
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop;
            RFX_HIP_TRY(hipGetDevice(&dev));
            RFX_HIP_TRY(hipGetDeviceProperties(&prop, dev));
            RFX_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, 0));
#ifdef MV_BLOCKS_PER_CU      // timing builds: the sensitivity of the kernel to its occupancy
            per_cu = std::min(per_cu, MV_BLOCKS_PER_CU);
#endif
            resident[variant] = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
        }
        int blocks_risky_q = 0, blocks_main_q = 0;
        if (rows_risky > 0) {
            const int64_t max_items = std::min<int64_t>((int64_t)cap_risky, rows_risky * nch);
            blocks_risky_q = (int)std::max<int64_t>(1, std::min<int64_t>(resident[variant] / 8, (max_items + 4 * U - 1) / (4 * U)));
        }
        if (rows_main > 0) {
            const int64_t max_items = std::min<int64_t>((int64_t)cap, rows_main * nch);
            blocks_main_q = (int)std::max<int64_t>(1, std::min<int64_t>(resident[variant] - blocks_risky_q, (max_items + 4 * U - 1) / (4 * U)));
        }
        hipLaunchKernelGGL(kern, dim3(blocks_main_q + blocks_risky_q), dim3(256), 0, st, P, dmax_bits, queue, q_cap, queue_risky, q_cap_risky,
                           (unsigned)blocks_main_q, skip, nimg, fgimg, tsdf, weight, color);
        RFX_LAUNCH_CHECK();
        return RFX_OK;
    }
    if (!whole) return RFX_ERR_UNSUPPORTED;      // the tile-form fallback (literal decode everywhere) is not slab-aware
    if (P.risky_rows > 0 && !P.literal_all) {
        // boundary strips (all x): tile rows [0, s) and [all_ty - s, all_ty); the main window is clipped to what is left
        const int s_rows = std::min((P.risky_rows + TY - 1) / TY, all_ty / 2);
        P.win_x0[1] = 0; P.win_y0[1] = 0; P.win_wx[1] = all_tx; P.win_wy[1] = s_rows;
        P.win_x0[2] = 0; P.win_y0[2] = all_ty - s_rows; P.win_wx[2] = all_tx; P.win_wy[2] = s_rows;
        ty0 = std::max(ty0, s_rows); ty1 = std::min(ty1, all_ty - s_rows);
        if (ty1 < ty0) ty1 = ty0;
    }
    P.win_x0[0] = tx0; P.win_y0[0] = ty0; P.win_wx[0] = tx1 - tx0; P.win_wy[0] = ty1 - ty0;
    int64_t tiles = 0;
    for (int i = 0; i < 3; ++i) tiles += (int64_t)P.win_wx[i] * P.win_wy[i];
    if (tiles == 0) return RFX_OK;
    const int blocks = (int)((tiles + 3) / 4);
    hipLaunchKernelGGL((mv_integrate_kernel<TX, TY, U>), dim3(blocks), dim3(256), 0, st, P, dimg, dmax_bits,
                       cpk, tsdf, weight, color);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_integrate(float* tsdf, float* weight, float* color, int dx, int dy, int dz,
                       const float origin[3], float voxel, const float K[9], const float c2w[16],
                       const float* color_packed, const float* depth, int H, int W,
                       float trunc, float obs_weight, int weight_clamp, int reintegrate,
                       const float old_bnd[6], int index_decode,
                       void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (!color_packed) return RFX_ERR_ARG;
    return integrate_slab(tsdf, weight, color, dx, dy, dz, 0, dx, origin, voxel, K, c2w, color_packed, nullptr, depth, H, W, trunc,
                          obs_weight, weight_clamp, reintegrate, old_bnd, index_decode, workspace, workspace_bytes, stream);
}

int rfx_tsdf_integrate_slab(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1,
                            const float origin[3], float voxel, const float K[9], const float c2w[16],
                            const float* color_packed, const float* depth, int H, int W,
                            float trunc, float obs_weight, int weight_clamp, int reintegrate,
                            const float old_bnd[6], int index_decode,
                            void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (!color_packed) return RFX_ERR_ARG;
    return integrate_slab(tsdf, weight, color, dx, dy, dz, x0, x1, origin, voxel, K, c2w, color_packed, nullptr, depth, H, W, trunc,
                          obs_weight, weight_clamp, reintegrate, old_bnd, index_decode, workspace, workspace_bytes, stream);
}

int rfx_tsdf_integrate_rgb(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1,
                           const float origin[3], float voxel, const float K[9], const float c2w[16],
                           const float* rgb255, const float* depth, int H, int W,
                           float trunc, float obs_weight, int weight_clamp, int reintegrate,
                           const float old_bnd[6], int index_decode,
                           void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (!rgb255) return RFX_ERR_ARG;
    return integrate_slab(tsdf, weight, color, dx, dy, dz, x0, x1, origin, voxel, K, c2w, nullptr, rgb255, depth, H, W, trunc,
                          obs_weight, weight_clamp, reintegrate, old_bnd, index_decode, workspace, workspace_bytes, stream);
}

int rfx_pack_color(const float* rgb255, float* packed, int64_t n, rfx_stream stream) {
    if (!rgb255 || !packed || n < 0) return RFX_ERR_ARG;
    if (n == 0) return RFX_OK;
    hipLaunchKernelGGL(pack_color_kernel, dim3(sweep_blocks(n)), dim3(256), 0, as_stream(stream), rgb255, packed, n);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_fill(float* tsdf, float* weight, float* color, int64_t n, rfx_stream stream) {
    if (!tsdf || !weight || !color || n < 0) return RFX_ERR_ARG;
    if (((uintptr_t)tsdf | (uintptr_t)weight | (uintptr_t)color) & 15) return RFX_ERR_ARG;
    if (n == 0) return RFX_OK;
    hipLaunchKernelGGL(mv_fill_kernel, dim3(sweep_blocks(n >> 2)), dim3(256), 0, as_stream(stream), tsdf, weight, color, n);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_copy(const float* tsdf, const float* weight, const float* color,
                  float* tsdf_back, float* weight_back, float* color_back, int64_t n, rfx_stream stream) {
    if (!tsdf || !weight || !color || !tsdf_back || !weight_back || !color_back || n < 0) return RFX_ERR_ARG;
    if (((uintptr_t)tsdf | (uintptr_t)weight | (uintptr_t)color | (uintptr_t)tsdf_back | (uintptr_t)weight_back |
         (uintptr_t)color_back) & 15) return RFX_ERR_ARG;
    if (n == 0) return RFX_OK;
    hipLaunchKernelGGL(mv_copy_kernel, dim3(sweep_blocks(n >> 2)), dim3(256), 0, as_stream(stream), tsdf, weight,
                       color, tsdf_back, weight_back, color_back, n);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

// V2 for the x-planes [x0, x1) of the new volume, reading x-planes [ox_a, ox_b) of the old one (both array sets hold
// only those planes); everything else as rfx_tsdf_shift: global coordinates, global index decode.
static int shift_slab(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1, const float origin[3],
                      const float* old_tsdf, const float* old_weight, const float* old_color,
                      int odx, int ody, int odz, int ox_a, int ox_b, const float old_origin[3], float voxel,
                      int index_decode, rfx_stream stream) {
    if (!tsdf || !weight || !color || !origin || !old_origin) return RFX_ERR_ARG;
    if (dx <= 0 || dy <= 0 || dz <= 0 || odx <= 0 || ody <= 0 || odz <= 0 || !(voxel > 0.0f)) return RFX_ERR_ARG;
    if (x0 < 0 || x1 > dx || x1 <= x0 || ox_a < 0 || ox_b > odx || ox_b < ox_a) return RFX_ERR_ARG;
    if (ox_b > ox_a && (!old_tsdf || !old_weight || !old_color)) return RFX_ERR_ARG;
    if ((int64_t)dx * dy * dz >= (1LL << 31)) return RFX_ERR_UNSUPPORTED;
    ShiftParams P;
    P.dx = dx; P.dy = dy; P.dz = dz; P.odx = odx; P.ody = ody; P.odz = odz; P.voxel = voxel;
    P.x0 = x0; P.ox_a = ox_a; P.ox_b = ox_b;
    for (int i = 0; i < 3; ++i) { P.origin[i] = origin[i]; P.old_origin[i] = old_origin[i]; }
    decode_split(dx, dy, dz, index_decode, &P.risky_rows, &P.literal_all);
    hipLaunchKernelGGL(mv_shift_kernel, dim3((unsigned)((int64_t)(x1 - x0) * dy)), dim3(256), 0, as_stream(stream), P, tsdf,
                       weight, color, old_tsdf, old_weight, old_color);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_shift(float* tsdf, float* weight, float* color, int dx, int dy, int dz, const float origin[3],
                   const float* old_tsdf, const float* old_weight, const float* old_color,
                   int odx, int ody, int odz, const float old_origin[3], float voxel,
                   int index_decode, rfx_stream stream) {
    if (!old_tsdf || !old_weight || !old_color) return RFX_ERR_ARG;
    return shift_slab(tsdf, weight, color, dx, dy, dz, 0, dx, origin, old_tsdf, old_weight, old_color, odx, ody, odz, 0, odx, old_origin,
                      voxel, index_decode, stream);
}

int rfx_tsdf_shift_slab(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1, const float origin[3],
                        const float* old_tsdf, const float* old_weight, const float* old_color,
                        int odx, int ody, int odz, int ox_a, int ox_b, const float old_origin[3], float voxel,
                        int index_decode, rfx_stream stream) {
    return shift_slab(tsdf, weight, color, dx, dy, dz, x0, x1, origin, old_tsdf, old_weight, old_color, odx, ody, odz, ox_a, ox_b,
                      old_origin, voxel, index_decode, stream);
}

// old x-planes a slab of the new volume reads (host; same arithmetic as mv_shift_kernel, one plane of slack either side,
// clipped to the old volume): [*a, *b), empty when the slab lies outside the old volume
int rfx_tsdf_shift_source_planes(int x0, int x1, const float origin[3], int odx, const float old_origin[3], float voxel, int* a, int* b) {
    if (!origin || !old_origin || !a || !b || x1 <= x0 || !(voxel > 0.0f)) return RFX_ERR_ARG;
    const float w0 = fmaf((float)x0, voxel, origin[0]), w1 = fmaf((float)(x1 - 1), voxel, origin[0]);
    const long lo = lroundf((w0 - old_origin[0]) / voxel) - 1, hi = lroundf((w1 - old_origin[0]) / voxel) + 2;
    *a = (int)std::max<long>(0, std::min<long>(odx, lo));
    *b = (int)std::max<long>(*a, std::min<long>(odx, hi));
    return RFX_OK;
}

int rfx_tsdf_trilerp(const float* tsdf, const float* weight, const float* color, int dx, int dy, int dz,
                     const float origin[3], float voxel, const float* pts, int64_t n, float* out5,
                     rfx_stream stream) {
    (void)weight;
    if (!tsdf || !color || !origin || !pts || !out5 || n < 0) return RFX_ERR_ARG;
    if (dx <= 0 || dy <= 0 || dz <= 0 || !(voxel > 0.0f)) return RFX_ERR_ARG;
    if (n == 0) return RFX_OK;
    VolView V; V.dx = dx; V.dy = dy; V.dz = dz; V.voxel = voxel;
    for (int i = 0; i < 3; ++i) V.origin[i] = origin[i];
    hipLaunchKernelGGL(mv_trilerp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), V, 0, dx, tsdf,
                       color, (const float*)nullptr, (const float*)nullptr, pts, n, out5, (unsigned char*)nullptr);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_trilerp_slab(const float* tsdf, const float* color, int dx, int dy, int dz, int x0, int x1,
                          const float* tsdf_halo, const float* color_halo, const float origin[3], float voxel,
                          const float* pts, int64_t n, float* out5, uint8_t* inside, rfx_stream stream) {
    if (!tsdf || !color || !origin || !pts || !out5 || !inside || n < 0) return RFX_ERR_ARG;
    if (dx <= 0 || dy <= 0 || dz <= 0 || !(voxel > 0.0f) || x0 < 0 || x1 > dx || x1 <= x0) return RFX_ERR_ARG;
    if (x1 < dx && (!tsdf_halo || !color_halo)) return RFX_ERR_ARG;          // every slab but the last needs plane x1
    if (n == 0) return RFX_OK;
    VolView V; V.dx = dx; V.dy = dy; V.dz = dz; V.voxel = voxel;
    for (int i = 0; i < 3; ++i) V.origin[i] = origin[i];
    hipLaunchKernelGGL(mv_trilerp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), V, x0, x1, tsdf,
                       color, tsdf_halo, color_halo, pts, n, out5, inside);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_filter(float* tsdf, float* weight, float* color, int64_t n, float weight_threshold, rfx_stream stream) {
    if (!tsdf || !weight || !color || n < 0) return RFX_ERR_ARG;
    if (n == 0) return RFX_OK;
    // `float weight_threshold=(int) other_params[0]` (Volume.py:468)
    hipLaunchKernelGGL(mv_filter_kernel, dim3(sweep_blocks(n)), dim3(256), 0, as_stream(stream), tsdf, weight, color, n,
                       (float)(int)weight_threshold);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_truncated_pc_slab(const float* tsdf, const float* color, int dx, int dy, int dz, int x0, int x1,
                               const float origin[3], float voxel, float trunc, int pc_num, float trunc_tsdf,
                               float* pc7, uint32_t* count, uint8_t* hit, int index_decode, rfx_stream stream) {
    if (!tsdf || !color || !origin || !pc7 || !count || pc_num <= 0) return RFX_ERR_ARG;
    if (dx <= 0 || dy <= 0 || dz <= 0 || x0 < 0 || x1 > dx || x1 < x0) return RFX_ERR_ARG;
    if ((int64_t)dx * dy * dz >= (1LL << 31)) return RFX_ERR_UNSUPPORTED;
    VolView V; V.dx = dx; V.dy = dy; V.dz = dz; V.voxel = voxel;
    for (int i = 0; i < 3; ++i) V.origin[i] = origin[i];
    int rr, la;
    decode_split(dx, dy, dz, index_decode, &rr, &la);
    const int64_t plane = (int64_t)dy * dz;
    hipLaunchKernelGGL(mv_truncated_pc_kernel, dim3((unsigned)((pc_num + 255) / 256)), dim3(256), 0, as_stream(stream),
                       V, tsdf, color, trunc, pc_num, trunc_tsdf, pc7, count, rr, la, x0 * plane, x1 * plane, hit);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_tsdf_truncated_pc(const float* tsdf, const float* color, int dx, int dy, int dz,
                          const float origin[3], float voxel, float trunc, int pc_num, float trunc_tsdf,
                          float* pc7, uint32_t* count, int index_decode, rfx_stream stream) {
    return rfx_tsdf_truncated_pc_slab(tsdf, color, dx, dy, dz, 0, dx, origin, voxel, trunc, pc_num, trunc_tsdf, pc7, count, nullptr,
                                      index_decode, stream);
}

int rfx_gbv_integrate(float* trgb, float* w, int res, const float box[6], const float K[9],
                      const float* c2w_dev, const float* rgb01, const float* depth, int H, int W,
                      float trunc, float obs_weight, rfx_stream stream) {
    if (!trgb || !w || !box || !K || !c2w_dev || !rgb01 || !depth) return RFX_ERR_ARG;
    if (res <= 0 || H <= 0 || W <= 0) return RFX_ERR_ARG;
    if ((int64_t)res * res * res >= (1LL << 31)) return RFX_ERR_UNSUPPORTED;
    if ((uintptr_t)trgb & 15) return RFX_ERR_ARG;
    GbvParams P;
    for (int i = 0; i < 9; ++i) P.K[i] = K[i];
    for (int i = 0; i < 6; ++i) P.box[i] = box[i];
    P.res = res; P.H = H; P.W = W; P.voxel_size = 1.0f / (float)res;   // mapper.py:225 (python float -> fp32)
    P.trunc = trunc; P.obs_weight = obs_weight;
    const int64_t n = (int64_t)res * res * res;
    hipLaunchKernelGGL(gbv_integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), P,
                       c2w_dev, reinterpret_cast<float4*>(trgb), w, rgb01, depth);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_gbv_clear(float* trgb, int64_t n_voxels, rfx_stream stream) {
    if (!trgb || n_voxels < 0) return RFX_ERR_ARG;
    if ((uintptr_t)trgb & 15) return RFX_ERR_ARG;
    if (n_voxels == 0) return RFX_OK;
    hipLaunchKernelGGL(gbv_clear_kernel, dim3(sweep_blocks(n_voxels)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float4*>(trgb), n_voxels);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"
