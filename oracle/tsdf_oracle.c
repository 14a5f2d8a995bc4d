/*
 * oracle/tsdf_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle + cpu_baseline).
 *
 * Plain-C, single-thread restatement of the reference's PyCUDA kernel *text* for the
 * moving TSDF volume (MV) and the global explicit volume (GBV).  Nothing under
 * remixfusion_amd/ may import, link or call this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, and only as the checker.
 *
 * Each function cites the reference lines it follows (paths under /root/reference):
 *   orc_mv_integrate      model/Volume.py:196-336   (kernel `integrate`),  host :713-757
 *   orc_mv_shift          model/Volume.py:128-194   (kernel `swap_rot_trans`), host :796-855
 *   orc_mv_trilerp        model/Volume.py:337-458   (kernel `tri_intepolate`), host :760-794
 *   orc_mv_filter         model/Volume.py:462-487   (kernel `filter_tsdf`)
 *   orc_mv_truncated_pc   model/Volume.py:489-559   (kernel `get_truncated_pc`)
 *   orc_mv_fill           model/Volume.py:561-583   (kernel `clean_tsdf`)
 *   orc_mv_copy           model/Volume.py:585-610   (kernel `copy_volume`)
 *   orc_gbv_integrate     mp_slam/mapper.py:37-158  (kernel `integrate`),  host :823-872
 *   orc_gbv_clear         mp_slam/mapper.py:161-183 (kernel `clean_tsdf`)
 *
 * PARITY STATUS: the reference ships no golden vectors, tests or CPU path for these kernels
 *  and PyCUDA cannot run here, so this oracle is pinned only by
 * analytic known-answer tests (tests/test_oracle_tsdf.py) -> "parity unpinned" against a
 * live run of the reference.  Two things the kernel text does not determine are modelled
 * explicitly and switchable at compile time:
 *
 *   ORC_FMA (default 1): nvcc compiles the PyCUDA strings with -fmad=true, contracting
 *     a*b+c into fma.  We model LLVM's left-first rule: (a*b)+(c*d) -> fma(a,b,c*d);
 *     x+(a*b) -> fma(a,b,x); (a*b)-c -> fma(a,b,-c).  ORC_FMA=0 builds the uncontracted
 *     variant; tests bound the difference between the two (it only moves round-off).
 *     This file must be compiled with -ffp-contract=off so that *only* the explicit
 *     fmaf() calls below fuse.
 *
 *   index decode: the kernels recover (x,y,z) from the linear index with fp32 divisions
 *     (Volume.py:224-226, mapper.py:73-75).  For N > 2^24 voxels (float)idx is inexact and a
 *     few voxels next to a slab boundary decode to a neighbouring (x, y=-1, z) cell.  That
 *     is what the reference executes, so decode_mode=0 ("reference") reproduces it
 *     literally; decode_mode=1 ("exact") uses integer arithmetic.
 *
 * Deliberate deviations: the reference's `voxel_idx > N` guards are
 * off by one (thread idx==N touches memory past the arrays); loops here run idx in [0,N).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#ifndef ORC_FMA
#define ORC_FMA 1
#endif

#if ORC_FMA
#define MADD(a, b, c) fmaf((a), (b), (c))          /* a*b + c, one rounding */
#else
static inline float orc_madd(float a, float b, float c) { float p = a * b; return p + c; }
#define MADD(a, b, c) orc_madd((a), (b), (c))
#endif

/* CUDA __float2int_rn: round-half-even (Volume.py:261-262).  Default rounding mode is RNE. */
static inline int f2i_rn(float v) { return (int)rintf(v); }

/* ---- index decode, z fastest (Volume.py:224-226) --------------------------------------- */
static inline void mv_decode(int64_t idx, int dx, int dy, int dz, int mode,
                             float* vx, float* vy, float* vz) {
    (void)dx;
    if (mode == 1) {
        int64_t x = idx / ((int64_t)dy * dz);
        int64_t r = idx - x * dy * dz;
        int64_t y = r / dz;
        *vx = (float)x; *vy = (float)y; *vz = (float)(r - y * dz);
        return;
    }
    /* literal: the reference's voxel_idx is a 32-bit int; (dy*dz) is an int product */
    int vi = (int)idx;
    float fx = floorf(((float)vi) / ((float)(dy * dz)));
    float fy = floorf(((float)(vi - ((int)fx) * dy * dz)) / ((float)dz));
    float fz = (float)(vi - ((int)fx) * dy * dz - ((int)fy) * dz);
    *vx = fx; *vy = fy; *vz = fz;
}

/* world->camera with c2w row-major [16] (Volume.py:251-256, mapper.py:83-88) */
static inline void to_cam(const float* c2w, float px, float py, float pz,
                          float* cx, float* cy, float* cz) {
    float tx = px - c2w[0 * 4 + 3];
    float ty = py - c2w[1 * 4 + 3];
    float tz = pz - c2w[2 * 4 + 3];
    /* a*tx + b*ty + c*tz  ==  fma(c,tz, fma(a,tx, b*ty))  under the left-first rule */
    *cx = MADD(c2w[2 * 4 + 0], tz, MADD(c2w[0 * 4 + 0], tx, c2w[1 * 4 + 0] * ty));
    *cy = MADD(c2w[2 * 4 + 1], tz, MADD(c2w[0 * 4 + 1], tx, c2w[1 * 4 + 1] * ty));
    *cz = MADD(c2w[2 * 4 + 2], tz, MADD(c2w[0 * 4 + 2], tx, c2w[1 * 4 + 2] * ty));
}

/* projective sdf for a camera-space point that already passed cam_z>0
 * (Volume.py:261-285 == mapper.py:95-113).  Returns 0 if the voxel is skipped. */
static inline int project_sdf(const float* K, float cx, float cy, float cz, int H, int W,
                              const float* depth, int* pix, float* sdf) {
    int px = f2i_rn(MADD(K[0], (cx / cz), K[2]));
    int py = f2i_rn(MADD(K[4], (cy / cz), K[5]));
    if (px < 0 || px >= W || py < 0 || py >= H) return 0;
    float d = depth[py * W + px];
    if (d <= 0) return 0;
    float vx = (((float)px) - K[2]) / K[0];
    float vy = (((float)py) - K[5]) / K[4];
    float lambda = sqrtf(MADD(vx, vx, vy * vy) + 1.0f);
    float norm = sqrtf(MADD(cz, cz, MADD(cx, cx, cy * cy)));
    /* (-1.f) * ((1.f/lambda)*cam_norm - depth) */
    *sdf = (-1.0f) * MADD((1.0f / lambda), norm, -d);
    *pix = py * W + px;
    return 1;
}

/* counters for the algorithmic-bytes figure of SURVEY.md 8(d): U updated, C colour band */
typedef struct { int64_t updated; int64_t colour; } orc_counts;

/* V1 -- model/Volume.py:196-336.  color_packed = floor(B*65536+G*256+R) (host :728). */
/* voxels idx0 <= idx < idx1 (one thread per voxel in the reference: any split of the index range gives the same
 * volume, which is how bench.py's cpu_baseline spreads the sweep over the host's cores) */
void orc_mv_integrate_range(float* tsdf, float* weight, float* color,
                            int dx, int dy, int dz, const float* origin, float voxel,
                            const float* K, const float* c2w,
                            const float* color_packed, const float* depth, int H, int W,
                            float trunc, float obs_weight, float weight_clamp, float reintegrate,
                            const float* old_bnd, int decode_mode, int64_t idx0, int64_t idx1, orc_counts* counts) {
    int64_t n = (int64_t)dx * dy * dz;
    if (idx0 < 0) idx0 = 0;
    if (idx1 > n) idx1 = n;
    /* int origin_x = vol_origin[0];  -- C truncation toward zero (:230-232) */
    int ox = (int)origin[0], oy = (int)origin[1], oz = (int)origin[2];
    int64_t nu = 0, nc = 0;
    for (int64_t idx = idx0; idx < idx1; ++idx) {
        float vx, vy, vz;
        mv_decode(idx, dx, dy, dz, decode_mode, &vx, &vy, &vz);
        float px = MADD(vx, voxel, (float)ox);
        float py = MADD(vy, voxel, (float)oy);
        float pz = MADD(vz, voxel, (float)oz);
        if (reintegrate == 1) {
            if (px < old_bnd[0] || px >= old_bnd[1] || py < old_bnd[2] || py >= old_bnd[3] ||
                pz < old_bnd[4] || pz >= old_bnd[5]) continue;
        }
        float cx, cy, cz;
        to_cam(c2w, px, py, pz, &cx, &cy, &cz);
        if (cz <= 0) continue;
        int pix; float sdf;
        if (!project_sdf(K, cx, cy, cz, H, W, depth, &pix, &sdf)) continue;
        if (sdf >= -trunc) {
            float dist = fminf(1.0f, sdf / trunc);
            float cur = tsdf[idx];
            float w_old = weight[idx];
            float w_new = w_old + obs_weight;
            float new_tsdf = MADD(cur, w_old, obs_weight * dist) / w_new;
            float new_w = w_new;
            if (weight_clamp == 1.0f) {
                new_w = fminf(w_new, 128.0f);
                if (new_w > 40) new_w = 40;
            }
            tsdf[idx] = new_tsdf;
            weight[idx] = new_w;
            ++nu;
            if (sdf >= -trunc && sdf <= trunc) {
                float nc_ = color_packed[pix];
                float nb = floorf(nc_ / (256 * 256));
                float ng = floorf((nc_ - nb * 256 * 256) / 256);
                float nr = nc_ - nb * 256 * 256 - ng * 256;
                float oc = color[idx];
                float ob = floorf(oc / (256 * 256));
                float og = floorf((oc - ob * 256 * 256) / 256);
                float or_ = oc - ob * 256 * 256 - og * 256;
                nb = fminf(roundf(MADD(ob, w_old, obs_weight * nb) / w_new), 255.0f);
                ng = fminf(roundf(MADD(og, w_old, obs_weight * ng) / w_new), 255.0f);
                nr = fminf(roundf(MADD(or_, w_old, obs_weight * nr) / w_new), 255.0f);
                color[idx] = nb * 256 * 256 + ng * 256 + nr;
                ++nc;
            }
            if (obs_weight == -1.0f && w_old <= 1 && reintegrate == 1.0f) {
                tsdf[idx] = 1.0f; weight[idx] = 0; color[idx] = 0;
            }
        }
    }
    if (counts) { counts->updated = nu; counts->colour = nc; }
}

void orc_mv_integrate(float* tsdf, float* weight, float* color,
                      int dx, int dy, int dz, const float* origin, float voxel,
                      const float* K, const float* c2w,
                      const float* color_packed, const float* depth, int H, int W,
                      float trunc, float obs_weight, float weight_clamp, float reintegrate,
                      const float* old_bnd, int decode_mode, orc_counts* counts) {
    orc_mv_integrate_range(tsdf, weight, color, dx, dy, dz, origin, voxel, K, c2w, color_packed, depth, H, W, trunc, obs_weight,
                           weight_clamp, reintegrate, old_bnd, decode_mode, 0, (int64_t)dx * dy * dz, counts);
}

/* V2 -- model/Volume.py:128-194.  Full-precision fp32 origins (no int cast here). */
void orc_mv_shift(float* tsdf, float* weight, float* color,
                  const float* old_tsdf, const float* old_weight, const float* old_color,
                  int dx, int dy, int dz, const float* origin,
                  int odx, int ody, int odz, const float* old_origin, float voxel,
                  int decode_mode) {
    int64_t n = (int64_t)dx * dy * dz;
    for (int64_t idx = 0; idx < n; ++idx) {
        float vx, vy, vz;
        mv_decode(idx, dx, dy, dz, decode_mode, &vx, &vy, &vz);
        float wx = MADD(vx, voxel, origin[0]);
        float wy = MADD(vy, voxel, origin[1]);
        float wz = MADD(vz, voxel, origin[2]);
        int ox = (int)roundf((wx - old_origin[0]) / voxel);
        int oy = (int)roundf((wy - old_origin[1]) / voxel);
        int oz = (int)roundf((wz - old_origin[2]) / voxel);
        if (0 <= ox && ox < odx && 0 <= oy && oy < ody && 0 <= oz && oz < odz) {
            int64_t o = (int64_t)oz + (int64_t)oy * odz + (int64_t)ox * ody * odz;
            tsdf[idx] = old_tsdf[o]; weight[idx] = old_weight[o]; color[idx] = old_color[o];
        } else {
            tsdf[idx] = 1.0f; weight[idx] = 0; color[idx] = 0;
        }
    }
}

/* V3 -- model/Volume.py:337-458.  `auto tri_x = 0.0` makes the accumulators double; each
 * term is an fp32 product.  out is [n,5] = (tsdf, r, g, b, tsdf@low corner). */
void orc_mv_trilerp(const float* tsdf, const float* weight, const float* color,
                    int dx, int dy, int dz, const float* origin, float voxel,
                    const float* pts, int64_t n, float* out) {
    (void)weight;
    for (int64_t p = 0; p < n; ++p) {
        float x = pts[p * 3], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
        int lx = (int)floorf((x - origin[0]) / voxel);
        int ly = (int)floorf((y - origin[1]) / voxel);
        int lz = (int)floorf((z - origin[2]) / voxel);
        float xo = MADD((float)lx, voxel, origin[0]);
        float yo = MADD((float)ly, voxel, origin[1]);
        float zo = MADD((float)lz, voxel, origin[2]);
        if (lx < 0 || lx >= dx - 1 || ly < 0 || ly >= dy - 1 || lz < 0 || lz >= dz - 1) {
            out[p * 5] = 1.0f; out[p * 5 + 1] = 0; out[p * 5 + 2] = 0; out[p * 5 + 3] = 0; out[p * 5 + 4] = 0;
            continue;
        }
        float u = (x - xo) / voxel, v = (y - yo) / voxel, w = (z - zo) / voxel;
        double t = 0.0, cb = 0.0, cg = 0.0, cr = 0.0;
        int64_t low0 = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
            int64_t id = (int64_t)(lz + k) + (int64_t)(ly + j) * dz + (int64_t)(lx + i) * dy * dz;
            if (!i && !j && !k) low0 = id;
            float c = color[id];
            float b = floorf(c / 65536);
            float g = floorf((c - b * 65536) / 256);
            float r = floorf(c - b * 65536 - g * 256);
            float wu = MADD((float)i, u, (float)(1 - i) * (1 - u));
            float wv = MADD((float)j, v, (float)(1 - j) * (1 - v));
            float ww = MADD((float)k, w, (float)(1 - k) * (1 - w));
            float wt = wu * wv * ww;
            t += wt * tsdf[id]; cb += wt * b; cg += wt * g; cr += wt * r;
        }
        out[p * 5] = (float)t;
        out[p * 5 + 1] = (float)floor(cr);
        out[p * 5 + 2] = (float)floor(cg);
        out[p * 5 + 3] = (float)floor(cb);
        out[p * 5 + 4] = tsdf[low0];
    }
}

/* V4 -- model/Volume.py:462-487: `float weight_threshold=(int) other_params[0]` */
void orc_mv_filter(float* tsdf, float* weight, float* color, int64_t n, float thr) {
    float t = (float)(int)thr;
    for (int64_t i = 0; i < n; ++i) {
        if (weight[i] >= t || weight[i] == 0) continue;
        weight[i] = 0; tsdf[i] = 1; color[i] = 0;
    }
}

/* V5 -- model/Volume.py:489-559.  Slot = idx % pc_num; later voxels overwrite earlier ones
 * (the CUDA scatter is racy; a serial sweep in index order is one valid outcome). Returns
 * the value the reference accumulates in pc_count. */
int64_t orc_mv_truncated_pc(const float* tsdf, const float* color, int dx, int dy, int dz,
                            const float* origin, float voxel, float trunc, int pc_num,
                            float trunc_tsdf, float* pc7, int decode_mode) {
    int64_t n = (int64_t)dx * dy * dz, cnt = 0;
    for (int64_t idx = 0; idx < n; ++idx) {
        float t = tsdf[idx];
        float oc = color[idx];
        float ob = floorf(oc / (256 * 256));
        float og = floorf((oc - ob * 256 * 256) / 256);
        float or_ = oc - ob * 256 * 256 - og * 256;
        if (t <= -trunc_tsdf || t >= trunc_tsdf) continue;
        float vx, vy, vz;
        mv_decode(idx, dx, dy, dz, decode_mode, &vx, &vy, &vz);
        float px = MADD((vx + 0.5f), voxel, origin[0]);
        float py = MADD((vy + 0.5f), voxel, origin[1]);
        float pz = MADD((vz + 0.5f), voxel, origin[2]);
        int64_t s = idx % pc_num;
        pc7[s * 7 + 0] = px; pc7[s * 7 + 1] = py; pc7[s * 7 + 2] = pz;
        pc7[s * 7 + 3] = t * trunc;
        pc7[s * 7 + 4] = or_; pc7[s * 7 + 5] = og; pc7[s * 7 + 6] = ob;
        ++cnt;
    }
    return cnt;
}

/* V6 -- model/Volume.py:561-583 */
void orc_mv_fill(float* tsdf, float* weight, float* color, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { tsdf[i] = 1.0f; weight[i] = 0; color[i] = 0; }
}

/* V7 -- model/Volume.py:585-610 */
void orc_mv_copy(const float* tsdf, const float* weight, const float* color,
                 float* tsdf_b, float* weight_b, float* color_b, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { tsdf_b[i] = tsdf[i]; weight_b[i] = weight[i]; color_b[i] = color[i]; }
}

/* G1 -- mp_slam/mapper.py:37-158.  trgb is [R^3,4] interleaved (t,r,g,b), x fastest;
 * box = (x0,x1,y0,y1,z0,z1); rgb01 is [H,W,3] floats in 0..1; voxel_size = 1/R (host :225). */
void orc_gbv_integrate(float* trgb, float* w, int rx, int ry, int rz, float voxel_size,
                       const float* box, const float* K, const float* c2w,
                       const float* rgb01, const float* depth, int H, int W,
                       float trunc, float obs_weight, int decode_mode, orc_counts* counts) {
    int64_t n = (int64_t)rx * ry * rz;
    int64_t nu = 0;
    for (int64_t idx = 0; idx < n; ++idx) {
        float vx, vy, vz;
        if (decode_mode == 1) {
            int64_t z = idx / ((int64_t)rx * ry), r = idx - z * rx * ry, y = r / rx;
            vz = (float)z; vy = (float)y; vx = (float)(r - y * rx);
        } else {  /* literal (mapper.py:73-75) */
            int vi = (int)idx;
            vz = floorf(((float)vi) / ((float)(rx * ry)));
            vy = floorf(((float)(vi - ((int)vz) * rx * ry)) / ((float)rx));
            vx = (float)(vi - ((int)vz) * rx * ry - ((int)vy) * rx);
        }
        /* pt = start + ((v)*voxel_size)*(end-start) */
        float px = MADD((vx * voxel_size), (box[1] - box[0]), box[0]);
        float py = MADD((vy * voxel_size), (box[3] - box[2]), box[2]);
        float pz = MADD((vz * voxel_size), (box[5] - box[4]), box[4]);
        float cx, cy, cz;
        to_cam(c2w, px, py, pz, &cx, &cy, &cz);
        if (cz <= 0) continue;
        int pix; float diff;
        if (!project_sdf(K, cx, cy, cz, H, W, depth, &pix, &diff)) continue;
        if (diff < -1 * trunc) continue;
        float dist = fminf(1.0f, diff / trunc);
        float w_old = w[idx];
        float w_new = w_old + obs_weight;
        float new_t = MADD(trgb[idx * 4], w_old, obs_weight * dist) / w_new;
        if (obs_weight < 0 && w_old <= 1) {
            trgb[idx * 4] = 1.0f; w[idx] = 0.0f;
            trgb[idx * 4 + 1] = 0; trgb[idx * 4 + 2] = 0; trgb[idx * 4 + 3] = 0;
            continue;
        }
        if (new_t > 1.0f) continue;
        trgb[idx * 4] = new_t;
        float ob = trgb[idx * 4 + 3], og = trgb[idx * 4 + 2], or_ = trgb[idx * 4 + 1];
        float nb = rgb01[pix * 3 + 2], ng = rgb01[pix * 3 + 1], nr = rgb01[pix * 3];
        nb = fminf(MADD(ob, w_old, obs_weight * nb) / w_new, 1.0f);
        ng = fminf(MADD(og, w_old, obs_weight * ng) / w_new, 1.0f);
        nr = fminf(MADD(or_, w_old, obs_weight * nr) / w_new, 1.0f);
        trgb[idx * 4 + 1] = nr; trgb[idx * 4 + 2] = ng; trgb[idx * 4 + 3] = nb;
        w[idx] = w_new;
        ++nu;
    }
    if (counts) { counts->updated = nu; counts->colour = nu; }
}

/* G2 -- mp_slam/mapper.py:161-183 */
void orc_gbv_clear(float* trgb, int64_t n) {
    for (int64_t i = 0; i < n; ++i) { trgb[i * 4] = 1.0f; trgb[i * 4 + 1] = 0; trgb[i * 4 + 2] = 0; trgb[i * 4 + 3] = 0; }
}

int orc_fma_mode(void) { return ORC_FMA; }
