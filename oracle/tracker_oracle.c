/*
 * oracle/tracker_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle).
 *
 * Plain-C restatement of the three PyCUDA tracker kernels of the reference
 * (/root/reference/model/ROtracker.py):
 *   orc_tr_vertex   :272-344  compute_vertex   (host :426-451)
 *   orc_tr_normal   :346-403  compute_normal   (host :453-468)
 *   orc_tr_evaluate :144-270  compute_tsdf_value (host :536-604)
 * Same conventions as tsdf_oracle.c: ORC_FMA selects the nvcc -fmad=true contraction model
 * (left-first), compile with -ffp-contract=off.
 *
 * PARITY STATUS: "parity unpinned" against a live run (no vectors in the reference, PyCUDA cannot
 * run here); pinned by known-answer tests.  compute_vertex draws its per-row jitter from cuRAND
 * XORWOW (curand_init(seed, row, 0)): that stream is not reproduced -- the caller passes the
 * uniform numbers (u1, u2 per image row); with RO.sample_range = 0 (every shipped config) the
 * jitter is exactly zero and the kernel is deterministic.  compute_tsdf_value accumulates with
 * float atomics in an arbitrary order; the oracle sums in pixel order.
 */
#include <math.h>
#include <stdint.h>

#ifndef ORC_FMA
#define ORC_FMA 1
#endif
#if ORC_FMA
#define MADD(a, b, c) fmaf((a), (b), (c))
#else
static inline float orc_madd2(float a, float b, float c) { float p = a * b; return p + c; }
#define MADD(a, b, c) orc_madd2((a), (b), (c))
#endif

/* T1.  vertex4 [H*W,4] = (x, y, z, target tsdf).  u1/u2: per-row uniforms in (0,1] (curand_uniform). */
void orc_tr_vertex(const float* depth, float* vertex4, const float* K, int H, int W, float cutdist, float trunc,
                   float sample_range, const float* u1, const float* u2) {
    for (int pi = 0; pi < H; ++pi) for (int pj = 0; pj < W; ++pj) {
        float d = depth[pi * W + pj];
        if (d > cutdist) d = 0.f;
        float* o = vertex4 + (pi * W + pj) * 4;
        if (d <= 0) { o[0] = o[1] = o[2] = o[3] = 0.0f; continue; }
        /* sample = u*(sample_range+1) - sample_range */
        float sample = MADD(u1[pi], (sample_range + 1), -sample_range);
        float z_val = sample * trunc;
        if (sample_range < 1) {
            sample = MADD((u2[pi] * 2), sample_range, -sample_range);   /* (u*2*sample_range) - sample_range */
            z_val = sample * trunc;
        }
        float gt = -sample;
        if (z_val < -1 * trunc) gt = 1.0f;
        if (z_val > 1 * trunc) gt = 1.0f;
        float cz = d + z_val;
        float cx = ((float)pj - K[2]) * cz / K[0];
        float cy = ((float)pi - K[5]) * cz / K[4];
        o[0] = cx; o[1] = cy; o[2] = cz; o[3] = gt;
    }
}

/* T2.  normal3 [H*W,3]; border pixels are left untouched (the kernel returns before writing). */
void orc_tr_normal(const float* v, float* n3, int H, int W) {
    for (int pi = 1; pi <= H - 2; ++pi) for (int pj = 1; pj <= W - 2; ++pj) {
        int c = pi * W + pj, l = c - 1, r = c + 1, u = c - W, dn = c + W;
        float* o = n3 + c * 3;
        if (v[c * 4 + 2] == 0 || v[l * 4 + 2] == 0 || v[r * 4 + 2] == 0 || v[u * 4 + 2] == 0 || v[dn * 4 + 2] == 0) {
            o[0] = o[1] = o[2] = 0.f;
            continue;
        }
        float hx = v[l * 4] - v[r * 4], hy = v[l * 4 + 1] - v[r * 4 + 1], hz = v[l * 4 + 2] - v[r * 4 + 2];
        float vx = v[u * 4] - v[dn * 4], vy = v[u * 4 + 1] - v[dn * 4 + 1], vz = v[u * 4 + 2] - v[dn * 4 + 2];
        float nx = MADD(-hz, vy, hy * vz);          /* -hor_z*ver_y + hor_y*ver_z */
        float ny = MADD(hz, vx, -(hx * vz));        /*  hor_z*ver_x - hor_x*ver_z */
        float nz = MADD(-hy, vx, hx * vy);          /* -hor_y*ver_x + hor_x*ver_y */
        float len = sqrtf(MADD(nz, nz, MADD(nx, nx, ny * ny)));
        nx = nx / len; ny = ny / len; nz = nz / len;
        if (nz > 0) { nx *= -1; ny *= -1; nz *= -1; }
        o[0] = nx; o[1] = ny; o[2] = nz;
    }
}

/* T3.  For every candidate `node` (q6 [P,6] scaled by search_size[6]) and every sub-sampled pixel
 * (pi = i*level+level_index, pj = j*level+level_index; i < H/level, j < W/level): transform the
 * vertex, nearest-voxel lookup in the moving volume, accumulate |tsdf - target| and a hit count. */
void orc_tr_evaluate(const float* tsdf, int dx, int dy, int dz, const float* origin, float voxel,
                     const float* vertex4, const float* normal3, const float* R, const float* T,
                     const float* q6, const float* ss, int P, const float* K, int H, int W, int level, int level_index,
                     float* value, float* count, long long* value_q30) {
    int ox = (int)origin[0], oy = (int)origin[1], oz = (int)origin[2];      /* (int) other_params[3..5] */
    int gh = H / level, gw = W / level;                                       /* grid dims (int(self.im_h/level)) */
    int im_h = gh * level, im_w = gw * level;
    for (int node = 0; node < P; ++node) {
        float cnt = 0.f;
        float accf = 0.f;          /* the float running sum in pixel order (one of the orders the reference's atomics may take) */
        long long accq = 0;        /* the same terms, each truncated to a multiple of 2^-30, as an integer: what the product
                                      (ABI 8) defines as THE sum, because it does not depend on the order (value_q30 may be NULL) */
        for (int i = 0; i < gh; ++i) for (int j = 0; j < gw; ++j) {
            int pi = i * level + level_index, pj = j * level + level_index;
            if (pi > im_h - 1 || pj > im_w - 1 || pi < 0 || pj < 0) continue;
            int c = pi * W + pj;
            if (normal3[c * 3] == 0 && normal3[c * 3 + 1] == 0 && normal3[c * 3 + 2] == 0) continue;
            float x = vertex4[c * 4], y = vertex4[c * 4 + 1], z = vertex4[c * 4 + 2], gt = vertex4[c * 4 + 3];
            if (x == 0 && y == 0 && z == 0) continue;
            /* a*x + b*y + c*z -> fma(c, z, fma(a, x, b*y)) */
            float gx = MADD(R[2], z, MADD(R[0], x, R[1] * y));
            float gy = MADD(R[5], z, MADD(R[3], x, R[4] * y));
            float gz = MADD(R[8], z, MADD(R[6], x, R[7] * y));
            float tx = q6[node * 6 + 0] * ss[0], ty = q6[node * 6 + 1] * ss[1], tz = q6[node * 6 + 2] * ss[2];
            float q1 = q6[node * 6 + 3] * ss[3], q2 = q6[node * 6 + 4] * ss[4], q3 = q6[node * 6 + 5] * ss[5];
            /* sqrt(1 - q1*q1 - q2*q2 - q3*q3): ((1 - q1q1) - q2q2) - q3q3 with fma(-q,q,acc) */
            float q0 = sqrtf(MADD(-q3, q3, MADD(-q2, q2, MADD(-q1, q1, 1.0f))));
            float qw = -(MADD(gz, q3, MADD(gx, q1, gy * q2)));
            float qx = MADD(q2, gz, MADD(q0, gx, -(q3 * gy)));       /* q0*gx - q3*gy + q2*gz */
            float qy = MADD(-q1, gz, MADD(q3, gx, q0 * gy));         /* q3*gx + q0*gy - q1*gz */
            float qz = MADD(q0, gz, MADD(-q2, gx, q1 * gy));         /* -q2*gx + q1*gy + q0*gz */
            /* x = q_x*q0 + q_w*(-q1) - q_z*(-q2) + q_y*(-q3) + t_x + T0 */
            float nx_ = MADD(qy, -q3, MADD(-qz, -q2, MADD(qx, q0, qw * (-q1)))) + tx + T[0];
            float ny_ = MADD(-qx, -q3, MADD(qw, -q2, MADD(qy, q0, qz * (-q1)))) + ty + T[1];
            float nz_ = MADD(qw, -q3, MADD(qx, -q2, MADD(qz, q0, -(qy * (-q1))))) + tz + T[2];
            float vx = nx_ - T[0], vy = ny_ - T[1], vz = nz_ - T[2];
            float cx = MADD(R[6], vz, MADD(R[0], vx, R[3] * vy));
            float cy = MADD(R[7], vz, MADD(R[1], vx, R[4] * vy));
            float cz = MADD(R[8], vz, MADD(R[2], vx, R[5] * vy));
            int px = (int)((cx * K[0]) / cz + K[2] + 0.5f);
            int py = (int)((cy * K[4]) / cz + K[5] + 0.5f);
            if (px >= 0 && py >= 0 && px < W && py < H && cz >= 0) {
                int vxi = (int)roundf((nx_ - ox) / voxel);
                int vyi = (int)roundf((ny_ - oy) / voxel);
                int vzi = (int)roundf((nz_ - oz) / voxel);
                if (vxi < 1 || vxi >= dx - 1 || vyi < 1 || vyi >= dy - 1 || vzi < 1 || vzi >= dz - 1) continue;
                int64_t idx = (int64_t)vzi + (int64_t)vyi * dz + (int64_t)vxi * dy * dz;
                accf += fabsf(tsdf[idx] - gt);
                {   /* terms above 3 and NaN count as 3 (csrc/rfx_tracker.hip): defined conversion, a NaN never helps a candidate */
                    float term = fabsf(tsdf[idx] - gt);
                    accq += (long long)(unsigned)((term <= 3.0f ? term : 3.0f) * 1073741824.0f);
                }
                cnt += 1.0f;
            }
        }
        value[node] = accf; count[node] = cnt;
        if (value_q30) value_q30[node] = accq;
    }
}
