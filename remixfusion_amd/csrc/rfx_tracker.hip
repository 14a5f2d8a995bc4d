// rfx_tracker.hip -- the three tracker kernels of the random-optimisation pose search:
// replaces compute_vertex / compute_normal / compute_tsdf_value of model/ROtracker.py:144-403.
//
// compute_tsdf_value is the hot one.  The reference launches one thread per (candidate, pixel) and
// combines them with one system-scope float atomic each (3e6 atomics onto 1e4 addresses).  Here a
// thread owns a candidate pose and walks a slab of the sub-sampled pixels, accumulating in registers:
// the vertex / normal of a pixel is wave-uniform (scalar loads), only the nearest-voxel TSDF lookup
// is a per-lane gather, and the slabs of one candidate are combined with a handful of atomics.
#include "rfx_common.h"
#include <algorithm>

namespace rfx {

// counter-based uniform in (0,1] standing in for curand_uniform (XORWOW is not reproduced)
__device__ __forceinline__ float hash_uniform(unsigned seed, unsigned row, unsigned draw) {
    unsigned x = seed * 0x9E3779B1u ^ (row + 0x7F4A7C15u) * 0x85EBCA77u ^ (draw + 1u) * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void track_vertex_kernel(const float* __restrict__ depth, float4* __restrict__ vertex,
                                                           int H, int W, float fx, float fy, float cx, float cy, float cutdist,
                                                           float trunc, float sample_range, unsigned seed,
                                                           const float* __restrict__ u_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int pi = i / W, pj = i - pi * W;
    float d = depth[i];
    if (d > cutdist) d = 0.f;
    if (d <= 0.f) { vertex[i] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const float u1 = u_rows ? u_rows[pi * 2] : hash_uniform(seed, pi, 0);
    const float u2 = u_rows ? u_rows[pi * 2 + 1] : hash_uniform(seed, pi, 1);
    float sample = madd(u1, sample_range + 1.0f, -sample_range);
    float z_val = sample * trunc;
    if (sample_range < 1.0f) {
        sample = madd(u2 * 2.0f, sample_range, -sample_range);
        z_val = sample * trunc;
    }
    float gt = -sample;
    if (z_val < -1.0f * trunc) gt = 1.0f;
    if (z_val > 1.0f * trunc) gt = 1.0f;
    const float cz = d + z_val;
    vertex[i] = make_float4(((float)pj - cx) * cz / fx, ((float)pi - cy) * cz / fy, cz, gt);
}

__global__ __launch_bounds__(256) void track_normal_kernel(const float4* __restrict__ v, float* __restrict__ n3, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int pi = i / W, pj = i - pi * W;
    if (pi > H - 2 || pj > W - 2 || pi < 1 || pj < 1) return;
    const float4 c = v[i], l = v[i - 1], r = v[i + 1], u = v[i - W], dn = v[i + W];
    if (c.z == 0.f || l.z == 0.f || r.z == 0.f || u.z == 0.f || dn.z == 0.f) {
        n3[i * 3] = 0.f; n3[i * 3 + 1] = 0.f; n3[i * 3 + 2] = 0.f;
        return;
    }
    const float hx = l.x - r.x, hy = l.y - r.y, hz = l.z - r.z;
    const float vx = u.x - dn.x, vy = u.y - dn.y, vz = u.z - dn.z;
    float nx = madd(-hz, vy, hy * vz);
    float ny = madd(hz, vx, -(hx * vz));
    float nz = madd(-hy, vx, hx * vy);
    const float len = sqrtf(madd(nz, nz, madd(nx, nx, ny * ny)));
    nx = nx / len; ny = ny / len; nz = nz / len;
    if (nz > 0.f) { nx *= -1.f; ny *= -1.f; nz *= -1.f; }
    n3[i * 3] = nx; n3[i * 3 + 1] = ny; n3[i * 3 + 2] = nz;
}

struct EvalK {
    float R[9], T[3], ss[6], K[9];
    int dx, dy, dz, ox, oy, oz;
    int x0, x1;            // the x-planes tsdf holds (one slab of the volume; whole volume: 0, dx)
    float voxel;
    int P, H, W, level, level_index, gh, gw, n_slabs;
};

__global__ __launch_bounds__(256) void track_evaluate_kernel(EvalK E, const float* __restrict__ tsdf,
                                                             const float4* __restrict__ vertex, const float* __restrict__ n3,
                                                             const float* __restrict__ q6, float* __restrict__ value,
                                                             float* __restrict__ count) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = node < E.P;
    float tx = 0.f, ty = 0.f, tz = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if (live) {
        tx = q6[node * 6 + 0] * E.ss[0]; ty = q6[node * 6 + 1] * E.ss[1]; tz = q6[node * 6 + 2] * E.ss[2];
        q1 = q6[node * 6 + 3] * E.ss[3]; q2 = q6[node * 6 + 4] * E.ss[4]; q3 = q6[node * 6 + 5] * E.ss[5];
    }
    const float q0 = sqrtf(madd(-q3, q3, madd(-q2, q2, madd(-q1, q1, 1.0f))));
    const int n_pix = E.gh * E.gw;
    const int per = (n_pix + E.n_slabs - 1) / E.n_slabs;
    const int p0 = blockIdx.y * per, p1 = min(n_pix, p0 + per);
    const int im_h = E.gh * E.level, im_w = E.gw * E.level;
    float acc = 0.f, cnt = 0.f;
    for (int p = p0; p < p1; ++p) {                      // wave-uniform pixel walk
        const int i = p / E.gw, j = p - i * E.gw;
        const int pi = i * E.level + E.level_index, pj = j * E.level + E.level_index;
        if (pi > im_h - 1 || pj > im_w - 1) continue;
        const int c = pi * E.W + pj;
        if (n3[c * 3] == 0.f && n3[c * 3 + 1] == 0.f && n3[c * 3 + 2] == 0.f) continue;
        const float4 v = vertex[c];
        if (v.x == 0.f && v.y == 0.f && v.z == 0.f) continue;
        const float gx = madd(E.R[2], v.z, madd(E.R[0], v.x, E.R[1] * v.y));
        const float gy = madd(E.R[5], v.z, madd(E.R[3], v.x, E.R[4] * v.y));
        const float gz = madd(E.R[8], v.z, madd(E.R[6], v.x, E.R[7] * v.y));
        const float qw = -(madd(gz, q3, madd(gx, q1, gy * q2)));
        const float qx = madd(q2, gz, madd(q0, gx, -(q3 * gy)));
        const float qy = madd(-q1, gz, madd(q3, gx, q0 * gy));
        const float qz = madd(q0, gz, madd(-q2, gx, q1 * gy));
        const float x = madd(qy, -q3, madd(-qz, -q2, madd(qx, q0, qw * (-q1)))) + tx + E.T[0];
        const float y = madd(-qx, -q3, madd(qw, -q2, madd(qy, q0, qz * (-q1)))) + ty + E.T[1];
        const float z = madd(qw, -q3, madd(qx, -q2, madd(qz, q0, -(qy * (-q1))))) + tz + E.T[2];
        const float vx = x - E.T[0], vy = y - E.T[1], vz = z - E.T[2];
        const float cx = madd(E.R[6], vz, madd(E.R[0], vx, E.R[3] * vy));
        const float cy = madd(E.R[7], vz, madd(E.R[1], vx, E.R[4] * vy));
        const float cz = madd(E.R[8], vz, madd(E.R[2], vx, E.R[5] * vy));
        const int px = (int)((cx * E.K[0]) / cz + E.K[2] + 0.5f);
        const int py = (int)((cy * E.K[4]) / cz + E.K[5] + 0.5f);
        if (!(live && px >= 0 && py >= 0 && px < E.W && py < E.H && cz >= 0.f)) continue;
        const int vxi = (int)roundf((x - (float)E.ox) / E.voxel);
        const int vyi = (int)roundf((y - (float)E.oy) / E.voxel);
        const int vzi = (int)roundf((z - (float)E.oz) / E.voxel);
        if (vxi < 1 || vxi >= E.dx - 1 || vyi < 1 || vyi >= E.dy - 1 || vzi < 1 || vzi >= E.dz - 1) continue;
        if (vxi < E.x0 || vxi >= E.x1) continue;             // another slab's voxel: that rank adds this term
        const int64_t idx = (int64_t)vzi + (int64_t)vyi * E.dz + (int64_t)(vxi - E.x0) * E.dy * E.dz;
        acc += fabsf(tsdf[idx] - v.w);
        cnt += 1.0f;
    }
    if (live && cnt > 0.f) {
        if (E.n_slabs == 1) { value[node] = acc; count[node] = cnt; }
        else { atomicAdd(value + node, acc); atomicAdd(count + node, cnt); }
    }
}

}  // namespace rfx

using namespace rfx;

extern "C" {

int rfx_track_vertex(const float* depth, float* vertex4, const float K[9], int H, int W, float cut_dist, float trunc,
                     float sample_range, uint32_t seed, const float* u_rows, rfx_stream stream) {
    if (!depth || !vertex4 || !K || H <= 0 || W <= 0) return RFX_ERR_ARG;
    if ((uintptr_t)vertex4 & 15) return RFX_ERR_ARG;
    hipLaunchKernelGGL(track_vertex_kernel, dim3((H * W + 255) / 256), dim3(256), 0, as_stream(stream), depth,
                       reinterpret_cast<float4*>(vertex4), H, W, K[0], K[4], K[2], K[5], cut_dist, trunc, sample_range, seed, u_rows);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_track_normal(const float* vertex4, float* normal3, int H, int W, rfx_stream stream) {
    if (!vertex4 || !normal3 || H <= 0 || W <= 0) return RFX_ERR_ARG;
    hipLaunchKernelGGL(track_normal_kernel, dim3((H * W + 255) / 256), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(vertex4), normal3, H, W);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_track_evaluate(const float* tsdf, int dx, int dy, int dz, const float origin[3], float voxel,
                       const float* vertex4, const float* normal3, const float R[9], const float T[3],
                       const float* q6, const float search_size[6], int n_candidates, const float K[9], int H, int W,
                       int level, int level_index, float* value, float* count, rfx_stream stream) {
    return rfx_track_evaluate_slab(tsdf, dx, dy, dz, 0, dx, origin, voxel, vertex4, normal3, R, T, q6, search_size, n_candidates, K, H, W,
                                   level, level_index, value, count, stream);
}

int rfx_track_evaluate_slab(const float* tsdf, int dx, int dy, int dz, int x0, int x1, const float origin[3], float voxel,
                            const float* vertex4, const float* normal3, const float R[9], const float T[3],
                            const float* q6, const float search_size[6], int n_candidates, const float K[9], int H, int W,
                            int level, int level_index, float* value, float* count, rfx_stream stream) {
    if (!tsdf || !origin || !vertex4 || !normal3 || !R || !T || !q6 || !search_size || !K || !value || !count) return RFX_ERR_ARG;
    if (dx <= 2 || dy <= 2 || dz <= 2 || n_candidates <= 0 || H <= 0 || W <= 0 || level <= 0 || !(voxel > 0.f)) return RFX_ERR_ARG;
    if (x0 < 0 || x1 > dx || x1 < x0) return RFX_ERR_ARG;
    EvalK E;
    E.x0 = x0; E.x1 = x1;
    for (int i = 0; i < 9; ++i) { E.R[i] = R[i]; E.K[i] = K[i]; }
    for (int i = 0; i < 3; ++i) E.T[i] = T[i];
    for (int i = 0; i < 6; ++i) E.ss[i] = search_size[i];
    E.dx = dx; E.dy = dy; E.dz = dz;
    E.ox = (int)origin[0]; E.oy = (int)origin[1]; E.oz = (int)origin[2];     // (int) other_params[3..5]
    E.voxel = voxel; E.P = n_candidates; E.H = H; E.W = W; E.level = level; E.level_index = level_index;
    E.gh = H / level; E.gw = W / level;
    const int n_pix = E.gh * E.gw;
    if (n_pix <= 0) return RFX_ERR_ARG;
    const int blocks_x = (n_candidates + 255) / 256;
    // enough (candidate-block, pixel-slab) pairs to fill 256 CUs several times over
    E.n_slabs = std::max(1, std::min(n_pix, (256 * 8 + blocks_x - 1) / blocks_x));
    hipStream_t st = as_stream(stream);
    RFX_HIP_TRY(hipMemsetAsync(value, 0, sizeof(float) * n_candidates, st));
    RFX_HIP_TRY(hipMemsetAsync(count, 0, sizeof(float) * n_candidates, st));
    hipLaunchKernelGGL(track_evaluate_kernel, dim3(blocks_x, E.n_slabs), dim3(256), 0, st, E, tsdf,
                       reinterpret_cast<const float4*>(vertex4), normal3, q6, value, count);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"
