// rfx_tracker.hip -- the three tracker kernels of the random-optimisation pose search:
// replaces compute_vertex / compute_normal / compute_tsdf_value of model/ROtracker.py:144-403.
//
// compute_tsdf_value is the hot one.  The reference launches one thread per (candidate, pixel) and
// combines them with one system-scope float atomic each (3e6 atomics onto 1e4 addresses).  Here a
// thread owns a candidate pose and walks a slab of the sub-sampled pixels, accumulating in registers:
// the vertex / normal of a pixel is wave-uniform (scalar loads), only the nearest-voxel TSDF lookup
// is a per-lane gather, and the slabs of one candidate are combined with a handful of atomics.
//
// Round 5: the sums are ORDER-INDEPENDENT.  Every term |tsdf - target| (<= 2) is converted to 30-bit fixed point
// (truncation: (uint32)(term * 2^30)) and added as a 64-bit integer -- in the thread, between the pixel slabs of a candidate
// (integer atomics), between the x-slabs of a volume spread over several GPUs (an int64 all-reduce) -- so the sum is the same
// number whatever the grouping, and sharded == single GPU bit for bit; the hit counts are integers too.  (Up to round 4 the
// slabs were combined with float atomics, like the reference combines its pixels: a 20-iteration search amplifies the last
// bit of such a sum into centimetres between two runs.)  value = acc * 2^-30, rounded to float32 once, where it is used.
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

// one (candidate, pixel-slab) pair's share of compute_tsdf_value: thread = candidate `node`, walks pixel slab `slab`
__device__ __forceinline__ void evaluate_share(const EvalK& E, int node, int slab, const float* __restrict__ tsdf,
                                               const float4* __restrict__ vertex, const float* __restrict__ n3,
                                               const float* __restrict__ q6, unsigned long long* __restrict__ value,
                                               unsigned long long* __restrict__ count) {
    const bool live = node < E.P;
    float tx = 0.f, ty = 0.f, tz = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if (live) {
        tx = q6[node * 6 + 0] * E.ss[0]; ty = q6[node * 6 + 1] * E.ss[1]; tz = q6[node * 6 + 2] * E.ss[2];
        q1 = q6[node * 6 + 3] * E.ss[3]; q2 = q6[node * 6 + 4] * E.ss[4]; q3 = q6[node * 6 + 5] * E.ss[5];
    }
    const float q0 = sqrtf(madd(-q3, q3, madd(-q2, q2, madd(-q1, q1, 1.0f))));
    const int n_pix = E.gh * E.gw;
    const int per = (n_pix + E.n_slabs - 1) / E.n_slabs;
    const int p0 = slab * per, p1 = min(n_pix, p0 + per);
    const int im_h = E.gh * E.level, im_w = E.gw * E.level;
    unsigned long long acc = 0ull;                        // sum of the terms in 2^-30 units (a term is <= 2: tsdf and target lie in [-1, 1])
    unsigned cnt = 0u;
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
        // the term as a multiple of 2^-30, truncated.  |tsdf - target| <= 2 for valid data; anything above 3 -- and a NaN, which
        // in the reference poisons the candidate's float sum so that `m < origin` rejects it -- counts as 3: the conversion is
        // then defined in C++ (no reliance on v_cvt_u32_f32 saturating) and a NaN makes the candidate WORSE, never better
        const float term = fabsf(tsdf[idx] - v.w);
        acc += (unsigned long long)(unsigned)((term <= 3.0f ? term : 3.0f) * 1073741824.0f);
        cnt += 1u;
    }
    if (live && cnt > 0u) {
        if (E.n_slabs == 1) { value[node] = acc; count[node] = (unsigned long long)cnt; }
        else { atomicAdd(value + node, acc); atomicAdd(count + node, (unsigned long long)cnt); }
    }
}

__global__ __launch_bounds__(256) void track_evaluate_kernel(EvalK E, const float* __restrict__ tsdf,
                                                             const float4* __restrict__ vertex, const float* __restrict__ n3,
                                                             const float* __restrict__ q6, unsigned long long* __restrict__ value,
                                                             unsigned long long* __restrict__ count) {
    evaluate_share(E, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, tsdf, vertex, n3, q6, value, count);
}

// candidate blocks x pixel slabs of one evaluation: enough pairs to fill 256 CUs several times over
__host__ __device__ inline int eval_slabs(int blocks_x, int n_pix) {
    const int want = (256 * 8 + blocks_x - 1) / blocks_x;
    const int s = want < n_pix ? want : n_pix;
    return s > 1 ? s : 1;
}

// ---- the whole 20-iteration search on the device (rfx_track_search_*): the pose, the search box and the loop's flags live in
// `state` (RFX_TRACK_STATE_WORDS words), every launch reads them from there, the host reads them once per frame.
enum { ST_R = 0, ST_T = 9, ST_SS = 12, ST_PSS = 18, ST_MIN_TSDF = 24, ST_COUNT_PARTICLE = 32, ST_LEVEL_INDEX = 33, ST_SUCCESS = 34,
       ST_PREVIOUS_SUCCESS = 35, ST_FIRST_SUCCESS = 36, ST_ERROR = 37, ST_N_SUCCESS = 38, ST_ITERATION = 39 };
static_assert(RFX_TRACK_STATE_WORDS >= 40, "state words");

struct SearchK {
    const float* tsdf; const float4* vertex; const float* n3;
    const float* templates[RFX_TRACK_STEPS];
    int rows[RFX_TRACK_STEPS], n_eval[RFX_TRACK_STEPS], level[RFX_TRACK_STEPS];
    float K[9];
    int dx, dy, dz, x0, x1, ox, oy, oz, H, W;
    float voxel;
    int count_search, fix_level_index, iterative_scale, max_rows;
    double scale_d, beta_d;                                    // RO.scaling_coefficient, beta as the host's Python floats
    float* state; unsigned long long* value; unsigned long long* count;
};

// mean |tsdf - target| of a candidate from its fixed-point sum and hit count, in float32 like the host's numpy arrays:
// value = acc * 2^-30 rounded once (exact in double: acc < 2^53), then value / (count + 1e-6)
__device__ __forceinline__ float search_mean(const unsigned long long* __restrict__ value, const unsigned long long* __restrict__ count, int i) {
    return (float)((double)value[i] * 9.313225746154785e-10) / ((float)count[i] + 1e-6f);
}

__global__ __launch_bounds__(64) void track_search_begin_kernel(float* __restrict__ state, unsigned long long* __restrict__ value,
                                                                unsigned long long* __restrict__ count, int max_rows, EvalK init) {
    int* si = reinterpret_cast<int*>(state);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {
        if (t < 9) state[ST_R + t] = init.R[t];
        if (t < 3) state[ST_T + t] = init.T[t];
        if (t < 6) { state[ST_SS + t] = init.ss[t]; state[ST_PSS + t] = 0.f; }
        if (t >= 24 && t < RFX_TRACK_STATE_WORDS) si[t] = (t == ST_LEVEL_INDEX) ? 5 : 0;     // one owner per word; reference :737 `level_index = 5`
    }
    for (int i = t; i < max_rows; i += gridDim.x * blockDim.x) { value[i] = 0ull; count[i] = 0ull; }
}

__global__ __launch_bounds__(256) void track_search_evaluate_kernel(SearchK S) {
    const int* si = reinterpret_cast<const int*>(S.state);
    const int cp = si[ST_COUNT_PARTICLE];
    EvalK E;
    for (int i = 0; i < 9; ++i) { E.R[i] = S.state[ST_R + i]; E.K[i] = S.K[i]; }
    for (int i = 0; i < 3; ++i) E.T[i] = S.state[ST_T + i];
    for (int i = 0; i < 6; ++i) E.ss[i] = S.state[ST_SS + i];
    E.dx = S.dx; E.dy = S.dy; E.dz = S.dz; E.ox = S.ox; E.oy = S.oy; E.oz = S.oz; E.x0 = S.x0; E.x1 = S.x1; E.voxel = S.voxel;
    E.P = S.n_eval[cp]; E.H = S.H; E.W = S.W; E.level = S.level[cp]; E.level_index = si[ST_LEVEL_INDEX];
    E.gh = S.H / E.level; E.gw = S.W / E.level;
    const int blocks_x = (E.P + 255) / 256;
    E.n_slabs = eval_slabs(blocks_x, E.gh * E.gw);
    const int bx = blockIdx.x % blocks_x, slab = blockIdx.x / blocks_x;      // the grid is sized for the largest template
    if (slab >= E.n_slabs) return;
    evaluate_share(E, bx * 256 + threadIdx.x, slab, S.tsdf, S.vertex, S.n3, S.templates[cp], S.value, S.count);
}

// cal_transform + the bookkeeping of one iteration of random_optimization (model/ROtracker.py:606-709, :745-826), one block.
// Arithmetic types follow the host loop of remixfusion_amd/model/ROtracker.py = the reference under its numpy 1.21.6.
__global__ __launch_bounds__(1024) void track_search_update_kernel(SearchK S, int iteration) {
    __shared__ int wave_total[16], wave_base[16];
    __shared__ int sel_idx[RFX_TRACK_MAX_COUNT_SEARCH];
    __shared__ float sel_fit[RFX_TRACK_MAX_COUNT_SEARCH];
    __shared__ double col[9][RFX_TRACK_MAX_COUNT_SEARCH];
    __shared__ double sums[9];
    __shared__ int bad, n_better;
    int* si = reinterpret_cast<int*>(S.state);
    const int cp = si[ST_COUNT_PARTICLE];
    const int n_all = S.rows[cp], P = S.n_eval[cp];
    const float* __restrict__ cand = S.templates[cp];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float origin = search_mean(S.value, S.count, 0);
    const int ipt = (n_all + 1023) / 1024;                         // <= 16 (checked on the host)
    const int i0 = t * ipt, i1 = min(n_all, i0 + ipt);
    if (t == 0) bad = 0;
    // ---- rank of every candidate that beats candidate 0, in index order
    int mine = 0;
    for (int i = max(i0, 1); i < i1; ++i) {
        const float m = i < P ? search_mean(S.value, S.count, i) : 0.f;     // rows past P are never evaluated: sums 0 (host: zeros)
        mine += m < origin ? 1 : 0;
    }
    int scan = mine;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(scan, d, 64); if (lane >= d) scan += o; }
    if (lane == 63) wave_total[wave] = scan;
    __syncthreads();
    if (t == 0) { int a = 0; for (int w = 0; w < 16; ++w) { wave_base[w] = a; a += wave_total[w]; } n_better = a; }
    __syncthreads();
    int rank = wave_base[wave] + scan - mine;
    for (int i = max(i0, 1); i < i1 && rank < S.count_search; ++i) {
        const float m = i < P ? search_mean(S.value, S.count, i) : 0.f;
        if (m < origin) { sel_idx[rank] = i; sel_fit[rank] = m; ++rank; }
    }
    __syncthreads();
    const int m_sel = min(n_better, S.count_search);
    const float* ss = S.state + ST_SS;
    if (t < m_sel) {
        // types as the reference's numpy (1.21.6) gives them (oracle/tracker_host_oracle.py): float32 weights and products,
        // float64 sums; -ffp-contract=off keeps every product a rounded float32
        const int i = sel_idx[t];
        const float fit = sel_fit[t], w = origin - fit;
        const float c0 = cand[i * 6 + 0], c1 = cand[i * 6 + 1], c2 = cand[i * 6 + 2], c3 = cand[i * 6 + 3], c4 = cand[i * 6 + 4], c5 = cand[i * 6 + 5];
        const float qx = c3 * ss[3], qy = c4 * ss[4], qz = c5 * ss[5];
        const double rad = ((1.0 - (double)(qx * qx)) - (double)(qy * qy)) - (double)(qz * qz);
        if (rad < 0.0) atomicOr(&bad, 1);
        col[0][t] = (double)w; col[1][t] = (double)(fit * w); col[2][t] = (double)(c0 * w); col[3][t] = (double)(c1 * w);
        col[4][t] = (double)(c2 * w); col[5][t] = sqrt(rad < 0.0 ? 0.0 : rad) * (double)w; col[6][t] = (double)(c3 * w);
        col[7][t] = (double)(c4 * w); col[8][t] = (double)(c5 * w);
    }
    __syncthreads();
    if (t < 9) { double a = 0.0; for (int k = 0; k < m_sel; ++k) a += col[t][k]; sums[t] = a; }   // in candidate order, like the reference's loop
    __syncthreads();
    // ---- zero the sums for the next evaluation
    for (int i = t; i < S.max_rows; i += 1024) { S.value[i] = 0ull; S.count[i] = 0ull; }
    if (t != 0) return;
    float* st = S.state;
    const bool success = m_sel > 0;
    float mt[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    double tsdf_scale;                                             // `scale * tsdf` of update_PST
    if (success) {
        if (bad) si[ST_ERROR] = 1;                                 // invalid quaternion in the template (reference exits, :662-669)
        const double sw = sums[0];
        const double mean_tsdf = sums[1] / sw;
        mt[0] = (float)(sums[2] / sw * (double)ss[0]); mt[1] = (float)(sums[3] / sw * (double)ss[1]); mt[2] = (float)(sums[4] / sw * (double)ss[2]);
        const double q0 = sums[5] / sw, q1 = sums[6] / sw * (double)ss[3], q2 = sums[7] / sw * (double)ss[4], q3 = sums[8] / sw * (double)ss[5];
        const double lens = 1.0 / sqrt(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3);
        mt[3] = (float)(q0 * lens); mt[4] = (float)(q1 * lens); mt[5] = (float)(q2 * lens); mt[6] = (float)(q3 * lens);
        tsdf_scale = S.scale_d * mean_tsdf;
        st[ST_MIN_TSDF] = (float)mean_tsdf;
    } else {
        tsdf_scale = S.scale_d * (double)origin;
        st[ST_MIN_TSDF] = origin;
    }
    int count_particle = cp;
    if (success) {
        if (count_particle < RFX_TRACK_STEPS - 1) ++count_particle;
        const float qw = mt[3], qx = mt[4], qy = mt[5], qz = mt[6];
        const float Ri[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qz * qw), 2.f * (qx * qz + qy * qw),
                             2.f * (qx * qy + qz * qw), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qx * qw),
                             2.f * (qx * qz - qy * qw), 2.f * (qy * qz + qx * qw), 1.f - 2.f * (qx * qx + qy * qy)};
        float Rn[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rn[r * 3 + c] = (Ri[r * 3] * st[ST_R + c] + Ri[r * 3 + 1] * st[ST_R + 3 + c]) + Ri[r * 3 + 2] * st[ST_R + 6 + c];
        for (int k = 0; k < 9; ++k) st[ST_R + k] = Rn[k];
        for (int k = 0; k < 3; ++k) st[ST_T + k] += mt[k];
        si[ST_N_SUCCESS] += 1;
    }
    int level_index = S.fix_level_index ? 1 : si[ST_LEVEL_INDEX] + 5;
    si[ST_LEVEL_INDEX] = level_index % S.level[count_particle];
    si[ST_COUNT_PARTICLE] = success ? count_particle : 0;          // `if not success: count_particle = 0` at the top of the next iteration
    {   // update_PST (reference :493-534)
        const double min_scale = 1e-3;
        double s[6] = {fabs((double)mt[0]) + min_scale, fabs((double)mt[1]) + min_scale, fabs((double)mt[2]) + min_scale,
                       fabs((double)mt[4]) + min_scale, fabs((double)mt[5]) + min_scale, fabs((double)mt[6]) + min_scale};
        double n2 = 0.0;
        for (int k = 0; k < 6; ++k) n2 += s[k] * s[k];
        const double nrm = sqrt(n2);
        for (int k = 0; k < 6; ++k) st[ST_SS + k] = (float)(tsdf_scale * (s[k] / nrm) + min_scale);
    }
    const bool previous_success = si[ST_PREVIOUS_SUCCESS] != 0;
    if (previous_success && success) {
        for (int k = 0; k < 6; ++k) st[ST_SS + k] = (float)(S.beta_d * (double)st[ST_SS + k] + (1.0 - S.beta_d) * (double)st[ST_PSS + k]);
    } else if (success) {
        if (S.iterative_scale) si[ST_PREVIOUS_SUCCESS] = 1;
        for (int k = 0; k < 6; ++k) st[ST_PSS + k] = st[ST_SS + k];
    }
    if (!success) si[ST_PREVIOUS_SUCCESS] = 0;
    if (iteration == 0) si[ST_FIRST_SUCCESS] = success ? 1 : 0;
    si[ST_SUCCESS] = success ? 1 : 0;
    si[ST_ITERATION] = iteration + 1;
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
                       int level, int level_index, int64_t* value_q30, int64_t* count, rfx_stream stream) {
    return rfx_track_evaluate_slab(tsdf, dx, dy, dz, 0, dx, origin, voxel, vertex4, normal3, R, T, q6, search_size, n_candidates, K, H, W,
                                   level, level_index, value_q30, count, stream);
}

int rfx_track_evaluate_slab(const float* tsdf, int dx, int dy, int dz, int x0, int x1, const float origin[3], float voxel,
                            const float* vertex4, const float* normal3, const float R[9], const float T[3],
                            const float* q6, const float search_size[6], int n_candidates, const float K[9], int H, int W,
                            int level, int level_index, int64_t* value_q30, int64_t* count, rfx_stream stream) {
    if (!tsdf || !origin || !vertex4 || !normal3 || !R || !T || !q6 || !search_size || !K || !value_q30 || !count) return RFX_ERR_ARG;
    if (((uintptr_t)value_q30 | (uintptr_t)count) & 7) return RFX_ERR_ARG;
    unsigned long long* value = reinterpret_cast<unsigned long long*>(value_q30);
    unsigned long long* cnt64 = reinterpret_cast<unsigned long long*>(count);
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
    E.n_slabs = eval_slabs(blocks_x, n_pix);
    hipStream_t st = as_stream(stream);
    RFX_HIP_TRY(hipMemsetAsync(value, 0, sizeof(unsigned long long) * n_candidates, st));
    RFX_HIP_TRY(hipMemsetAsync(cnt64, 0, sizeof(unsigned long long) * n_candidates, st));
    hipLaunchKernelGGL(track_evaluate_kernel, dim3(blocks_x, E.n_slabs), dim3(256), 0, st, E, tsdf,
                       reinterpret_cast<const float4*>(vertex4), normal3, q6, value, cnt64);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

static int search_kernel_args(const rfx_track_search* s, SearchK* S, int* eval_blocks) {
    if (!s || !s->tsdf || !s->vertex4 || !s->normal3 || !s->state || !s->value_q30 || !s->count) return RFX_ERR_ARG;
    if (((uintptr_t)s->value_q30 | (uintptr_t)s->count) & 7) return RFX_ERR_ARG;
    if (s->dx <= 2 || s->dy <= 2 || s->dz <= 2 || s->H <= 0 || s->W <= 0 || !(s->voxel > 0.f)) return RFX_ERR_ARG;
    if (s->x0 < 0 || s->x1 > s->dx || s->x1 < s->x0) return RFX_ERR_ARG;
    if (s->count_search <= 0 || s->count_search > RFX_TRACK_MAX_COUNT_SEARCH) return RFX_ERR_ARG;
    if (((uintptr_t)s->vertex4 & 15) || ((uintptr_t)s->state & 15)) return RFX_ERR_ARG;
    S->tsdf = s->tsdf; S->vertex = reinterpret_cast<const float4*>(s->vertex4); S->n3 = s->normal3;
    int max_rows = 0, blocks = 0;
    for (int k = 0; k < RFX_TRACK_STEPS; ++k) {
        if (!s->templates[k] || s->template_rows[k] <= 0 || s->n_eval[k] <= 0 || s->n_eval[k] > s->template_rows[k] || s->level[k] <= 0)
            return RFX_ERR_ARG;
        if (s->template_rows[k] > 16 * 1024) return RFX_ERR_ARG;              // the update's block holds 16 candidates per thread
        const int n_pix = (s->H / s->level[k]) * (s->W / s->level[k]);
        if (n_pix <= 0) return RFX_ERR_ARG;
        S->templates[k] = s->templates[k]; S->rows[k] = s->template_rows[k]; S->n_eval[k] = s->n_eval[k]; S->level[k] = s->level[k];
        max_rows = std::max(max_rows, s->template_rows[k]);
        const int bx = (s->n_eval[k] + 255) / 256;
        blocks = std::max(blocks, bx * eval_slabs(bx, n_pix));
    }
    for (int i = 0; i < 9; ++i) S->K[i] = s->K[i];
    S->dx = s->dx; S->dy = s->dy; S->dz = s->dz; S->x0 = s->x0; S->x1 = s->x1;
    S->ox = (int)s->origin[0]; S->oy = (int)s->origin[1]; S->oz = (int)s->origin[2];
    S->H = s->H; S->W = s->W; S->voxel = s->voxel;
    S->count_search = s->count_search; S->fix_level_index = s->fix_level_index; S->iterative_scale = s->iterative_scale;
    S->max_rows = max_rows;
    S->scale_d = s->scaling_coefficient; S->beta_d = s->beta;
    S->state = s->state;
    S->value = reinterpret_cast<unsigned long long*>(s->value_q30); S->count = reinterpret_cast<unsigned long long*>(s->count);
    *eval_blocks = blocks;
    return RFX_OK;
}

size_t rfx_track_search_bytes(void) { return sizeof(rfx_track_search); }

int rfx_track_search_begin(const rfx_track_search* s, const float R[9], const float T[3], const float search_size[6], rfx_stream stream) {
    SearchK S; int blocks;
    if (int rc = search_kernel_args(s, &S, &blocks)) return rc;
    if (!R || !T || !search_size) return RFX_ERR_ARG;
    EvalK init = {};
    for (int i = 0; i < 9; ++i) init.R[i] = R[i];
    for (int i = 0; i < 3; ++i) init.T[i] = T[i];
    for (int i = 0; i < 6; ++i) init.ss[i] = search_size[i];
    hipLaunchKernelGGL(track_search_begin_kernel, dim3((S.max_rows + 63) / 64), dim3(64), 0, as_stream(stream), S.state, S.value, S.count,
                       S.max_rows, init);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_track_search_evaluate(const rfx_track_search* s, rfx_stream stream) {
    SearchK S; int blocks;
    if (int rc = search_kernel_args(s, &S, &blocks)) return rc;
    hipLaunchKernelGGL(track_search_evaluate_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), S);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_track_search_update(const rfx_track_search* s, int iteration, rfx_stream stream) {
    SearchK S; int blocks;
    if (int rc = search_kernel_args(s, &S, &blocks)) return rc;
    if (iteration < 0) return RFX_ERR_ARG;
    hipLaunchKernelGGL(track_search_update_kernel, dim3(1), dim3(1024), 0, as_stream(stream), S, iteration);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_track_search_run(const rfx_track_search* s, const float R[9], const float T[3], const float search_size[6], int iterations,
                         rfx_stream stream) {
    SearchK S; int blocks;
    if (int rc = search_kernel_args(s, &S, &blocks)) return rc;
    if (iterations < 0) return RFX_ERR_ARG;
    if (int rc = rfx_track_search_begin(s, R, T, search_size, stream)) return rc;
    for (int i = 0; i < iterations; ++i) {
        hipLaunchKernelGGL(track_search_evaluate_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), S);
        hipLaunchKernelGGL(track_search_update_kernel, dim3(1), dim3(1024), 0, as_stream(stream), S, i);
    }
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"
