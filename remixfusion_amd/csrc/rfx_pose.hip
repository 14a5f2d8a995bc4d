// rfx_pose.hip -- the RBA pose-refinement MLP as three launches instead of ~80 ATen ops.
// Replaces the host-side torch graph of the reference's model/rba.py:60-100 (7 -> 256 -> 256 -> 256 -> 6
// ELU MLP, residuals scaled by `scale`, camera 0 pinned, then kornia's angle_axis_to_rotation_matrix)
// that every bundle-adjustment iteration runs forward and backward (mp_slam/mapper.py:456,489-497).
// K (keyframes) is small -- tens to a few hundred -- so the work is latency, not throughput:
//   P1 rba_forward_kernel   block = camera, thread = hidden unit; activations kept for the backward
//   P2 rba_backward_kernel  block = camera: dL/dpose -> dL/d(axis-angle, t) -> pre-activation grads
//   P3 rba_wgrad_kernel     block = 8 rows of a layer, thread = column, cameras in order (deterministic sums)
#include "rfx_common.h"

namespace rfx {

constexpr int RBA_H = 256;                 // hidden width (reference model/rba.py:71-75)
constexpr int RBA_IN = 7;
constexpr int RBA_OUT = 6;
constexpr int RBA_ACT_LD = 8 + 3 * RBA_H + 8;      // inp[8] | H1 | H2 | H3 | aa[3], keep, pad
constexpr int RBA_GRAD_LD = 3 * RBA_H + 8;         // dP1 | dP2 | dP3 | dout[6], pad

__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float elu_grad_from_out(float h) { return h > 0.f ? 1.f : h + 1.f; }

// kornia 0.6.12 angle_axis_to_rotation_matrix: Rodrigues with w = aa / (theta + eps); first-order
// branch below theta^2 = eps.  R row-major.
__device__ void rodrigues(const float aa[3], float eps, float R[9]) {
    const float t2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (t2 > eps) {
        const float th = sqrtf(t2), inv = 1.0f / (th + eps);
        const float w0 = aa[0] * inv, w1 = aa[1] * inv, w2 = aa[2] * inv;
        const float c = cosf(th), s = sinf(th), k = 1.0f - c;
        R[0] = c + k * w0 * w0;      R[1] = -s * w2 + k * w0 * w1; R[2] = s * w1 + k * w0 * w2;
        R[3] = s * w2 + k * w1 * w0; R[4] = c + k * w1 * w1;       R[5] = -s * w0 + k * w1 * w2;
        R[6] = -s * w1 + k * w2 * w0; R[7] = s * w0 + k * w2 * w1; R[8] = c + k * w2 * w2;
    } else {
        R[0] = 1.f;    R[1] = -aa[2]; R[2] = aa[1];
        R[3] = aa[2];  R[4] = 1.f;    R[5] = -aa[0];
        R[6] = -aa[1]; R[7] = aa[0];  R[8] = 1.f;
    }
}

// dL/daa given G = dL/dR (row-major), for the same formula
__device__ void rodrigues_backward(const float aa[3], float eps, const float G[9], float daa[3]) {
    const float gS[3] = {G[7] - G[5], G[2] - G[6], G[3] - G[1]};
    const float t2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (!(t2 > eps)) { daa[0] = gS[0]; daa[1] = gS[1]; daa[2] = gS[2]; return; }
    const float th = sqrtf(t2), inv = 1.0f / (th + eps);
    const float w[3] = {aa[0] * inv, aa[1] * inv, aa[2] * inv};
    const float c = cosf(th), s = sinf(th), k = 1.0f - c;
    float Gw[3], Gtw[3], gWW = 0.f;
    for (int i = 0; i < 3; ++i) {
        Gw[i] = G[3 * i] * w[0] + G[3 * i + 1] * w[1] + G[3 * i + 2] * w[2];
        Gtw[i] = G[i] * w[0] + G[3 + i] * w[1] + G[6 + i] * w[2];
        gWW += Gw[i] * w[i];
    }
    const float dc = (G[0] + G[4] + G[8]) - gWW;
    const float ds = gS[0] * w[0] + gS[1] * w[1] + gS[2] * w[2];
    float dw[3], dth = -s * dc + c * ds;
    for (int i = 0; i < 3; ++i) {
        dw[i] = s * gS[i] + k * (Gw[i] + Gtw[i]);
        dth -= dw[i] * aa[i] * inv * inv;
    }
    for (int i = 0; i < 3; ++i) daa[i] = dw[i] * inv + dth * aa[i] / th;
}

// RBA.update_init_pose (reference model/rba.py:77-86): pose of a new keyframe -> init_c2w / init_t / init_r (rotation
// as angle-axis, kornia's rotation_matrix_to_angle_axis map evaluated from the antisymmetric part, the diagonal form only
// next to pi: the same expressions as remixfusion_amd/model/rba.py::rotation_matrix_to_angle_axis, which takes ~25 tiny
// ATen launches per keyframe).  One thread.
__global__ void rba_set_init_pose_kernel(const float* __restrict__ c2w, int cam, float* __restrict__ init_r,
                                         float* __restrict__ init_t, float* __restrict__ init_c2w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float m[16];
    for (int i = 0; i < 16; ++i) { m[i] = c2w[i]; init_c2w[(size_t)cam * 16 + i] = m[i]; }
    init_t[cam * 3] = m[3]; init_t[cam * 3 + 1] = m[7]; init_t[cam * 3 + 2] = m[11];
    const float v[3] = {0.5f * (m[9] - m[6]), 0.5f * (m[2] - m[8]), 0.5f * (m[4] - m[1])};
    const float s = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float c = 0.5f * (m[0] + m[5] + m[10] - 1.0f);
    const float theta = atan2f(s, c);
    const bool small = s < 1e-4f;
    const float scale = small ? 1.0f + theta * theta / 6.0f : theta / fmaxf(s, 1e-12f);
    float aa[3] = {v[0] * scale, v[1] * scale, v[2] * scale};
    if (small && c < 0.0f) {            // theta next to pi: |w_i| from R = 2 w w^T - I, signs from the row of the largest one
        float ax[3] = {sqrtf(fmaxf((m[0] + 1.0f) * 0.5f, 0.0f)), sqrtf(fmaxf((m[5] + 1.0f) * 0.5f, 0.0f)),
                       sqrtf(fmaxf((m[10] + 1.0f) * 0.5f, 0.0f))};
        int k = 0;
        if (ax[1] > ax[k]) k = 1;
        if (ax[2] > ax[k]) k = 2;
        float n2 = 0.0f;
        for (int i = 0; i < 3; ++i) {
            const float r = m[4 * k + i] + 1e-20f;
            ax[i] *= r > 0.0f ? 1.0f : (r < 0.0f ? -1.0f : 0.0f);
            n2 += ax[i] * ax[i];
        }
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        for (int i = 0; i < 3; ++i) aa[i] = ax[i] * inv * theta;
    }
    init_r[cam * 3] = aa[0]; init_r[cam * 3 + 1] = aa[1]; init_r[cam * 3 + 2] = aa[2];
}

// Per-frame pose bookkeeping of the tracker side (reference model/ROtracker.py:911-945 / mp_slam/tracker.py): store the frame's
// pose, and for a non-keyframe its pose relative to the newest keyframe, delta = c2w @ inverse(kf_c2w) (general 4x4 inverse, like torch's .inverse()).
__global__ void frame_pose_kernel(const float* __restrict__ c2w, const float* __restrict__ kf, float* __restrict__ est_out,
                                  float* __restrict__ rel_out) {
    // one wave, two memory round trips: the pose element of lane t and (scalar loads) the keyframe pose are fetched
    // together; every lane inverts the keyframe pose itself and takes its row of c2w from the other lanes' registers
    const int t = threadIdx.x & 15;
    const float cv = c2w[t];
    if (threadIdx.x < 16) est_out[t] = cv;
    if (!rel_out) return;
    // 4x4 inverse from its 2x2 sub-determinants (all indices static: registers, no scratch)
    const float a00 = kf[0], a01 = kf[1], a02 = kf[2], a03 = kf[3], a10 = kf[4], a11 = kf[5], a12 = kf[6], a13 = kf[7];
    const float a20 = kf[8], a21 = kf[9], a22 = kf[10], a23 = kf[11], a30 = kf[12], a31 = kf[13], a32 = kf[14], a33 = kf[15];
    const float s0 = a00 * a11 - a10 * a01, s1 = a00 * a12 - a10 * a02, s2 = a00 * a13 - a10 * a03;
    const float s3 = a01 * a12 - a11 * a02, s4 = a01 * a13 - a11 * a03, s5 = a02 * a13 - a12 * a03;
    const float c5 = a22 * a33 - a32 * a23, c4 = a21 * a33 - a31 * a23, c3 = a21 * a32 - a31 * a22;
    const float c2 = a20 * a33 - a30 * a23, c1 = a20 * a32 - a30 * a22, c0 = a20 * a31 - a30 * a21;
    const float id = 1.0f / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    const int c = t & 3;                 // the lane's column of the inverse: rows 0..3
    float i0, i1, i2, i3;
    if (c == 0) {
        i0 = (a11 * c5 - a12 * c4 + a13 * c3) * id;   i1 = (-a10 * c5 + a12 * c2 - a13 * c1) * id;
        i2 = (a10 * c4 - a11 * c2 + a13 * c0) * id;   i3 = (-a10 * c3 + a11 * c1 - a12 * c0) * id;
    } else if (c == 1) {
        i0 = (-a01 * c5 + a02 * c4 - a03 * c3) * id;  i1 = (a00 * c5 - a02 * c2 + a03 * c1) * id;
        i2 = (-a00 * c4 + a01 * c2 - a03 * c0) * id;  i3 = (a00 * c3 - a01 * c1 + a02 * c0) * id;
    } else if (c == 2) {
        i0 = (a31 * s5 - a32 * s4 + a33 * s3) * id;   i1 = (-a30 * s5 + a32 * s2 - a33 * s1) * id;
        i2 = (a30 * s4 - a31 * s2 + a33 * s0) * id;   i3 = (-a30 * s3 + a31 * s1 - a32 * s0) * id;
    } else {
        i0 = (-a21 * s5 + a22 * s4 - a23 * s3) * id;  i1 = (a20 * s5 - a22 * s2 + a23 * s1) * id;
        i2 = (-a20 * s4 + a21 * s2 - a23 * s0) * id;  i3 = (a20 * s3 - a21 * s1 + a22 * s0) * id;
    }
    const int r4 = t & 12;
    const float r0 = __shfl(cv, r4), r1 = __shfl(cv, r4 + 1), r2 = __shfl(cv, r4 + 2), r3 = __shfl(cv, r4 + 3);
    if (threadIdx.x < 16) rel_out[t] = r0 * i0 + r1 * i1 + r2 * i2 + r3 * i3;
}

struct RbaW {
    const float *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3;
};
struct RbaG {
    float *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3;
};

// y[j] = sum_i W[j][i] v[i] with FOUR threads per output row: thread (j, q) takes the float4s 16 i + 4 q of the row, so a
// quad reads 64 contiguous bytes and a wave touches 16 rows per load instruction instead of 64 (a thread that walks its own
// row made a 256 x 256 layer cost 7 us), and all sixteen loads of a thread are independent.  v: 256 inputs in LDS (16-byte
// aligned).  The result is valid in all four threads of the quad.
__device__ __forceinline__ float quad_rows_dot(const float* __restrict__ Wm, const float* v, int j, int q) {
    const float4* __restrict__ row = reinterpret_cast<const float4*>(Wm + (size_t)j * RBA_H);
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(v);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 a = row[4 * i + q], x = v4[4 * i + q];
        acc = fmaf(a.w, x.w, fmaf(a.z, x.z, fmaf(a.y, x.y, fmaf(a.x, x.x, acc))));
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    return acc;
}

constexpr int RBA_FWD_THREADS = 4 * RBA_H;

__global__ __launch_bounds__(RBA_FWD_THREADS) void rba_forward_kernel(RbaW W, const float* __restrict__ init_r,
                                                                      const float* __restrict__ init_t,
                                                                      const int64_t* __restrict__ cam_ids, int num_cams, float scale,
                                                                      float eps, float* __restrict__ poses, float* __restrict__ acts) {
    __shared__ __attribute__((aligned(16))) float sa[RBA_H], sb[RBA_H];
    __shared__ float sin7[8], sout[8];
    const int k = blockIdx.x, t = threadIdx.x, j = t >> 2, q = t & 3;
    const int64_t id = cam_ids[k];
    float* __restrict__ row = acts + (size_t)k * RBA_ACT_LD;
    if (t < 8) {
        float v = 0.f;
        if (t == 0) v = ((float)id / (float)num_cams) * 2.0f - 1.0f;
        else if (t < 4) v = init_r[id * 3 + (t - 1)];
        else if (t < 7) v = init_t[id * 3 + (t - 4)];
        sin7[t] = v;
        row[t] = v;
    }
    __syncthreads();
    if (q == 0) {
        float h = W.b0[j];
        for (int i = 0; i < RBA_IN; ++i) h = fmaf(W.w0[j * RBA_IN + i], sin7[i], h);
        h = elu(h);
        sa[j] = h; row[8 + j] = h;
    }
    __syncthreads();
    float h = elu(W.b1[j] + quad_rows_dot(W.w1, sa, j, q));
    if (q == 0) { sb[j] = h; row[8 + RBA_H + j] = h; }
    __syncthreads();
    h = elu(W.b2[j] + quad_rows_dot(W.w2, sb, j, q));
    if (q == 0) { sa[j] = h; row[8 + 2 * RBA_H + j] = h; }          // sa was last read before the previous barrier
    __syncthreads();
    if (j < RBA_OUT) {
        const float y = quad_rows_dot(W.w3, sa, j, q);
        if (q == 0) {
            const float keep = id != 0 ? 1.0f : 0.0f;    // camera 0 is the gauge (reference rba.py:90-91)
            sout[j] = (W.b3[j] + y) * scale * keep;
        }
    }
    __syncthreads();
    if (t == 0) {
        const float aa[3] = {sout[0] + sin7[1], sout[1] + sin7[2], sout[2] + sin7[3]};
        float R[9];
        rodrigues(aa, eps, R);
        float* __restrict__ P = poses + (size_t)k * 16;
        for (int r = 0; r < 3; ++r) {
            P[4 * r] = R[3 * r]; P[4 * r + 1] = R[3 * r + 1]; P[4 * r + 2] = R[3 * r + 2];
            P[4 * r + 3] = sout[3 + r] + sin7[4 + r];
        }
        P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
        float* tail = row + 8 + 3 * RBA_H;
        tail[0] = aa[0]; tail[1] = aa[1]; tail[2] = aa[2]; tail[3] = id != 0 ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(RBA_H) void rba_backward_kernel(RbaW W, const float* __restrict__ acts,
                                                             const float* __restrict__ dposes, float scale, float eps,
                                                             float* __restrict__ grads) {
    __shared__ float sa[RBA_H], sb[RBA_H], sd[8];
    const int k = blockIdx.x, j = threadIdx.x;
    const float* __restrict__ row = acts + (size_t)k * RBA_ACT_LD;
    float* __restrict__ g = grads + (size_t)k * RBA_GRAD_LD;
    if (j == 0) {
        const float* __restrict__ D = dposes + (size_t)k * 16;
        const float* tail = row + 8 + 3 * RBA_H;
        const float aa[3] = {tail[0], tail[1], tail[2]};
        const float G[9] = {D[0], D[1], D[2], D[4], D[5], D[6], D[8], D[9], D[10]};
        float daa[3];
        rodrigues_backward(aa, eps, G, daa);
        const float f = scale * tail[3];
        sd[0] = daa[0] * f; sd[1] = daa[1] * f; sd[2] = daa[2] * f;
        sd[3] = D[3] * f; sd[4] = D[7] * f; sd[5] = D[11] * f;
        for (int o = 0; o < RBA_OUT; ++o) g[3 * RBA_H + o] = sd[o];
    }
    __syncthreads();
    float v = 0.f;
    for (int o = 0; o < RBA_OUT; ++o) v = fmaf(W.w3[o * RBA_H + j], sd[o], v);
    v *= elu_grad_from_out(row[8 + 2 * RBA_H + j]);
    sa[j] = v; g[2 * RBA_H + j] = v;                 // dP3
    __syncthreads();
    v = 0.f;
    for (int i = 0; i < RBA_H; ++i) v = fmaf(W.w2[(size_t)i * RBA_H + j], sa[i], v);
    v *= elu_grad_from_out(row[8 + RBA_H + j]);
    sb[j] = v; g[RBA_H + j] = v;                     // dP2
    __syncthreads();
    v = 0.f;
    for (int i = 0; i < RBA_H; ++i) v = fmaf(W.w1[(size_t)i * RBA_H + j], sb[i], v);
    v *= elu_grad_from_out(row[8 + j]);
    g[j] = v;                                        // dP1
}

// parameter gradients, overwritten:  dW_l[i][j] = sum_k dP_l[k][i] * in_l[k][j],  db_l[i] = sum_k dP_l[k][i],  k over the cameras IN
// ORDER (deterministic, and the same sums whatever the tiling).
// Rounds 2-6 ran one thread per parameter element with a loop over the cameras: two loads per camera and element, 133 k threads --
// 4.8 us at 5 keyframes, 57 us at 190, 110+ at 380: a long stream's pose iteration grew by half (profiles/r6_notes.md 7.5).
// Now a block owns WG_TI rows i of one layer and all its columns j (thread = column): per camera ONE coalesced load of the layer's
// input row (WG_UB cameras' loads in flight, the next batch's issued before this one is added) and the WG_TI pre-activation
// gradients from an LDS panel the block fetched WG_KC cameras at a time -- a fraction of the loads, a round trip per 32 cameras.
// Rows of the panel beyond K are zero: fmaf(0, a, acc) == acc, so no tail handling, and the sums are the element-per-thread
// loop's bit for bit.
constexpr int WG_TI = 4, WG_KC = 128, WG_UB = 32;
constexpr int WG_TILES_H = RBA_H / WG_TI;                       // row tiles of a 256-row layer
constexpr int WG_TILES_OUT = (RBA_OUT + WG_TI - 1) / WG_TI;     // ... of the 6-row output layer
constexpr int WG_BLOCKS = 3 * WG_TILES_H + WG_TILES_OUT;        // w0, w1, w2, then w3
static_assert(WG_TI == 4 && RBA_H % WG_TI == 0 && WG_KC % WG_UB == 0 && WG_KC * WG_TI % 256 == 0, "rba_wgrad tiling (a panel row is one float4)");

__global__ __launch_bounds__(256) void rba_wgrad_kernel(const float* __restrict__ acts, const float* __restrict__ grads, int K,
                                                        RbaG G) {
    __shared__ __attribute__((aligned(16))) float gs[WG_KC][WG_TI];
    const int layer = min((int)blockIdx.x / WG_TILES_H, 3), tile = (int)blockIdx.x - layer * WG_TILES_H;      // (w3's tiles: 0, 1)
    const int rows = layer == 3 ? RBA_OUT : RBA_H;
    const int n_in = layer == 0 ? RBA_IN : RBA_H;
    const int g_off = layer * RBA_H;                              // dP1 | dP2 | dP3 | dout
    const int a_off = layer == 0 ? 0 : 8 + (layer - 1) * RBA_H;   // inp[8] | H1 | H2 | H3
    float* __restrict__ out_w = layer == 0 ? G.w0 : layer == 1 ? G.w1 : layer == 2 ? G.w2 : G.w3;
    float* __restrict__ out_b = layer == 0 ? G.b0 : layer == 1 ? G.b1 : layer == 2 ? G.b2 : G.b3;
    const int i0 = tile * WG_TI, t = threadIdx.x;
    const bool active = t < n_in;
    const float* __restrict__ aq = acts + a_off + (active ? t : 0);
    float acc[WG_TI], bacc[WG_TI];
#pragma unroll
    for (int r = 0; r < WG_TI; ++r) { acc[r] = 0.f; bacc[r] = 0.f; }
    for (int kc = 0; kc < K; kc += WG_KC) {
        const int nk = min(WG_KC, K - kc);
#pragma unroll
        for (int q = 0; q < WG_KC * WG_TI / 256; ++q) {
            const int idx = t + q * 256, k = idx / WG_TI, r = idx % WG_TI;
            const bool ok = k < nk && i0 + r < rows;
            const float g = grads[(size_t)(kc + min(k, nk - 1)) * RBA_GRAD_LD + g_off + min(i0 + r, rows - 1)];      // (unconditional load)
            gs[k][r] = ok ? g : 0.f;
        }
        __syncthreads();
        float av[WG_UB], an[WG_UB];
#pragma unroll
        for (int u = 0; u < WG_UB; ++u) av[u] = aq[(size_t)min(kc + u, K - 1) * RBA_ACT_LD];
        for (int k0 = 0; k0 < nk; k0 += WG_UB) {
            const bool more = k0 + WG_UB < nk;                    // block-uniform
            if (more) {
#pragma unroll
                for (int u = 0; u < WG_UB; ++u) an[u] = aq[(size_t)min(kc + k0 + WG_UB + u, K - 1) * RBA_ACT_LD];
            }
#pragma unroll
            for (int u = 0; u < WG_UB; ++u) {
                const float4 g = *reinterpret_cast<const float4*>(&gs[k0 + u][0]);
                const float gr[WG_TI] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int r = 0; r < WG_TI; ++r) { acc[r] = fmaf(gr[r], av[u], acc[r]); bacc[r] += gr[r]; }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < WG_UB; ++u) av[u] = an[u];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < WG_TI; ++r) {
        if (i0 + r >= rows) break;
        if (active && out_w) out_w[(size_t)(i0 + r) * n_in + t] = acc[r];
        if (t == 0 && out_b) out_b[i0 + r] = bacc[r];
    }
}

// ---- the tail of a pose iteration in ONE launch: d rays -> d poses -> pose-MLP backward ------------------------------------
// What ray_grad_reduce_kernel (rfx_ba.hip), pose_grad_kernel (rfx_render.hip) and rba_backward_kernel above do one after the
// other: all three are block = camera (or trivially regrouped that way), each is a few microseconds of latency, and a
// dependent kernel boundary costs ~4.7 us here.  Block = camera k, 1 024 threads:
//   1. the rays riding on pose k, in ray order (ballot + prefix over the block, 1 024 rays per round);
//   2. sixteen lanes per ray: d rays_o = sum_s dx / extent, d rays_d = sum_s z dx / extent (scene_rep.py:388,443), folded straight into
//      dposes[k] (rotation rows += g_d (x) d_cam, translation += g_o); the groups' sums combined in a fixed order: deterministic;
//   3. the MLP backward with four threads per hidden unit (each a quarter of the 256-term column dot products).
// Sums are grouped differently from the three separate kernels: equal to rounding, not bit-identical.
constexpr int PCB_THREADS = 1024, PCB_MAX_RAYS = 8192;
struct PoseChainK {
    const float *dx, *z, *d_cam; const int* pose_idx; int64_t n; int S; float ex, ey, ez;
};

__global__ __launch_bounds__(PCB_THREADS) void pose_chain_backward_kernel(PoseChainK p, RbaW W, const float* __restrict__ acts,
                                                                         float scale, float eps, float* __restrict__ grads,
                                                                         float* __restrict__ dposes) {
    __shared__ int list[PCB_MAX_RAYS];
    __shared__ int wcnt[PCB_THREADS / 64];
    __shared__ float accs[PCB_THREADS / 16][12];
    __shared__ float sa[RBA_H], sb[RBA_H], part[4][RBA_H], sd[8], D[12];
    const int k = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    // 1. this camera's rays
    int total = 0;
    for (int64_t r0 = 0; r0 < p.n; r0 += PCB_THREADS) {
        const int64_t i = r0 + t;
        const bool hit = i < p.n && p.pose_idx[i] == k;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, round_total = 0;
#pragma unroll
        for (int w = 0; w < PCB_THREADS / 64; ++w) { const int c = wcnt[w]; round_total += c; if (w < wv) before += c; }
        if (hit) list[total + before + __popcll(bal & ((1ull << lane) - 1ull))] = (int)i;
        total += round_total;
        __syncthreads();
    }
    // 2. ray gradients -> dposes[k]: sixteen lanes per ray (four rays per wave, 64 per block in flight: a ray is a chain of
    //    dependent latencies, and a camera has only a few hundred of them)
    float a[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) a[q] = 0.f;
    const int grp = t >> 4, sub = t & 15;
    for (int e = grp; e < total; e += PCB_THREADS / 16) {
        const int64_t ray = list[e];
        float s6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = sub; j < p.S; j += 16) {
            const float* g = p.dx + (ray * p.S + j) * 3;
            const float zz = p.z[ray * p.S + j];
            const float px = g[0] / p.ex, py = g[1] / p.ey, pz = g[2] / p.ez;
            s6[0] += px; s6[1] += py; s6[2] += pz;
            s6[3] += px * zz; s6[4] += py * zz; s6[5] += pz * zz;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s6[q] += __shfl_xor(s6[q], o);
        }
        const float dc[3] = {p.d_cam[ray * 3], p.d_cam[ray * 3 + 1], p.d_cam[ray * 3 + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a[4 * r] += s6[3 + r] * dc[0]; a[4 * r + 1] += s6[3 + r] * dc[1]; a[4 * r + 2] += s6[3 + r] * dc[2];
            a[4 * r + 3] += s6[r];
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) accs[grp][q] = a[q];
    }
    __syncthreads();
    if (t < 16) {
        float v = 0.f;
        if (t < 12) {
            for (int w = 0; w < PCB_THREADS / 16; ++w) v += accs[w][t];
            D[t] = v;
        }
        dposes[(size_t)k * 16 + t] = v;
    }
    __syncthreads();
    // 3. pose-MLP backward (rba_backward_kernel's arithmetic)
    const float* __restrict__ row = acts + (size_t)k * RBA_ACT_LD;
    float* __restrict__ g = grads + (size_t)k * RBA_GRAD_LD;
    const int j = t & (RBA_H - 1), q4 = t >> 8;
    if (t == 0) {
        const float* tail = row + 8 + 3 * RBA_H;
        const float aa[3] = {tail[0], tail[1], tail[2]};
        const float G[9] = {D[0], D[1], D[2], D[4], D[5], D[6], D[8], D[9], D[10]};
        float daa[3];
        rodrigues_backward(aa, eps, G, daa);
        const float f = scale * tail[3];
        sd[0] = daa[0] * f; sd[1] = daa[1] * f; sd[2] = daa[2] * f;
        sd[3] = D[3] * f; sd[4] = D[7] * f; sd[5] = D[11] * f;
        for (int o = 0; o < RBA_OUT; ++o) g[3 * RBA_H + o] = sd[o];
    }
    __syncthreads();
    if (q4 == 0) {
        float v = 0.f;
        for (int o = 0; o < RBA_OUT; ++o) v = fmaf(W.w3[o * RBA_H + j], sd[o], v);
        v *= elu_grad_from_out(row[8 + 2 * RBA_H + j]);
        sa[j] = v; g[2 * RBA_H + j] = v;                 // dP3
    }
    __syncthreads();
    {
        float v = 0.f;
#pragma unroll 8
        for (int i = q4 * (RBA_H / 4); i < (q4 + 1) * (RBA_H / 4); ++i) v = fmaf(W.w2[(size_t)i * RBA_H + j], sa[i], v);
        part[q4][j] = v;
    }
    __syncthreads();
    if (q4 == 0) {
        float v = ((part[0][j] + part[1][j]) + part[2][j]) + part[3][j];
        v *= elu_grad_from_out(row[8 + RBA_H + j]);
        sb[j] = v; g[RBA_H + j] = v;                     // dP2
    }
    __syncthreads();
    {
        float v = 0.f;
#pragma unroll 8
        for (int i = q4 * (RBA_H / 4); i < (q4 + 1) * (RBA_H / 4); ++i) v = fmaf(W.w1[(size_t)i * RBA_H + j], sb[i], v);
        part[q4][j] = v;
    }
    __syncthreads();
    if (q4 == 0) {
        float v = ((part[0][j] + part[1][j]) + part[2][j]) + part[3][j];
        v *= elu_grad_from_out(row[8 + j]);
        g[j] = v;                                        // dP1
    }
}

// rfx_pose_grad (fed the ray gradients of dx01) + rfx_rba_backward in two launches instead of four; false: not applicable
// (too many rays for the block's list), the caller issues the separate stages
int pose_chain_backward(const float* dx01, const float* z_vals, const float* d_cam, const int32_t* pose_idx, int64_t n, int S,
                        const double bbox[6], int K, float* dposes16, const rfx_rba_params* prm, const float* acts, float scale,
                        const rfx_rba_grads* gr, float* workspace, rfx_stream stream, int* done) {
    *done = 0;
    if (n <= 0 || n > PCB_MAX_RAYS || S <= 0 || K <= 0) return RFX_OK;
    if (!prm || prm->hidden != RBA_H || !prm->w0 || !prm->b0 || !prm->w1 || !prm->b1 || !prm->w2 || !prm->b2 || !prm->w3 || !prm->b3)
        return RFX_ERR_ARG;
    if (!dx01 || !z_vals || !d_cam || !pose_idx || !bbox || !dposes16 || !acts || !gr || !workspace) return RFX_ERR_ARG;
    PoseChainK pk{dx01, z_vals, d_cam, pose_idx, n, S, (float)(bbox[1] - bbox[0]), (float)(bbox[3] - bbox[2]), (float)(bbox[5] - bbox[4])};
    RbaW W{prm->w0, prm->b0, prm->w1, prm->b1, prm->w2, prm->b2, prm->w3, prm->b3};
    hipLaunchKernelGGL(pose_chain_backward_kernel, dim3((unsigned)K), dim3(PCB_THREADS), 0, as_stream(stream), pk, W, acts, scale, 1e-6f,
                       workspace, dposes16);
    RFX_LAUNCH_CHECK();
    RbaG G{gr->w0, gr->b0, gr->w1, gr->b1, gr->w2, gr->b2, gr->w3, gr->b3};
    hipLaunchKernelGGL(rba_wgrad_kernel, dim3(WG_BLOCKS), dim3(256), 0, as_stream(stream), acts, workspace, K, G);
    RFX_LAUNCH_CHECK();
    *done = 1;
    return RFX_OK;
}

}  // namespace rfx

using namespace rfx;

extern "C" {

size_t rfx_rba_acts_floats(int64_t K) { return K > 0 ? (size_t)K * RBA_ACT_LD : 0; }
size_t rfx_rba_grads_floats(int64_t K) { return K > 0 ? (size_t)K * RBA_GRAD_LD : 0; }

static bool rba_params_ok(const rfx_rba_params* p) {
    return p && p->w0 && p->b0 && p->w1 && p->b1 && p->w2 && p->b2 && p->w3 && p->b3 && p->hidden == RBA_H;
}

int rfx_frame_pose(const float* c2w16, const float* kf_c2w16, float* est_c2w16, float* rel_c2w16, rfx_stream stream) {
    if (!c2w16 || !est_c2w16 || (rel_c2w16 && !kf_c2w16)) return RFX_ERR_ARG;
    hipLaunchKernelGGL(frame_pose_kernel, dim3(1), dim3(64), 0, as_stream(stream), c2w16, kf_c2w16, est_c2w16, rel_c2w16);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_rba_set_init_pose(const float* c2w16, int cam_id, int num_cams, float* init_r, float* init_t, float* init_c2w16,
                          rfx_stream stream) {
    if (!c2w16 || !init_r || !init_t || !init_c2w16 || cam_id < 0 || cam_id >= num_cams) return RFX_ERR_ARG;
    hipLaunchKernelGGL(rba_set_init_pose_kernel, dim3(1), dim3(64), 0, as_stream(stream), c2w16, cam_id, init_r, init_t, init_c2w16);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_rba_forward(const rfx_rba_params* p, const float* init_r, const float* init_t, const int64_t* cam_ids, int64_t K,
                    int num_cams, float scale, float* poses16, float* acts, rfx_stream stream) {
    if (K == 0) return RFX_OK;
    if (p && p->hidden != RBA_H) return RFX_ERR_UNSUPPORTED;
    if (!rba_params_ok(p) || !init_r || !init_t || !cam_ids || !poses16 || !acts || K < 0 || num_cams <= 0) return RFX_ERR_ARG;
    RbaW W{p->w0, p->b0, p->w1, p->b1, p->w2, p->b2, p->w3, p->b3};
    hipLaunchKernelGGL(rba_forward_kernel, dim3((unsigned)K), dim3(RBA_FWD_THREADS), 0, as_stream(stream), W, init_r, init_t, cam_ids,
                       num_cams, scale, 1e-6f, poses16, acts);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_rba_backward(const rfx_rba_params* p, const float* acts, int64_t K, const float* dposes16, float scale,
                     const rfx_rba_grads* g, float* workspace, rfx_stream stream) {
    if (K == 0) return RFX_OK;
    if (p && p->hidden != RBA_H) return RFX_ERR_UNSUPPORTED;
    if (!rba_params_ok(p) || !acts || !dposes16 || !g || !workspace || K < 0) return RFX_ERR_ARG;
    RbaW W{p->w0, p->b0, p->w1, p->b1, p->w2, p->b2, p->w3, p->b3};
    hipLaunchKernelGGL(rba_backward_kernel, dim3((unsigned)K), dim3(RBA_H), 0, as_stream(stream), W, acts, dposes16, scale, 1e-6f,
                       workspace);
    RFX_LAUNCH_CHECK();
    RbaG G{g->w0, g->b0, g->w1, g->b1, g->w2, g->b2, g->w3, g->b3};
    hipLaunchKernelGGL(rba_wgrad_kernel, dim3(WG_BLOCKS), dim3(256), 0, as_stream(stream), acts, workspace, (int)K, G);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"
