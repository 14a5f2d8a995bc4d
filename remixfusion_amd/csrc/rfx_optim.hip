// rfx_optim.hip -- Adam step for the mapper's two optimizers, one launch for all tensors of an optimizer.
// Reference: torch.optim.Adam(betas=(0.9, 0.99)) built in mp_slam/slam.py:271-286 and stepped in
// mp_slam/mapper.py:416-418, 497-499 (torch 1.13 _single_tensor_adam: L2 weight decay folded into the gradient,
// no amsgrad).  The hash table is one 1.6e6 ... 4e7 element tensor: a multi-tensor launch that deals one 64 K chunk
// to a block fills 26 of the 256 CUs at T = 2^16; here every tensor is cut into 4 K-element pieces so the step runs
// at HBM speed (28 B / element).
#include "rfx_common.h"

namespace rfx {

constexpr int ADAM_MAX = RFX_ADAM_MAX_TENSORS;
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_PIECE = 4096;      // elements per block

struct AdamK {
    rfx_adam_tensor t[ADAM_MAX];
    int first_block[ADAM_MAX + 1];
    int count;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const rfx_adam_tensor& a) {
    // expression tree of _single_tensor_adam; madd() where the reference's elementwise kernels contract
    if (a.weight_decay != 0.f) g = madd(p, a.weight_decay, g);            // grad.add(param, alpha=wd)
    m = madd(g, a.one_minus_beta1, m * a.beta1);                         // exp_avg.mul_(b1).add_(grad, alpha=1-b1)
    v = madd(a.one_minus_beta2 * g, g, v * a.beta2);                     // exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2)
    const float denom = sqrtf(v) / a.bias_correction2_sqrt + a.eps;      // (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps)
    p = madd(a.neg_step_size, m / denom, p);                             // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(ADAM_THREADS) void adam_step_kernel(AdamK k) {
    int ti = 0;
    while (ti + 1 < k.count && (int)blockIdx.x >= k.first_block[ti + 1]) ++ti;
    const rfx_adam_tensor a = k.t[ti];
    const int64_t base = (int64_t)(blockIdx.x - k.first_block[ti]) * ADAM_PIECE;
    const int64_t end = min(a.n, base + ADAM_PIECE);
    const bool vec = ((a.n & 3) == 0) && ((((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) & 15) == 0);
    if (vec) {
        for (int64_t i = base + (int64_t)threadIdx.x * 4; i < end; i += ADAM_THREADS * 4) {
            float4 p = *reinterpret_cast<float4*>(a.param + i);
            const float4 g = *reinterpret_cast<const float4*>(a.grad + i);
            float4 m = *reinterpret_cast<float4*>(a.exp_avg + i);
            float4 v = *reinterpret_cast<float4*>(a.exp_avg_sq + i);
            adam_one(p.x, g.x, m.x, v.x, a); adam_one(p.y, g.y, m.y, v.y, a);
            adam_one(p.z, g.z, m.z, v.z, a); adam_one(p.w, g.w, m.w, v.w, a);
            *reinterpret_cast<float4*>(a.param + i) = p;
            *reinterpret_cast<float4*>(a.exp_avg + i) = m;
            *reinterpret_cast<float4*>(a.exp_avg_sq + i) = v;
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += ADAM_THREADS) {
            float p = a.param[i], m = a.exp_avg[i], v = a.exp_avg_sq[i];
            adam_one(p, a.grad[i], m, v, a);
            a.param[i] = p; a.exp_avg[i] = m; a.exp_avg_sq[i] = v;
        }
    }
}

}  // namespace rfx

using namespace rfx;

extern "C" {

size_t rfx_adam_tensor_bytes(void) { return sizeof(rfx_adam_tensor); }

int rfx_adam_step(const rfx_adam_tensor* tensors, int count, rfx_stream stream) {
    if (count == 0) return RFX_OK;
    if (!tensors || count < 0 || count > ADAM_MAX) return RFX_ERR_ARG;
    AdamK k;
    k.count = 0;
    int64_t blocks = 0;
    for (int i = 0; i < count; ++i) {
        const rfx_adam_tensor& a = tensors[i];
        if (a.n < 0 || (a.n > 0 && (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq))) return RFX_ERR_ARG;
        if (!(a.bias_correction2_sqrt > 0.f)) return RFX_ERR_ARG;          // step >= 1
        if (a.n == 0) continue;
        k.t[k.count] = a;
        k.first_block[k.count] = (int)blocks;
        blocks += (a.n + ADAM_PIECE - 1) / ADAM_PIECE;
        if (blocks > 0x7fffffff) return RFX_ERR_UNSUPPORTED;
        ++k.count;
    }
    if (k.count == 0) return RFX_OK;
    k.first_block[k.count] = (int)blocks;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, as_stream(stream), k);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"
