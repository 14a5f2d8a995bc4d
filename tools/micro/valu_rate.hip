// dev micro-benchmark (inline-asm v_fma_f32, so that the compiler cannot pack them): VALU issue rate per SIMD with 1, 2, 4 waves per SIMD (independent v_fma_f32 streams), and the same
// beside fp32 MFMAs.  build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MFMA>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x16 acc = {0};
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
            F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
        }
        if (MFMA) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, a3, acc, 0, 0, 0); }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MFMA>
static void run(int waves_per_simd) {
    float* d; hipMalloc(&d, 256 * 1024 * 64 * sizeof(float));
    const int iters = 20000, blocks = 256, threads = 256 * waves_per_simd;      // one block per CU: threads/64 waves, 4 SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MFMA>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MFMA>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double valu = (double)iters * 64, per_wave_cycles = ms * 1e-3 * 2.4e9;
    printf("mfma=%d waves/SIMD=%d: %.3f ms -> %.2f cycles (at 2.4 GHz) per VALU instr per wave, %.2f per SIMD-slot%s\n", MFMA, waves_per_simd, ms,
           per_wave_cycles / valu, per_wave_cycles / valu / waves_per_simd, MFMA ? " (+2 MFMA 32x32x2 per 64 VALU)" : "");
    hipFree(d);
}
int main() { for (int w : {1, 2, 4}) run<0>(w); for (int w : {1, 2, 4}) run<1>(w); return 0; }
