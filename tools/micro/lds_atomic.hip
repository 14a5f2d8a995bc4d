// micro-benchmark: cost of ds_add_f32 on gfx950 as a function of the active-lane fraction and address pattern.
// 1024-thread blocks (one per CU, 128 KB LDS like the scatter kernel); each thread issues `iters` atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T>
__global__ __launch_bounds__(1024) void ki(float* out, int iters, unsigned keep_mask) {
    extern __shared__ unsigned char raw_[];
    T* acc = reinterpret_cast<T*>(raw_);
    const int n = 131072 / sizeof(T);
    for (int i = threadIdx.x; i < n; i += 1024) acc[i] = 0;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const unsigned a = (h >> 15) & (n - 1);
        if (((h >> 7) & keep_mask) == 0) atomicAdd(&acc[a], (T)(h & 255));
    }
    __syncthreads();
    T s = 0;
    for (int i = threadIdx.x; i < n; i += 1024) s += acc[i];
    if (s == (T)12345) out[0] = 1.f;
}

__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned keep_mask, int mode) {
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < 32768; i += 1024) acc[i] = 0.f;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        unsigned a = (mode == 0) ? (h >> 17) & 32767u          // random address
                   : (mode == 1) ? ((h >> 17) & 32736u) | (lane & 31)   // random row, lane-consecutive (conflict-free)
                                 : ((h >> 17) & 511u);            // few addresses: heavy same-address collisions
        if (((h >> 7) & keep_mask) == 0) atomicAdd(&acc[a], 1.0f);
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 32768; i += 1024) s += acc[i];
    if (s == -1.f) out[0] = s;
}
// eight LDS operations per loop trip (the single-operation loops above bottom out at ~8.7 clk of loop overhead per operation):
// OP 0 = ds_add_f64, 1 = ds_write_b64, 2 = ds_write_b128, 3 = ds_add_u64, each to a random address, `keep_mask` thins the lanes
template <int OP>
__global__ __launch_bounds__(1024) void k8(float* out, int iters, unsigned keep_mask) {
    extern __shared__ unsigned char raw8[];
    double* acc = reinterpret_cast<double*>(raw8);
    for (int i = threadIdx.x; i < 16384; i += 1024) acc[i] = 0;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned hh = h ^ (0x9e3779b9u * (u + 1));
            const unsigned a = (hh >> 15) & 16383u;
            if (((hh >> 7) & keep_mask) == 0) {
                if (OP == 0) atomicAdd(&acc[a], (double)(hh & 255));
                else if (OP == 1) acc[a] = (double)(hh & 255);
                else if (OP == 2) *reinterpret_cast<double2*>(&acc[a & ~1u]) = make_double2((double)(hh & 255), 1.0);
                else atomicAdd(reinterpret_cast<unsigned long long*>(&acc[a]), (unsigned long long)(hh & 255));
            }
        }
    }
    __syncthreads();
    double s = 0;
    for (int i = threadIdx.x; i < 16384; i += 1024) s += acc[i];
    if (s == 12345.0) out[0] = 1.f;
}

int main() {
    float* out; hipMalloc(&out, 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256;
    for (int mode = 0; mode < 3; ++mode)
        for (unsigned km : {0u, 1u, 3u, 7u, 31u}) {
            k<<<blocks, 1024, 131072>>>(out, iters, km, mode);
            hipEventRecord(e0);
            k<<<blocks, 1024, 131072>>>(out, iters, km, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_cu = 16.0 * iters;               // wave-level ds_add instructions per CU
            printf("mode %d active 1/%u: %.3f ms -> %.1f clk per wave-instruction per CU (at 2.1 GHz), %.2f lane-ops/clk/CU\n", mode, km + 1, ms,
                   ms * 1e-3 * 2.1e9 / instr_per_cu, 1024.0 * iters / (km + 1) / (ms * 1e-3 * 2.1e9));
        }
    auto bench_int = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        for (unsigned km : {0u, 3u, 31u}) {
            kern<<<blocks, 1024, 131072>>>(out, iters, km);
            hipEventRecord(e0);
            kern<<<blocks, 1024, 131072>>>(out, iters, km);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s active 1/%u: %.3f ms -> %.1f clk per wave-instruction per CU, %.2f lane-ops/clk/CU\n", name, km + 1, ms,
                   ms * 1e-3 * 2.1e9 / (16.0 * iters), 1024.0 * iters / (km + 1) / (ms * 1e-3 * 2.1e9));
        }
    };
    auto bench8 = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        for (unsigned km : {0u, 1u, 3u, 7u, 15u}) {
            kern<<<blocks, 1024, 131072>>>(out, iters / 8, km);
            hipEventRecord(e0);
            kern<<<blocks, 1024, 131072>>>(out, iters / 8, km);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("x8 %s active 1/%u: %.3f ms -> %.1f clk per wave-instruction per CU\n", name, km + 1, ms, ms * 1e-3 * 2.1e9 / (16.0 * iters));
        }
    };
    bench8(k8<0>, "ds_add_f64");
    bench8(k8<3>, "ds_add_u64");
    bench8(k8<1>, "ds_write_b64");
    bench8(k8<2>, "ds_write_b128");
    bench_int(ki<float>, "f32 (typed kernel)");
    bench_int(ki<unsigned>, "u32");
    bench_int(ki<unsigned long long>, "u64");
    bench_int(ki<double>, "f64");
    return 0;
}
