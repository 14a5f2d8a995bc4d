// micro-benchmark: does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) drop the in-stream barrier on gfx950?
// A: 8 blocks spin ~200 us and write their end clock.  B (one block) writes its start clock.  B is launched right behind A on the
// same stream, once normally and once with the flag: with the barrier gone B starts while A spins.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64();
}
__global__ void stamp(unsigned long long* out) { if (threadIdx.x == 0) out[0] = wall_clock64(); }
int main() {
    unsigned long long *a, *b;
    hipMalloc(&a, 64 * 8); hipMalloc(&b, 8);
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, 20000ull, a);      // 100 MHz clock: 200 us
            if (mode == 0) hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, st, b);
            else hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, b);
            hipStreamSynchronize(st);
            unsigned long long ha[8], hb;
            hipMemcpy(ha, a, 64, hipMemcpyDeviceToHost); hipMemcpy(&hb, b, 8, hipMemcpyDeviceToHost);
            unsigned long long end = 0; for (int i = 0; i < 8; ++i) end = ha[i] > end ? ha[i] : end;
            printf("%s: B started %.1f us %s the end of A\n", mode ? "any-order" : "in-order ", (double)((long long)hb - (long long)end) / 100.0,
                   hb >= end ? "after" : "BEFORE");
        }
    }
    return 0;
}
