// dev micro-benchmark: what a boundary between two DEPENDENT kernels costs the GPU -- launched one by one on a stream, against
// the same chain captured into a hipGraph.  Kernels of 256 blocks x 256 threads that do ~2 us of work each.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/graph_gap tools/micro/graph_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 1; } } while (0)

__global__ void work(float* p, int iters) {
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 1e-7f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
    const int N = 200, blocks = 256;
    float* d;
    CK(hipMalloc(&d, blocks * 256 * sizeof(float)));
    CK(hipMemset(d, 0, blocks * 256 * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int iters : {0, 500, 4000}) {
        // one kernel alone (its own duration)
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, st, d, iters);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, st, d, iters);
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        float one = 0; CK(hipEventElapsedTime(&one, a, b));
        // chain on the stream
        float best_s = 1e9f, best_g = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a, st));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, st, d, iters);
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            best_s = ms < best_s ? ms : best_s;
        }
        // the same chain as a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, st, d, iters);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0) best_g = ms < best_g ? ms : best_g;
        }
        printf("iters %5d: one kernel (event to event) %7.2f us; chain of %d on a stream %7.2f us per kernel; as a graph %7.2f us per kernel\n",
               iters, one * 1e3f, N, best_s * 1e3f / N, best_g * 1e3f / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
