// dev probe (gfx950, run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`): what does FETCH_SIZE count on V1's access shape?
// The guide calibrates the counter on wide coalesced streaming reads only (it reports 1/2 of the bytes there).  V1's chunk
// kernel reads one dword per lane, 64 consecutive floats of a 2 400-byte voxel row whose start is not line-aligned, lanes
// outside the row's z interval clamped onto the nearest voxel inside it.  Each kernel below reads a KNOWN set of bytes once;
// main() prints the number of distinct 32-, 64- and 128-byte blocks that set touches, tools/fetch_calib.py puts the counter
// beside them.
//   stream16   : 16 B per lane, whole array in order (the guide's case)
//   stream4    : 4 B per lane, whole array in order (the calibration profiles/r4_pmc_v1.txt used: mv_filter)
//   rowrel     : rows of ROW floats, interval [Z0, Z1) of every STRIDE-th row, 64-float chunks counted from the ROW start
//                (what mv_chunks_kernel did up to round 4), items dealt to waves like the chunk kernel (wave * 2, stride
//                n_waves * 2)
//   aligned    : the same intervals cut at absolute multiples of 64 floats of the array (256-byte aligned chunks)
//   rowrel_blk : rowrel with the items of one row kept in one BLOCK (no line is shared between blocks on different XCDs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

constexpr int ROW = 600, Z0 = 131, Z1 = 268, STRIDE = 3;      // 137 voxels per row: the mean interval of the bench's frame 25

__global__ __launch_bounds__(256) void stream16(const float4* __restrict__ a, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void stream4(const float* __restrict__ a, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 12345.678f) sink[0] = acc;
}

// MODE 0: row-relative chunks, items dealt wave-cyclically; 1: absolute 256-byte chunks, same dealing; 2: row-relative,
// one row per wave-pair trip kept inside a block (items of a row consecutive within one block's share)
template <int MODE>
__global__ __launch_bounds__(256) void rows(const float* __restrict__ a, int n_rows, int per_row, float* sink) {
    const int lane = threadIdx.x & 63;
    const unsigned n_items = (unsigned)n_rows * (unsigned)per_row;
    unsigned wave, n_waves;
    float acc = 0.f;
    if (MODE == 2) {
        // block b takes rows b, b + gridDim, ...; its 4 waves split the row's items
        for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
            const size_t row0 = (size_t)r * STRIDE * ROW;
            for (int c = (Z0 >> 6) + (threadIdx.x >> 6); c < (Z1 + 63) >> 6; c += 4) {
                const int lo = max(Z0 - (c << 6), 0), hi = min(Z1 - (c << 6), 64);
                const int l = min(max(lane, lo), hi - 1);
                acc += a[row0 + (c << 6) + l];
            }
        }
        if (acc == 12345.678f) sink[0] = acc;
        return;
    }
    wave = blockIdx.x * 4u + (threadIdx.x >> 6); n_waves = gridDim.x * 4u;
    for (unsigned it = wave * 2u; it < n_items; it += n_waves * 2u) {
#pragma unroll
        for (unsigned u = 0; u < 2; ++u) {
            const unsigned i = it + u;
            if (i >= n_items) break;
            const int r = (int)(i / (unsigned)per_row), k = (int)(i % (unsigned)per_row);
            const size_t row0 = (size_t)r * STRIDE * ROW;
            if (MODE == 0) {
                const int c = (Z0 >> 6) + k;
                if (c >= (Z1 + 63) >> 6) continue;
                const int lo = max(Z0 - (c << 6), 0), hi = min(Z1 - (c << 6), 64);
                const int l = min(max(lane, lo), hi - 1);
                acc += a[row0 + (c << 6) + l];
            } else {
                const size_t g0 = row0 + Z0, g1 = row0 + Z1;
                const size_t c = (g0 >> 6) + (size_t)k;
                if (c >= (g1 + 63) >> 6) continue;
                const size_t b = c << 6;
                const int lo = g0 > b ? (int)(g0 - b) : 0, hi = g1 - b < 64 ? (int)(g1 - b) : 64;
                const int l = min(max(lane, lo), hi - 1);
                acc += a[b + l];
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const size_t n = 800ull * 800ull * 600ull;            // the office0 volume: 1.536 GB per array
    const int n_rows = (int)(n / ROW / STRIDE);
    float *a, *sink;
    if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, n * 4);
    hipDeviceSynchronize();
    const int grid = 2048;
    // distinct blocks the row patterns touch (the same set for all three MODEs: the clamped lanes add no bytes)
    size_t b32 = 0, b64 = 0, b128 = 0;
    {
        size_t last32 = ~0ull, last64 = ~0ull, last128 = ~0ull;      // rows are visited in increasing address order
        for (int r = 0; r < n_rows; ++r) {
            const size_t lo = ((size_t)r * STRIDE * ROW + Z0) * 4, hi = ((size_t)r * STRIDE * ROW + Z1) * 4 - 1;
            for (size_t b = lo / 32; b <= hi / 32; ++b) if (b != last32) { ++b32; last32 = b; }
            for (size_t b = lo / 64; b <= hi / 64; ++b) if (b != last64) { ++b64; last64 = b; }
            for (size_t b = lo / 128; b <= hi / 128; ++b) if (b != last128) { ++b128; last128 = b; }
        }
    }
    printf("calib stream16 bytes %zu\n", n * 4);
    printf("calib stream4 bytes %zu\n", n * 4);
    printf("calib rows n_rows %d interval_bytes %zu blocks32 %zu (%zu B) blocks64 %zu (%zu B) blocks128 %zu (%zu B)\n", n_rows,
           (size_t)n_rows * (Z1 - Z0) * 4, b32, b32 * 32, b64, b64 * 64, b128, b128 * 128);
    const int per_row_rel = ((Z1 + 63) >> 6) - (Z0 >> 6), per_row_abs = (Z1 - Z0 + 63) / 64 + 1;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float4*>(a), n / 4, sink);
        hipLaunchKernelGGL(stream4, dim3(grid), dim3(256), 0, 0, a, n, sink);
        hipLaunchKernelGGL(rows<0>, dim3(grid), dim3(256), 0, 0, a, n_rows, per_row_rel, sink);
        hipLaunchKernelGGL(rows<1>, dim3(grid), dim3(256), 0, 0, a, n_rows, per_row_abs, sink);
        hipLaunchKernelGGL(rows<2>, dim3(grid), dim3(256), 0, 0, a, n_rows, per_row_rel, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("run failed\n"); return 1; }
    // wall times (events), for the record
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); printf("time %s %.1f us\n", name, ms * 1e3f);
    };
    timeit("stream16", [&] { hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float4*>(a), n / 4, sink); });
    timeit("stream4", [&] { hipLaunchKernelGGL(stream4, dim3(grid), dim3(256), 0, 0, a, n, sink); });
    timeit("rows<0>", [&] { hipLaunchKernelGGL(rows<0>, dim3(grid), dim3(256), 0, 0, a, n_rows, per_row_rel, sink); });
    timeit("rows<1>", [&] { hipLaunchKernelGGL(rows<1>, dim3(grid), dim3(256), 0, 0, a, n_rows, per_row_abs, sink); });
    timeit("rows<2>", [&] { hipLaunchKernelGGL(rows<2>, dim3(grid), dim3(256), 0, 0, a, n_rows, per_row_rel, sink); });
    return 0;
}
