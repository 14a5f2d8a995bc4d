// dev probe (gfx950): is the scalar offset of a raw buffer access part of the range check?
// descriptor: base = p, num_records = 256 bytes; lanes read voffset = lane*4 (+ 0 or 1024 for lanes >= 32) with soffset = 4096.
// prints the values lanes 0, 31, 32, 63 get (data = byte offset / 4) and what a store with the same addressing wrote.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float* p, float* out, unsigned soff) {
    const unsigned lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 256, 0x00020000);
    const unsigned voff = lane * 4u + (lane >= 32 ? 1024u : 0u);
    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    out[lane] = __uint_as_float(v);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-1.0f - (float)lane), r, voff, soff + 8192u, 0);
}
int main() {
    const int n = 1 << 16;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 4096u);
    float r[64];
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    printf("load  lane0 %.0f lane31 %.0f lane32 %.0f lane63 %.0f (soffset 4096 B = element 1024: in-range lanes read 1024+lane)\n", r[0], r[31], r[32], r[63]);
    printf("store elem[3072] %.0f elem[3103] %.0f elem[3360] %.0f (soffset 12288 B = element 3072; -1-lane where written)\n", h[3072], h[3103], h[3360]);
    return 0;
}
