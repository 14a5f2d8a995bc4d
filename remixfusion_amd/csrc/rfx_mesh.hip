// rfx_mesh.hip -- iso-surface extraction on the device: marching cubes over a dense
// [X,Y,Z] fp32 sample volume.  Replaces the host `skimage.measure.marching_cubes(raw, level, mask)` call
// of the reference's utils.py:158 (extract_mesh_github).  Two passes, both one thread per cell with z
// fastest so the 8 corner reads of a wave fall on 4 contiguous rows of the volume:
//   MC1 count : case index -> triangles of that cell (table n_tri[256])
//   MC2 emit  : the host scans the counts; each cell writes its triangles' corners at its offset, plus
//               a 64-bit key per corner (the cut grid edge) that the host uses to weld shared vertices.
// The tables come from the host (remixfusion_amd/mesh.py generates them); corner id = cx + 2cy + 4cz,
// edge e joins corners (ea[e], eb[e]) with ea < eb, i.e. ea is the lower corner along the edge's axis,
// so both cells sharing a grid edge interpolate it from the same two samples in the same order.
// HBM-bound integer/fp32 streaming work: 4 B/sample read (+1 B mask) and 4 B/cell written in MC1.
#include "rfx_common.h"

namespace rfx {

__constant__ int c_edge_a[12] = {0, 0, 0, 1, 1, 2, 2, 3, 4, 4, 5, 6};
__constant__ int c_edge_b[12] = {1, 2, 4, 3, 5, 3, 6, 7, 5, 6, 7, 7};

struct McGrid {
    int X, Y, Z;         // samples
    int cy, cz;          // cells along y, z
    long long n_cells;
    float level;
};

// corner samples + case of cell (x,y,z); returns -1 when a corner is masked out (or not finite)
__device__ __forceinline__ int cell_case(const float* __restrict__ vol, const uint8_t* __restrict__ mask, const McGrid& G,
                                         int x, int y, int z, float v[8]) {
    const size_t sY = (size_t)G.Z, sX = (size_t)G.Y * G.Z;
    const size_t base = (size_t)x * sX + (size_t)y * sY + z;
    int code = 0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const size_t i = base + (c & 1) * sX + ((c >> 1) & 1) * sY + ((c >> 2) & 1);
        v[c] = vol[i];
        if (mask) ok = ok && (mask[i] != 0);
        ok = ok && (v[c] == v[c]);                  // NaN samples disable the cell
        code |= (v[c] < G.level) ? (1 << c) : 0;
    }
    return ok ? code : -1;
}

__device__ __forceinline__ void cell_xyz(const McGrid& G, long long cell, int& x, int& y, int& z) {
    z = (int)(cell % G.cz);
    const long long r = cell / G.cz;
    y = (int)(r % G.cy);
    x = (int)(r / G.cy);
}

__global__ void __launch_bounds__(256) mc_count_kernel(const float* __restrict__ vol, const uint8_t* __restrict__ mask, McGrid G,
                                                       const int* __restrict__ n_tri, int* __restrict__ counts) {
    const long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= G.n_cells) return;
    int x, y, z;
    cell_xyz(G, cell, x, y, z);
    float v[8];
    const int code = cell_case(vol, mask, G, x, y, z, v);
    counts[cell] = (code <= 0 || code == 255) ? 0 : n_tri[code];
}

__global__ void __launch_bounds__(256) mc_emit_kernel(const float* __restrict__ vol, const uint8_t* __restrict__ mask, McGrid G,
                                                      const int* __restrict__ tri_edges, int row, const int* __restrict__ counts,
                                                      const long long* __restrict__ offsets, float* __restrict__ verts,
                                                      long long* __restrict__ keys) {
    const long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= G.n_cells) return;
    const int n = counts[cell];
    if (n == 0) return;
    int x, y, z;
    cell_xyz(G, cell, x, y, z);
    float v[8];
    const int code = cell_case(vol, mask, G, x, y, z, v);
    if (code <= 0 || code == 255) return;
    const long long out = offsets[cell] * 3;
    const int* __restrict__ te = tri_edges + (size_t)code * row;
    for (int k = 0; k < 3 * n; ++k) {
        const int e = te[k];
        const int a = c_edge_a[e], b = c_edge_b[e];
        const float va = v[a], vb = v[b];
        const float w = (G.level - va) / (vb - va);          // signs differ, so vb != va
        const int ax = x + (a & 1), ay = y + ((a >> 1) & 1), az = z + ((a >> 2) & 1);
        const int axis = (a ^ b) == 1 ? 0 : ((a ^ b) == 2 ? 1 : 2);
        float p[3] = {(float)ax, (float)ay, (float)az};
        p[axis] += w;
        float* o = verts + (size_t)(out + k) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        keys[out + k] = (((long long)ax * G.Y + ay) * G.Z + az) * 3 + axis;
    }
}

static bool make_grid(int X, int Y, int Z, float level, McGrid& G) {
    if (X < 2 || Y < 2 || Z < 2) return false;
    G.X = X; G.Y = Y; G.Z = Z; G.cy = Y - 1; G.cz = Z - 1;
    G.n_cells = (long long)(X - 1) * (Y - 1) * (Z - 1);
    G.level = level;
    return G.n_cells < (1ll << 31) * 256;
}

}  // namespace rfx

using namespace rfx;

extern "C" {

int rfx_mc_count(const float* volume, const uint8_t* mask, int X, int Y, int Z, float level, const int* n_tri, int* counts,
                 rfx_stream stream) {
    if (!volume || !n_tri || !counts) return RFX_ERR_ARG;
    McGrid G;
    if (!make_grid(X, Y, Z, level, G)) return RFX_ERR_ARG;
    hipLaunchKernelGGL(mc_count_kernel, dim3((unsigned)((G.n_cells + 255) / 256)), dim3(256), 0, as_stream(stream), volume, mask, G,
                       n_tri, counts);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

int rfx_mc_emit(const float* volume, const uint8_t* mask, int X, int Y, int Z, float level, const int* tri_edges, int max_tri,
                const int* counts, const long long* offsets, float* verts, long long* keys, rfx_stream stream) {
    if (!volume || !tri_edges || !counts || !offsets || !verts || !keys || max_tri <= 0) return RFX_ERR_ARG;
    McGrid G;
    if (!make_grid(X, Y, Z, level, G)) return RFX_ERR_ARG;
    hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)((G.n_cells + 255) / 256)), dim3(256), 0, as_stream(stream), volume, mask, G,
                       tri_edges, max_tri * 3, counts, offsets, verts, keys);
    RFX_LAUNCH_CHECK();
    return RFX_OK;
}

}  // extern "C"
