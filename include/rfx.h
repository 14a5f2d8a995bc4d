/*
 * include/rfx.h -- C ABI of librfx.so: the MI355X (gfx950) replacement for RemixFusion's
 * mapping hot-path kernels.
 *
 * The reference has no C ABI on this path: its kernels are CUDA-C strings JIT-compiled by
 * PyCUDA and tiny-cuda-nn torch modules.  Each entry point below names the reference
 * interface it replaces (file:line under the reference tree).  INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - extern "C", returns int: 0 = RFX_OK, <0 = error (never throws, never aborts).
 *   - Every `float*` / `const float*` documented as "dev" is a device pointer owned by the
 *     caller (e.g. torch tensor.data_ptr()); "host" pointers are small parameter arrays read
 *     before the call returns.  No hidden allocation: scratch comes from an explicit
 *     workspace whose size the matching *_workspace_bytes() query reports.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous w.r.t. the host and safe to capture into a hipGraph.
 *   - Scalars are passed in their true types (the reference packs them into fp32 arrays).
 *   - No global state; distinct streams / volumes may be driven from distinct threads.
 */
#ifndef RFX_H
#define RFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFX_OK               0
#define RFX_ERR_ARG         -1   /* null pointer / non-positive size / inconsistent shapes */
#define RFX_ERR_HIP         -2   /* a HIP runtime call failed (see rfx_last_hip_error)      */
#define RFX_ERR_UNSUPPORTED -3   /* configuration outside what the kernels implement        */
#define RFX_ERR_WORKSPACE   -4   /* workspace pointer null or too small                     */

#define RFX_ABI_VERSION 10

typedef void* rfx_stream;

int rfx_abi_version(void);
/* hipError_t of the most recent failing HIP call on this thread (0 if none). */
int rfx_last_hip_error(void);

/* Timing events for a caller without a HIP binding of its own (ABI 10; bench.py's live roofline): hipEventCreateWithFlags
 * (hipEventDisableSystemFence: for timing only -- waiting on such an event does NOT make device memory visible to the host) /
 * hipEventDestroy / hipEventSynchronize + hipEventElapsedTime (ms between two RECORDED events; RFX_ERR_HIP when either was
 * never recorded, and the runtime's sticky error is cleared).  The handles are plain hipEvent_t: a caller that has HIP may
 * pass its own. */
typedef void* rfx_event;
int rfx_event_create(rfx_event* out);
int rfx_event_destroy(rfx_event ev);
int rfx_event_elapsed_ms(rfx_event start, rfx_event stop, float* ms);

/* ======================================================================================
 * Moving TSDF volume (MV).  Layout: three fp32 arrays of dx*dy*dz voxels, z fastest
 * (idx = z + y*dz + x*dy*dz), tsdf init 1, weight init 0, colour packed B*65536+G*256+R.
 * ==================================================================================== */

/* V1: replaces kernel `integrate` model/Volume.py:196-336 + host wrapper :713-757.
 *   K[9] row-major intrinsics (host), c2w[16] row-major camera-to-world (host),
 *   color_packed/depth: dev [H*W]; old_bnd[6] = x0,x1,y0,y1,z0,z1 (host, used iff reintegrate).
 *   index_decode: 0 = reproduce the reference's fp32 index decode (incl. its rounding
 *   artefacts next to slab boundaries when dx*dy*dz > 2^24), 1 = exact integer decode.
 *   workspace: dev, 16-byte aligned, >= rfx_tsdf_integrate_workspace_bytes(dx, dy, dz, H, W): the packed
 *   {depth, 1/lambda} image, the fast path's classification image, a packed-colour image, coarse max-depth tiles and the
 *   queue of 64-voxel chunks the frustum touches (32 B per chunk of the volume: 205 MB at 800x800x600; contents are
 *   scratch, nothing is carried between calls).  ABI 2: the query takes the volume dimensions.  ABI 4: 32-byte items. */
size_t rfx_tsdf_integrate_workspace_bytes(int dx, int dy, int dz, int H, int W);
int rfx_tsdf_integrate(float* tsdf, float* weight, float* color, int dx, int dy, int dz,
                       const float origin[3], float voxel, const float K[9], const float c2w[16],
                       const float* color_packed, const float* depth, int H, int W,
                       float trunc, float obs_weight, int weight_clamp, int reintegrate,
                       const float old_bnd[6], int index_decode,
                       void* workspace, size_t workspace_bytes, rfx_stream stream);

/* V1 on one x-slab of the volume (multi-GPU: the volume is cut into contiguous x-slabs, x being the slowest axis of
 * the reference layout model/Volume.py:224-226, one slab per GPU, every GPU integrating the same frame).  dx, dy, dz,
 * origin describe the WHOLE volume; tsdf/weight/color hold only the planes [x0, x1) ((x1-x0)*dy*dz voxels each).  Every
 * voxel is evaluated with its global index and coordinates, so the slabs put together are bit-identical to
 * rfx_tsdf_integrate on the whole volume.  workspace >= rfx_tsdf_integrate_workspace_bytes(x1 - x0, dy, dz, H, W).
 * The reference has no counterpart (single GPU); semantics are those of V1. */
int rfx_tsdf_integrate_slab(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1,
                            const float origin[3], float voxel, const float K[9], const float c2w[16],
                            const float* color_packed, const float* depth, int H, int W,
                            float trunc, float obs_weight, int weight_clamp, int reintegrate,
                            const float old_bnd[6], int index_decode,
                            void* workspace, size_t workspace_bytes, rfx_stream stream);

/* V1 with the colour packing of model/Volume.py:725-728 folded into the call: rgb255 dev [H*W,3] (0..255 valued floats,
 * what the reference's host code packs with numpy before the upload) instead of color_packed; x0 = 0, x1 = dx for the
 * whole volume.  Saves the rfx_pack_color launch in the frame loop; results are those of rfx_pack_color + V1. */
int rfx_tsdf_integrate_rgb(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1,
                           const float origin[3], float voxel, const float K[9], const float c2w[16],
                           const float* rgb255, const float* depth, int H, int W,
                           float trunc, float obs_weight, int weight_clamp, int reintegrate,
                           const float old_bnd[6], int index_decode,
                           void* workspace, size_t workspace_bytes, rfx_stream stream);

/* host-side colour packing of model/Volume.py:725-728 moved to the device:
 * rgb255 dev [n,3] (0..255 valued floats) -> packed dev [n] = floor(B*65536+G*256+R). */
int rfx_pack_color(const float* rgb255, float* packed, int64_t n, rfx_stream stream);

/* V6: replaces `clean_tsdf` model/Volume.py:561-583 (host :656-677): tsdf=1, weight=0, color=0. */
int rfx_tsdf_fill(float* tsdf, float* weight, float* color, int64_t n, rfx_stream stream);

/* V7: replaces `copy_volume` model/Volume.py:585-610 (host :883-908). */
int rfx_tsdf_copy(const float* tsdf, const float* weight, const float* color,
                  float* tsdf_back, float* weight_back, float* color_back, int64_t n, rfx_stream stream);

/* V2: replaces `swap_rot_trans` model/Volume.py:128-194 (host :796-855): gather the new
 * volume (dims, origin) from the back copy (old_dims, old_origin); (1,0,0) outside. */
int rfx_tsdf_shift(float* tsdf, float* weight, float* color, int dx, int dy, int dz, const float origin[3],
                   const float* old_tsdf, const float* old_weight, const float* old_color,
                   int odx, int ody, int odz, const float old_origin[3], float voxel,
                   int index_decode, rfx_stream stream);

/* V2 on one x-slab (multi-GPU): the planes [x0, x1) of the new volume are gathered from the planes [ox_a, ox_b) of the
 * old one; tsdf/weight/color hold only [x0, x1), old_* only [ox_a, ox_b) (the caller fetched them from the ranks that
 * own them: rfx_tsdf_shift_source_planes says which are read).  dims/origins describe the WHOLE volumes; results are
 * bit-identical to rfx_tsdf_shift on the whole volume.  old_* may be NULL when ox_a == ox_b (slab outside the old box). */
int rfx_tsdf_shift_slab(float* tsdf, float* weight, float* color, int dx, int dy, int dz, int x0, int x1, const float origin[3],
                        const float* old_tsdf, const float* old_weight, const float* old_color,
                        int odx, int ody, int odz, int ox_a, int ox_b, const float old_origin[3], float voxel,
                        int index_decode, rfx_stream stream);
/* host: the old x-planes [*a, *b) that new planes [x0, x1) read in rfx_tsdf_shift_slab (one plane of slack either side). */
int rfx_tsdf_shift_source_planes(int x0, int x1, const float origin[3], int odx, const float old_origin[3], float voxel,
                                 int* a, int* b);

/* V3: replaces `tri_intepolate` model/Volume.py:337-458 (host :760-794).
 * pts dev [n,3] world coords; out5 dev [n,5] = (tsdf, r, g, b, tsdf at low corner). */
int rfx_tsdf_trilerp(const float* tsdf, const float* weight, const float* color, int dx, int dy, int dz,
                     const float origin[3], float voxel, const float* pts, int64_t n, float* out5,
                     rfx_stream stream);

/* V4: replaces `filter_tsdf` model/Volume.py:462-487 (host :857-881). */
/* V3 on one x-slab [x0, x1) of the volume (one GPU of several; dx, dy, dz, origin describe the WHOLE volume, tsdf/color hold
 * the slab's planes).  A point belongs to the slab that owns the plane of its lower corner (points outside the volume:
 * to the slab with x0 = 0, which writes the reference's default record); that slab writes out5[p] and inside[p] = 1, the
 * others leave out5[p] alone and write inside[p] = 0.  tsdf_halo / color_halo: plane x1 ([dy*dz] each, from the right
 * neighbour -- the 1-plane halo of SURVEY 8e), NULL for the last slab.  Same arithmetic as rfx_tsdf_trilerp on the whole
 * volume: the owning slab's record is bit-identical to it. */
int rfx_tsdf_trilerp_slab(const float* tsdf, const float* color, int dx, int dy, int dz, int x0, int x1,
                          const float* tsdf_halo, const float* color_halo, const float origin[3], float voxel,
                          const float* pts, int64_t n, float* out5, uint8_t* inside, rfx_stream stream);

int rfx_tsdf_filter(float* tsdf, float* weight, float* color, int64_t n, float weight_threshold,
                    rfx_stream stream);

/* V5: replaces `get_truncated_pc` model/Volume.py:489-559 (host :622-653).
 * pc7 dev [pc_num,7] (caller zero-fills), count dev uint32[1] (caller zero-fills). Slot = idx % pc_num;
 * when several voxels share a slot the highest voxel index wins (deterministic, unlike the
 * reference's race). */
int rfx_tsdf_truncated_pc(const float* tsdf, const float* color, int dx, int dy, int dz,
                          const float origin[3], float voxel, float trunc, int pc_num, float trunc_tsdf,
                          float* pc7, uint32_t* count, int index_decode, rfx_stream stream);

/* V5 on one x-slab [x0, x1) of the volume (tsdf / color hold the slab's planes; dims and origin describe the whole volume).
 * Slots are filled from the slab's own voxels only (global indices, so slot = global index % pc_num as on one GPU);
 * hit dev uint8[pc_num] (may be NULL) = 1 where this slab wrote.  Merging over the slabs -- the slab with the highest
 * x-planes that hit a slot wins it -- gives the record rfx_tsdf_truncated_pc leaves (remixfusion_amd/dist.py). */
int rfx_tsdf_truncated_pc_slab(const float* tsdf, const float* color, int dx, int dy, int dz, int x0, int x1,
                               const float origin[3], float voxel, float trunc, int pc_num, float trunc_tsdf,
                               float* pc7, uint32_t* count, uint8_t* hit, int index_decode, rfx_stream stream);

/* ======================================================================================
 * Global explicit volume (GBV/GBW): trgb dev [R^3,4] interleaved (tsdf,r,g,b), x fastest;
 * w dev [R^3].  Same memory the dense-grid lookup (rfx_field_*) reads.
 * ==================================================================================== */

/* G1: replaces `integrate` mp_slam/mapper.py:37-158 (host :823-872).  c2w is a DEVICE pointer
 * (the reference passes Holder(pose)); box[6] = x0,x1,y0,y1,z0,z1 (host); rgb01 dev [H,W,3]. */
int rfx_gbv_integrate(float* trgb, float* w, int res, const float box[6], const float K[9],
                      const float* c2w_dev, const float* rgb01, const float* depth, int H, int W,
                      float trunc, float obs_weight, rfx_stream stream);

/* G2: replaces `clean_tsdf` mp_slam/mapper.py:161-183 (host :267-282): t=1, rgb=0. */
int rfx_gbv_clear(float* trgb, int64_t n_voxels, rfx_stream stream);

/* ======================================================================================
 * Residual neural field.  Replaces tinycudann.Encoding (HashGrid: model/encodings.py:33-51;
 * OneBlob: :65-76; dense Grid: model/scene_rep.py:60-93) and the torch MLP
 * (model/decoder.py:116-146) for the queries of model/scene_rep.py:212-349.
 * ==================================================================================== */

#define RFX_MAX_LEVELS 16

typedef struct rfx_grid_desc {          /* one multi-resolution grid (tiny-cuda-nn layout) */
    int32_t  n_levels;                  /* <= RFX_MAX_LEVELS                                 */
    int32_t  n_feat;                    /* features per entry: 2 (hash), 4 (GBV), 1 (GBW)    */
    float    scale[RFX_MAX_LEVELS];     /* exp2(l*log2(pls))*base - 1                        */
    uint32_t res[RFX_MAX_LEVELS];       /* ceil(scale)+1                                     */
    uint32_t size[RFX_MAX_LEVELS];      /* entries in level                                  */
    uint32_t offset[RFX_MAX_LEVELS];    /* first entry of level                              */
    uint32_t hashed[RFX_MAX_LEVELS];    /* 1 = spatial hash, 0 = dense index                 */
} rfx_grid_desc;

typedef struct rfx_field_desc {
    rfx_grid_desc hash;                 /* embed_res_fn: 16 levels x 2 features              */
    const float*  hash_table;           /* dev, hash.offset[L-1]+size[L-1] entries * 2       */
    const float*  gbv;                  /* dev [R^3*4]                                       */
    int32_t       gbv_res;              /* R (globalV.base_resolution)                       */
    const float*  w1;                   /* dev [32,81] torch Linear.weight layout [out,in]   */
    const float*  w2;                   /* dev [16,32]                                       */
    const float*  w3;                   /* dev [32,66]                                       */
    const float*  w4;                   /* dev [3,32]                                        */
    float         tsdf_scale;           /* training.c_trunc / training.trunc is applied as
                                           (x*c_trunc)/trunc; both factors are passed:      */
    float         c_trunc;
    float         trunc;
    float         clamp_hi;             /* mapping.clamp when clamp mode is on, else 1       */
    int32_t       clamp_mode;           /* JointEncoding.clamp (scene_rep.py:332-337)        */
    int32_t       pos_fp16;             /* 0 = fp32 OneBlob (the reference: model/encodings.py:73
                                           passes dtype=torch.float); 1 = opt-in: outputs rounded
                                           to fp16 and fed to the fp16 matrix pipe              */
    const float*  staged;               /* optional dev [rfx_field_staged_floats()]: w1..w4 in
                                           MFMA operand order, written by rfx_field_stage_weights
                                           for the CURRENT weights; NULL = every block re-derives
                                           the layout from w1..w4 itself (slower prologue)      */
} rfx_field_desc;

/* D1 weight staging: one small launch per weight update instead of a gather per block.  `staged` must be
 * 16-byte aligned; f->staged is ignored here. */
size_t rfx_field_staged_floats(void);
int rfx_field_stage_weights(const rfx_field_desc* f, float* staged, rfx_stream stream);

/* E1 alone (query_sdf_res(embed=True), mp_slam/slam.py:209): x01 dev [n,3] -> feat dev [n, L*F]. */
int rfx_grid_encode_forward(const rfx_grid_desc* g, const float* table, const float* x01, int64_t n,
                            float* feat, rfx_stream stream);
/* backward of the above: dfeat dev [n, L*F] -> accumulated (float atomics) into dtable (dev, same
 * shape as table, caller zero-fills) and, if dx01 != NULL, dx01 dev [n,3] (overwritten).
 * workspace (optional, dev, >= rfx_grid_encode_backward_workspace_bytes): enables the LDS-privatised
 * scatter (per-segment accumulation in LDS, contiguous flush); NULL = direct atomics. */
size_t rfx_grid_encode_backward_workspace_bytes(int64_t n, int n_levels);
/* The size that lets the scatter keep ALL binned levels of grid `g` (levels of >= 12 segments of 8 192 entries; dense levels from 8) in flight at
 * once -- four launches per sweep instead of four per level; with the minimum above they go one level at a time.  Any size
 * in between is used for as many levels per group as fit.  Equals the minimum for grids without binned levels. (ABI 5) */
size_t rfx_grid_encode_backward_workspace_bytes_for(const rfx_grid_desc* g, int64_t n);
int rfx_grid_encode_backward(const rfx_grid_desc* g, const float* table, const float* x01, int64_t n,
                             const float* dfeat, float* dtable, float* dx01, void* workspace, size_t workspace_bytes,
                             rfx_stream stream);

/* E2 alone: tcnn OneBlob, x01 dev [n,3] -> dev [n, 3*n_bins]. */
int rfx_oneblob_forward(const float* x01, int64_t n, int n_bins, int pos_fp16, float* out, rfx_stream stream);

/* Q1 fused: x01 -> raw4 = (rgb + ex_rgb, sdf + tsdf) (model/scene_rep.py:314-349): E1+E2+E3,
 * tsdf rescale/clamp, MFMA MLP, residual add.  raw4 dev [n,4]. */
int rfx_field_forward(const rfx_field_desc* f, const float* x01, int64_t n, float* raw4, rfx_stream stream);

/* Backward of Q1.  draw4 dev [n,4].  Accumulates into d_hash (dev, hash table shape, float
 * atomics) and dw1..dw4 (dev, weight shapes, deterministic two-stage sum) -- like torch .grad the
 * caller zero-fills or keeps accumulating; any of them may be NULL.  dx01 (dev [n,3], may be
 * NULL) is overwritten with dL/dx01 (through hash grid, OneBlob and the GBV lookup).
 * workspace: dev, 16-byte aligned, >= rfx_field_backward_workspace_bytes(n). */
size_t rfx_field_backward_workspace_bytes(int64_t n);
int rfx_field_backward(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                       float* d_hash, float* dw1, float* dw2, float* dw3, float* dw4, float* dx01,
                       void* workspace, size_t workspace_bytes, rfx_stream stream);

/* The four stages rfx_field_backward runs, exposed so a caller can time / overlap them:
 *   _chain   : recompute forward + dX chain on the matrix cores; stages per-point rows in the workspace
 *   _weights : dW = dY^T X over points (matrix cores, hidden activations recomputed from the staged rows, deterministic
 *              two-stage sum), accumulates into dw*; reads draw4 as the chain staged it (the argument is not read)
 *   _scatter : hash-grid gradient scatter (LDS-privatised / binned, float atomics) into d_hash; writes the hash
 *              part of dx01 when dx01 != NULL
 *   _dx      : adds the OneBlob / GBV part of dL/dx01
 * All take the same workspace, x01, n and draw4; _chain must run first.  From 16 384 points on, _chain orders the points
 * with a non-zero draw4 row first and the later stages work on those only (a point with draw4 == 0 contributes exactly
 * nothing to any gradient; its dx01 row is written as zeros): same results up to the order of the additions. */
int rfx_field_backward_chain(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                             void* workspace, size_t workspace_bytes, rfx_stream stream);
/* _chain for callers that want input gradients only (pose phase: map frozen): stages just the dX1 rows, which is all
 * _scatter (with dx01) and _dx read; _weights and the d_hash part of _scatter must not follow it. */
int rfx_field_backward_chain_inputs(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                    void* workspace, size_t workspace_bytes, rfx_stream stream);
/* _chain for callers that want parameter gradients only (map phase: poses fixed): everything _weights reads, and of
 * dX1 just d_emb, which is all the d_hash part of _scatter(_merged) reads; _dx and the dx01 output of _scatter must
 * not follow it. */
int rfx_field_backward_chain_weights(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream);
/* The forward / backward pair of ONE optimisation step shares its hash lookups: rfx_field_forward_stash is
 * rfx_field_forward that also leaves the interpolated hash features of the n points in `workspace` (a buffer of
 * rfx_field_backward_workspace_bytes(n)), and rfx_field_backward_chain_stashed is the chain stage that reads them instead
 * of looking the table up again (the reference's autograd keeps the encoding output alive between forward and backward the
 * same way: model/scene_rep.py:141-160 run_network -> tcnn backward).  Valid only while x01, n, the hash table and the
 * workspace are what the forward saw (no optimiser step, no other _chain on that workspace in between); results are
 * bit-identical to the un-stashed entry points. */
int rfx_field_forward_stash(const rfx_field_desc* f, const float* x01, int64_t n, float* raw4, void* workspace,
                            size_t workspace_bytes, rfx_stream stream);
int rfx_field_backward_chain_stashed(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                     void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_field_backward_chain_inputs_stashed(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                            void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_field_backward_chain_weights_stashed(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4,
                                             void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_field_backward_weights(int64_t n, const float* draw4, float* dw1, float* dw2, float* dw3, float* dw4,
                               void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_field_backward_scatter(const rfx_field_desc* f, const float* x01, int64_t n, float* d_hash, float* dx01,
                               void* workspace, size_t workspace_bytes, rfx_stream stream);
/* hash-gradient scatter of the chain's d_emb rows (as rfx_field_backward_scatter with dx01 = NULL) MERGED with a
 * second point set -- extra_x01 dev [m,3], extra_dfeat dev [m, 2L] row-major, e.g. the TV lattice and its feature
 * gradient -- so that one sweep over the table segments serves both.  scatter_ws: dev, >=
 * rfx_grid_encode_backward_workspace_bytes(n + extra_n, L), or NULL for direct atomics.  d_hash is accumulated into. */
int rfx_field_backward_scatter_merged(const rfx_field_desc* f, const float* x01, int64_t n, const float* extra_x01,
                                      const float* extra_dfeat, int64_t extra_n, float* d_hash, void* workspace,
                                      size_t workspace_bytes, void* scatter_ws, size_t scatter_bytes, rfx_stream stream);
int rfx_field_backward_dx(const rfx_field_desc* f, const float* x01, int64_t n, const float* draw4, float* dx01,
                          void* workspace, size_t workspace_bytes, rfx_stream stream);

/* ---- a hash table partitioned by LEVEL over several GPUs (ABI 6) -------------------------------------------------
 * One scene on N GPUs (the reference is single-GPU, mp_slam/mapper.py:392-423 is the iteration whose result
 * is reproduced): rank q keeps -- looks up, accumulates gradients for, steps -- the hash levels [l_q, l_{q+1}) only, for
 * EVERY sample point of the iteration, and each rank runs the decoder on its share of the rays.  What travels between the
 * ranks is per-point rows (8 B per point and level, twice per iteration: features out, feature gradients back) instead of
 * the dense table gradient (6.6 ... 166 MB).  The four entry points below are the pieces that differ from the
 * single-GPU path; everything else is the public stages above, given a sub-grid descriptor (rfx_grid_desc with
 * n_levels = the own levels and their true offsets into the full table). */
typedef struct rfx_level_rows {          /* where the two features of hash level l of point p live:            */
    const float* rows[RFX_MAX_LEVELS];  /*   rows[l][p * ld[l] + col[l] + {0, 1}]   (dev, 8-byte aligned;       */
    int32_t      ld[RFX_MAX_LEVELS];    /*   ld, col even) -- e.g. one row-major block [points, 2 k_q] per       */
    int32_t      col[RFX_MAX_LEVELS];   /*   owning rank q, col = 2 (l - l_q)                                    */
} rfx_level_rows;
/* rows -> the forward's stash inside `workspace` (rfx_field_backward_workspace_bytes(n)): what rfx_field_forward_stash
 * would have left there had it looked all 16 levels up itself. */
int rfx_field_stash_put(const rfx_level_rows* rows, int64_t n, void* workspace, size_t workspace_bytes, rfx_stream stream);
/* rfx_field_forward for points whose hash features are in the stash (rfx_field_stash_put, or an earlier
 * rfx_field_forward_stash on the same points): the hash table is not read.  Bit-identical to rfx_field_forward on the
 * features' table.  The _stashed chain stages follow it as they follow rfx_field_forward_stash. */
int rfx_field_forward_stashed(const rfx_field_desc* f, const float* x01, int64_t n, float* raw4, void* workspace,
                              size_t workspace_bytes, rfx_stream stream);
/* After a chain stage on n points: its d_emb rows (gradient w.r.t. the 32 hash features) written back as level rows
 * (the pointers are written through), in the caller's point order, zeros for the points the selection dropped.
 * Optionally (loss_partials != NULL) the same launch adds n_loss_partials x 8 doubles of per-block loss sums up to
 * loss_total8 dev double[8], the quantity rfx_mapping_loss_finalize takes after the ranks have summed it. */
int rfx_field_backward_demb_rows(int64_t n, const rfx_level_rows* rows, const double* loss_partials, int n_loss_partials,
                                 double* loss_total8, void* workspace, size_t workspace_bytes, rfx_stream stream);
/* rfx_grid_encode_backward's table gradient for TWO point sets in one sweep (e.g. the sample points with the gradient
 * rows received from the rendering ranks, and the TV lattice with its own): dtable += scatter(a) + scatter(b).
 * workspace (optional): >= rfx_grid_encode_backward_workspace_bytes(n_a + n_b, g->n_levels). */
int rfx_grid_encode_backward_merged(const rfx_grid_desc* g, const float* table, const float* x01_a, int64_t n_a,
                                    const float* dfeat_a, const float* x01_b, int64_t n_b, const float* dfeat_b, float* dtable,
                                    void* workspace, size_t workspace_bytes, rfx_stream stream);

/* Q2 point queries (model/scene_rep.py:212-310).  out dev [n] or [n,3] as documented. */
int rfx_field_query_sdf(const rfx_field_desc* f, const float* x01, int64_t n, float* sdf, rfx_stream stream);          /* query_sdf_res      */
int rfx_field_query_color(const rfx_field_desc* f, const float* x01, int64_t n, float* rgb3, rfx_stream stream);       /* query_color_residual */

/* ======================================================================================
 * Ray sampling + volume rendering (model/scene_rep.py:107-127,156-179,407-456).
 * ==================================================================================== */
typedef struct rfx_sampler_desc {
    float   near, far, range_d;         /* cam.near, cam.far, training.range_d               */
    int32_t n_range_d, n_samples_d;     /* training.n_range_d, training.n_samples_d          */
    float   perturb;                    /* training.perturb (>0 = stratified jitter)         */
} rfx_sampler_desc;

/* S1: target_d dev [n]; u01 dev [n,S] uniform draws (NULL or perturb<=0 -> no jitter);
 * z_vals dev [n,S], S = n_range_d + n_samples_d, sorted ascending. */
int rfx_sample_z(const rfx_sampler_desc* s, const float* target_d, const float* u01, int64_t n_rays,
                 float* z_vals, rfx_stream stream);

/* x01 = ((o + d*z) - bb_min) / (bb_max - bb_min) (model/scene_rep.py:443,:388): rays_o/rays_d dev
 * [n,3], bbox[6] = x0,x1,y0,y1,z0,z1 (host, double).  bbox_f64 != 0 evaluates the normalisation
 * in float64 and rounds to fp32 -- what torch's type promotion does in the reference whenever
 * mapping.bound contains a non-integer (the bound tensor is then float64); 0 = all fp32.
 * x01 dev [n,S,3]. */
int rfx_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays, int S,
                   const double bbox[6], int bbox_f64, float* x01, rfx_stream stream);

/* R1 forward: raw4 dev [n,S,4], z dev [n,S] -> rgb dev [n,3], depth dev [n]; weights dev [n,S]
 * (normalised, saved for backward, may be NULL). */
int rfx_composite_forward(const float* raw4, const float* z_vals, int64_t n_rays, int S, float trunc,
                          float sc_factor, float* rgb, float* depth, float* weights, rfx_stream stream);
/* R1 backward: d_rgb dev [n,3], d_depth dev [n] -> d_raw4 dev [n,S,4] (overwritten). */
int rfx_composite_backward(const float* raw4, const float* z_vals, int64_t n_rays, int S, float trunc,
                           float sc_factor, const float* d_rgb, const float* d_depth, float* d_raw4,
                           rfx_stream stream);

/* L1 fused: the four mapping losses of JointEncoding.mapping (model/scene_rep.py:493-527 with
 * model/utils.py:170-256) from raw/z and the rendered maps.  sums8 dev double[RFX_LOSS_WS_DOUBLES] (scratch,
 * need not be initialised: per-block partial sums, added in a fixed order),
 * losses4 dev float[4] = (rgb, depth, sdf, fs), coef4 dev float[4] (kept for the backward).
 * trunc_loss = training.trunc * data.sc_factor; rgb_missing_on = (training.rgb_missing > 0). */
#define RFX_LOSS_WS_DOUBLES 2048
int rfx_mapping_loss_forward(const float* raw4, const float* z_vals, const float* rgb_map, const float* depth_map,
                             const float* target_rgb, const float* target_d, int64_t n_rays, int S, float trunc_loss,
                             float depth_trunc, int rgb_missing_on, double* sums8, float* losses4, float* coef4,
                             rfx_stream stream);
/* The same forward in two halves, for a ray batch spread over several GPUs (multi-GPU mapping: every rank renders a
 * share of the batch).  rfx_mapping_loss_sums reduces THIS rank's rays to total8 dev double[8] (scratch: dev
 * double[RFX_LOSS_WS_DOUBLES]); the caller adds total8 over the ranks (all-reduce of 64 bytes); rfx_mapping_loss_finalize
 * forms the four losses and the backward's coefficients of the WHOLE batch (n_rays_total rays) from the summed total8.
 * The loss weights fs_w / sdf_w depend on sample counts of the whole batch (model/utils.py:170-198), so the loss is not a
 * sum of per-rank losses: this split keeps it identical to the single-GPU value. */
int rfx_mapping_loss_sums(const float* raw4, const float* z_vals, const float* rgb_map, const float* depth_map,
                          const float* target_rgb, const float* target_d, int64_t n_rays, int S, float trunc_loss,
                          float depth_trunc, int rgb_missing_on, double* scratch, double* total8, rfx_stream stream);
int rfx_mapping_loss_finalize(const double* total8, int64_t n_rays_total, int S, float* losses4, float* coef4, rfx_stream stream);
/* d_raw4 = d(sum_i gout4[i] * loss_i)/d raw4 including the path through the compositing (R1
 * backward); g_rgb_map / g_depth_map: optional extra grads on the rendered maps (may be NULL). */
int rfx_mapping_loss_backward(const float* raw4, const float* z_vals, const float* rgb_map, const float* depth_map,
                              const float* target_rgb, const float* target_d, int64_t n_rays, int S, float trunc,
                              float sc_factor, float trunc_loss, float depth_trunc, int rgb_missing_on, const float* coef4,
                              const float* gout4, const float* g_rgb_map, const float* g_depth_map, float* d_raw4,
                              rfx_stream stream);

/* TV1: total variation of lattice features feat dev [P,P,P,C] (mp_slam/slam.py:211-215):
 * *sum1 (dev double) = sum of squared forward differences along x, y, z.  Backward writes
 * dfeat = gscale_dev[0] * scale * d(sum)/d feat (gscale_dev NULL: 1). */
int rfx_tv_forward(const float* feat, int P, int C, double* sum1, rfx_stream stream);
int rfx_tv_backward(const float* feat, int P, int C, float scale, const float* gscale_dev, float* dfeat, rfx_stream stream);
/* the lattice itself (mp_slam/slam.py:198-207): pts dev [P^3,3] = ((ijk + u6[3:6]) * voxel + lo + u6[0:3] * offset_max
 * + margin), offset_max = hi - lo - P*voxel - 2*margin, divided by (hi - lo) after subtracting lo when
 * `normalise`; u6 dev [6] uniform(0,1).  bbox = (x0,x1,y0,y1,z0,z1); bbox_f64 = 1 reproduces torch's float64 promotion
 * for non-integer bounds, 0 the integer-bound case (jitter truncated to 0, fp32 arithmetic). */
int rfx_tv_lattice(const float* u6, int P, float voxel, float margin, const double bbox[6], int bbox_f64, int normalise,
                   float* pts, rfx_stream stream);

/* k distinct pseudo-random indices out of range(population), on the device: replaces python's
 * random.sample in the ray samplers (model/keyframe.py:33,89; mp_slam/mapper.py:396).  out dev int64[k].
 * Deterministic in (seed, population). */
/* out[e] = the e-th uniform of stream `stream_id` under `seed`: word 0 of Philox4x32-10(counter (e, stream_id), key seed)
 * as a float in [0, 1) with 24 random bits (torch.rand's resolution).  Replaces the torch.rand((n, S)) / torch.rand(6) draws
 * of scene_rep.py:437 and mp_slam/slam.py:198-203 where a caller lets rfx_ba_forward_backward draw for itself
 * (rfx_ba_desc.seed_u): this entry point reproduces those draws in a buffer. */
int rfx_uniform_draws(uint64_t seed, int stream_id, int64_t n, float* out, rfx_stream stream);

int rfx_random_subset(uint64_t seed, int64_t population, int64_t k, int64_t* out, rfx_stream stream);
/* The same with the population on the DEVICE: *population_dev (dev int64, <= fallback_population) if it exceeds k, else
 * fallback_population, in which case *used_fallback_dev (dev int32, may be NULL) is set to 1 (else 0).  Replaces the
 * `num_valid > num_rays_to_save ? sample(range(num_valid)) : sample(range(H*W))` of KeyFrameDatabase.sample_single_keyframe_rays
 * (model/keyframe.py:37-47) without bringing num_valid to the host -- a synchronisation that would drain the mapper's queue
 * once per keyframe.  Same permutation as rfx_random_subset(seed, N, k) for the N it ends up with. */
int rfx_random_subset_dev(uint64_t seed, const int64_t* population_dev, int64_t fallback_population, int64_t k, int64_t* out,
                          int32_t* used_fallback_dev, rfx_stream stream);

/* M1 ray batch of one optimisation iteration: replaces the host glue of mp_slam/mapper.py:394-409
 * (KeyFrameDatabase.sample_global_rays + random.sample over the current frame + torch.cat + poses_all[ids]
 * + rays_o/rays_d).  Rays are 7 floats (camera dir 3 | rgb 3 | depth 1).  The first n_kf_samples rays are
 * rfx_random_subset(seed_kf, rays_per_kf*num_kf) of kf_rays dev [num_kf*rays_per_kf, 7], posed by
 * poses16[kf_frame_ids[slot] / keyframe_every]; the last n_cur are rfx_random_subset(seed_cur,
 * cur_population) of cur_rays, posed by poses16[K-1].  Outputs dev: rays_o/rays_d/target_rgb/d_cam [n,3],
 * target_d [n], pose_idx int32 [n]. */
int rfx_gather_rays(const float* kf_rays, int64_t rays_per_kf, int64_t num_kf, const int64_t* kf_frame_ids, int keyframe_every,
                    const float* cur_rays, int64_t cur_population, int64_t n_kf_samples, int64_t n_cur, uint64_t seed_kf,
                    uint64_t seed_cur, const float* poses16, int K, float* rays_o, float* rays_d, float* target_rgb,
                    float* target_d, float* d_cam, int32_t* pose_idx, rfx_stream stream);
/* its backward w.r.t. the poses: dposes16 dev [K,16] (overwritten) from g_o/g_d dev [n,3] (either may be NULL). */
int rfx_pose_grad(const float* g_o, const float* g_d, const float* d_cam, const int32_t* pose_idx, int64_t n, int K,
                  float* dposes16, rfx_stream stream);

/* Fused render (SLAM.render_single, mp_slam/slam.py:290-344): S1 + points + Q1 + R1 in one launch,
 * one wave per ray; nothing but rays, rgb and depth touches HBM.  u01 dev [n,S] (NULL = no jitter). */
int rfx_render_rays(const rfx_field_desc* f, const rfx_sampler_desc* s, const float* rays_o,
                    const float* rays_d, const float* target_d, const float* u01, int64_t n_rays,
                    const double bbox[6], int bbox_f64, float sc_factor, float* rgb, float* depth,
                    rfx_stream stream);

/* ======================================================================================
 * Tracker kernels of the random-optimisation pose search (model/ROtracker.py), SURVEY 8(f1).
 * ==================================================================================== */

/* T1: replaces `compute_vertex` model/ROtracker.py:272-344 (host :426-451).  depth dev [H*W] ->
 * vertex4 dev [H*W,4] = back-projected (x,y,z) of depth + per-row jitter, and the target tsdf.
 * The reference draws the jitter from cuRAND (curand_init(seed,row,0)); here u_rows dev [H,2]
 * supplies the two uniforms per row, or (NULL) a counter-based hash of (seed,row) does.  With
 * RO.sample_range = 0 (every shipped config) the jitter is exactly 0. */
int rfx_track_vertex(const float* depth, float* vertex4, const float K[9], int H, int W, float cut_dist, float trunc,
                     float sample_range, uint32_t seed, const float* u_rows, rfx_stream stream);

/* T2: replaces `compute_normal` model/ROtracker.py:346-403 (host :453-468): central-difference
 * normals, flipped towards the camera; border pixels are not written (caller zero-fills once). */
int rfx_track_normal(const float* vertex4, float* normal3, int H, int W, rfx_stream stream);

/* T3: replaces `compute_tsdf_value` model/ROtracker.py:144-270 (host :536-604).  For each of the
 * n_candidates pose perturbations q6 dev [n,6] (translation, quaternion vector part; scaled by
 * search_size[6], host) around the pose (R[9] row-major, T[3], host): sum over the sub-sampled pixels
 * (stride `level`, offset `level_index`) of |tsdf(nearest voxel of the transformed vertex) - target|
 * into value_q30 dev [n], hit count into count dev [n] (both int64, both overwritten).
 * ABI 8: the sum is kept in FIXED POINT -- every term (<= 2) truncated to a multiple of 2^-30 and added as an integer -- so
 * that it does not depend on the order of the additions (the reference adds its pixels with float atomics in whatever order
 * they arrive: model/ROtracker.py:262-266): value = (float)((double)value_q30 * 2^-30). */
int rfx_track_evaluate(const float* tsdf, int dx, int dy, int dz, const float origin[3], float voxel,
                       const float* vertex4, const float* normal3, const float R[9], const float T[3],
                       const float* q6, const float search_size[6], int n_candidates, const float K[9], int H, int W,
                       int level, int level_index, int64_t* value_q30, int64_t* count, rfx_stream stream);

/* T3 on one x-slab [x0, x1) of the volume (tsdf holds the slab's planes): value_q30 / count receive the terms of the pixels
 * whose nearest voxel lies in the slab; added over the slabs (integers: exactly) they ARE rfx_track_evaluate's. */
int rfx_track_evaluate_slab(const float* tsdf, int dx, int dy, int dz, int x0, int x1, const float origin[3], float voxel,
                            const float* vertex4, const float* normal3, const float R[9], const float T[3],
                            const float* q6, const float search_size[6], int n_candidates, const float K[9], int H, int W,
                            int level, int level_index, int64_t* value_q30, int64_t* count, rfx_stream stream);

/* ---- the whole pose search of one frame on the device (round 4) ------------------------------------------------------
 * `random_optimization` (model/ROtracker.py:713-831) drives T3 from the host: 20 x (launch, copy 2 x P floats back, pick the
 * candidates that beat candidate 0 in Python -- cal_transform :606-709 --, move the pose, rescale the search box :493-534).
 * Here the pose, the search box and the loop's flags live in `state` on the device, T3 reads them from there and a one-block
 * kernel does cal_transform + the bookkeeping, so a frame's search is 2 launches per iteration and the host reads `state`
 * once, at the end.  Arithmetic follows the host loop (float32 state, float64 weighted sums in candidate order).
 *
 * templates[k] / template_rows[k] / n_eval[k] / level[k], k = `count_particle` 0..19: the template
 * `ALL_PST[...]` of tiff_index[k] (dev [rows,6]), the number of its candidates T3 evaluates (`int(PST_size/1024)*1024`; the rest
 * count as sum 0, as on the host) and depth_level[k].  x0, x1: the x-planes `tsdf` holds (whole volume: 0, dx).
 * state words (float unless noted): 0..8 R, 9..11 T, 12..17 search_size, 18..23 previous_search_size, 24 min_tsdf of the last
 * iteration; int32: 32 count_particle, 33 level_index, 34 success, 35 previous_success, 36 success of iteration 0,
 * 37 error (1 = a selected candidate had 1 - |q|^2 < 0: the reference exits there, :662-669), 38 successful iterations,
 * 39 iterations done.  value / count: dev [max template_rows] scratch, owned by the search between begin and the last update. */
#define RFX_TRACK_STEPS 20
#define RFX_TRACK_STATE_WORDS 64
#define RFX_TRACK_MAX_COUNT_SEARCH 512
typedef struct rfx_track_search {
    const float* tsdf; int32_t dx, dy, dz, x0, x1; float origin[3]; float voxel;
    const float* vertex4; const float* normal3;
    const float* templates[RFX_TRACK_STEPS];
    int32_t template_rows[RFX_TRACK_STEPS], n_eval[RFX_TRACK_STEPS], level[RFX_TRACK_STEPS];
    float K[9]; int32_t H, W;
    int32_t count_search, fix_level_index, iterative_scale, reserved;
    double scaling_coefficient, beta;            /* RO.scaling_coefficient; the smoothing of the box (0.9 in the reference) */
    float* state; int64_t* value_q30; int64_t* count;     /* sums as in rfx_track_evaluate: [max template rows] each */
} rfx_track_search;
/* begin: state <- (R, T, search_size), flags cleared, level_index 5, value_q30/count zeroed.  evaluate: T3 of the current state
 * (adds this volume's -- or slab's -- sums into value_q30/count).  update: cal_transform + bookkeeping of iteration `iteration`,
 * then zeroes them.  Between evaluate and update a rank of a sharded volume adds value_q30/count over the ranks (int64: exact).
 * run = begin + iterations x (evaluate, update). */
size_t rfx_track_search_bytes(void);           /* sizeof(rfx_track_search), for bindings to check their mirror */
int rfx_track_search_begin(const rfx_track_search* s, const float R[9], const float T[3], const float search_size[6], rfx_stream stream);
int rfx_track_search_evaluate(const rfx_track_search* s, rfx_stream stream);
int rfx_track_search_update(const rfx_track_search* s, int iteration, rfx_stream stream);
int rfx_track_search_run(const rfx_track_search* s, const float R[9], const float T[3], const float search_size[6], int iterations,
                         rfx_stream stream);

/* ---- iso-surface extraction -------------------------------------------------------
 * MC1/MC2 replace the host `skimage.measure.marching_cubes(raw, level=isolevel, mask=mask)` call of
 * utils.py:158 (extract_mesh_github).  volume dev [X,Y,Z] fp32 (z fastest), mask dev [X,Y,Z] u8 or NULL
 * (a cell is polygonised only when its 8 samples are unmasked and not NaN); a sample < level is inside.
 * Cells are numbered (x*(Y-1) + y)*(Z-1) + z.  Tables are host-generated (remixfusion_amd/mesh.py):
 * n_tri dev [256] int32; tri_edges dev [256, max_tri*3] int32 edge ids 0..11 in the order
 * (0,1)(0,2)(0,4)(1,3)(1,5)(2,3)(2,6)(3,7)(4,5)(4,6)(5,7)(6,7) of corner ids cx + 2cy + 4cz. */
int rfx_mc_count(const float* volume, const uint8_t* mask, int X, int Y, int Z, float level, const int* n_tri, int* counts,
                 rfx_stream stream);
/* offsets dev [cells] int64 = exclusive scan of counts; verts dev [3*total, 3] fp32 in sample-index
 * coordinates; keys dev [3*total] int64 identify the cut grid edge (equal keys == the same vertex). */
int rfx_mc_emit(const float* volume, const uint8_t* mask, int X, int Y, int Z, float level, const int* tri_edges, int max_tri,
                const int* counts, const long long* offsets, float* verts, long long* keys, rfx_stream stream);

/* ---- RBA pose-refinement MLP -------------------------------------------------------
 * P1-P3 replace the torch graph of model/rba.py:60-100 (RBA.forward: Linear(7,256)-ELU-[Linear(256,256)-ELU]x2
 * -Linear(256,6), residual * scale, camera 0 pinned, kornia angle_axis_to_rotation_matrix, make_c2w) and
 * its autograd backward, which mp_slam/mapper.py:456,489-497 run in every bundle-adjustment iteration.
 * Weights are torch Linear layouts ([out,in] row-major) on the device; hidden must be 256. */
typedef struct rfx_rba_params {
    const float *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3;
    int32_t hidden;
} rfx_rba_params;
typedef struct rfx_rba_grads {          /* any pointer may be NULL (that gradient is skipped) */
    float *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3;
} rfx_rba_grads;
size_t rfx_rba_acts_floats(int64_t K);   /* size of `acts` (forward -> backward) */
size_t rfx_rba_grads_floats(int64_t K);  /* size of the backward workspace */
/* cam_ids dev [K] int64 (rows of init_r/init_t dev [num_cams,3]); poses16 dev [K,16] row-major c2w. */
/* Per-frame pose bookkeeping of the tracker side (model/ROtracker.py:911-945, mp_slam/tracker.py): est_c2w16 <- c2w16 and,
 * when rel_c2w16 is given (non-keyframes), rel_c2w16 <- c2w16 @ inverse(kf_c2w16), the pose relative to the newest
 * keyframe.  All dev [16], row-major. */
int rfx_frame_pose(const float* c2w16, const float* kf_c2w16, float* est_c2w16, float* rel_c2w16, rfx_stream stream);
/* RBA.update_init_pose (model/rba.py:77-86): c2w16 dev [16] row-major pose of keyframe slot cam_id -> init_c2w16[cam_id],
 * init_t[cam_id] (translation), init_r[cam_id] (rotation as angle-axis).  init_r / init_t dev [num_cams,3], init_c2w16 dev
 * [num_cams,16]. */
int rfx_rba_set_init_pose(const float* c2w16, int cam_id, int num_cams, float* init_r, float* init_t, float* init_c2w16,
                          rfx_stream stream);
int rfx_rba_forward(const rfx_rba_params* p, const float* init_r, const float* init_t, const int64_t* cam_ids, int64_t K,
                    int num_cams, float scale, float* poses16, float* acts, rfx_stream stream);
/* dposes16 dev [K,16] = dL/dc2w; parameter gradients are overwritten (deterministic sums over K). */
int rfx_rba_backward(const rfx_rba_params* p, const float* acts, int64_t K, const float* dposes16, float scale,
                     const rfx_rba_grads* g, float* workspace, rfx_stream stream);

/* ---- one bundle-adjustment iteration, forward + backward, in one call (M1) -------------------------------
 * Sequences the entry points above for the loop bodies of Mapper.global_mapping / global_pose
 * (mp_slam/mapper.py:394-420, 470-505): rfx_gather_rays -> rfx_sample_z -> rfx_ray_points -> rfx_field_forward ->
 * rfx_composite_forward -> rfx_mapping_loss_forward -> TV term (rfx_tv_lattice, rfx_grid_encode_forward, rfx_tv_forward)
 * -> rfx_mapping_loss_backward -> rfx_field_backward_chain/_weights[/_scatter(dx)/_dx -> ray gradients -> rfx_pose_grad]
 * -> rfx_tv_backward -> rfx_field_backward_scatter_merged.  Optimizers and random draws stay with the caller. */
typedef struct rfx_ba_desc {
    rfx_field_desc   field;             /* clamp mode of this phase; `staged`, if set, is refreshed from     */
                                        /* w1..w4 at the start of the call (rfx_field_stage_weights)         */
    rfx_sampler_desc sampler;
    double        bbox[6];
    int32_t       bbox_f64;
    float         sc_factor, depth_trunc, trunc;
    int32_t       rgb_missing_on;
    const float*  loss_w_dev;           /* dev [4]: d total / d (rgb, depth, sdf, fs) loss                   */
    int32_t       tv_P;                 /* lattice points per axis (training.smooth_pts - 1)                 */
    float         tv_voxel, tv_margin;
    float         tv_scale;             /* training.smooth_weight / smooth_pts^3                             */
    int32_t       tv_normalise;
    const float*  kf_rays;              /* see rfx_gather_rays                                               */
    int64_t       rays_per_kf, num_kf;
    const int64_t* kf_frame_ids;
    int32_t       keyframe_every;
    const float*  cur_rays;
    int64_t       cur_population, n_kf_samples, n_cur;
    uint64_t      seed_kf, seed_cur;
    const float*  poses16;              /* dev [K,16]                                                        */
    int32_t       K;
    const float*  u_z;                  /* dev [n,S] uniforms of the sampler jitter, or NULL                 */
    const float*  u6;                   /* dev [6] uniforms of the TV lattice                                */
    uint64_t      seed_u;               /* != 0: the call draws both sets of uniforms itself (u_z, u6 not read): element e */
                                        /* of rfx_uniform_draws(seed_u, 0, n*S) jitters sample e, (seed_u, 1, 6) is u6      */
    int64_t       hash_entries;         /* entries (of n_feat floats) of the hash table = size of d_hash     */
    float*        d_hash;               /* out dev: hash-table gradient (zeroed here); d_hash and d_w both NULL: */
    float*        d_w;                  /* out dev [5312]: dW1 | dW2 | dW3 | dW4 (zeroed here)  | no map gradients */
                                        /* (pose phase of global_pose, where no optimizer consumes them, mapper.py:494-499:
                                         * the weight-gradient, table-scatter and TV-gradient stages are not run; d_poses16
                                         * is then required, and the TV term is evaluated only if tv_sum is given)         */
    float*        d_poses16;            /* out dev [K,16] or NULL (poses fixed: no ray gradients computed)   */
    float*        losses8;              /* out dev [8] or NULL: the four losses, then their coefficients     */
    double*       tv_sum;               /* out dev [1] or NULL: un-normalised TV sum (only its gradient enters the
                                         * update, so the value is evaluated just when asked for)            */
    /* optional (ABI 5), with d_poses16: carry on into the pose MLP's backward -- rfx_rba_backward(rba, rba_acts, K,
     * d_poses16, rba_scale, rba_grads, rba_ws) -- inside the call: the ray-gradient reduction, rfx_pose_grad and the MLP
     * backward then share one launch (same results to rounding: the sums are grouped differently).  All NULL: not done. */
    const struct rfx_rba_params* rba;
    const float*  rba_acts;             /* dev: what rfx_rba_forward left for these K cameras                */
    float         rba_scale;
    const struct rfx_rba_grads* rba_grads;
    float*        rba_ws;               /* dev, rfx_rba_grads_floats(K)                                      */
    /* optional (ABI 10): host array of RFX_BA_STAGE_EVENTS events (rfx_event_create / hipEventCreate), or NULL.  The call
     * records entry [i] on `stream` BEHIND the last launch of stage i (RFX_BA_EV_*), so the time between two consecutive
     * recorded entries is that stage's device time inside the one-call iteration -- the same launches, nothing un-fused,
     * no host round trip (bench.py's roofline; the stage-by-stage issue costs an iteration ~0.15 ms).  Stages a phase does
     * not run record nothing; a NULL entry is skipped. */
    const rfx_event* stage_events;
} rfx_ba_desc;
#define RFX_BA_EV_START     0   /* in front of the first launch                                                        */
#define RFX_BA_EV_PROLOGUE  1   /* ray batch, S1, points (+ weight staging, TV lattice and lookups, zero-fill)          */
#define RFX_BA_EV_FORWARD   2   /* rfx_field_forward_stash                                                             */
#define RFX_BA_EV_LOSS      3   /* R1 + L1 forward / backward, TV backward (+ the TV value, if asked)                  */
#define RFX_BA_EV_CHAIN     4   /* row selection + backward chain (+ loss finalize)                                    */
#define RFX_BA_EV_WEIGHTS   5   /* decoder weight gradients (map gradients only; in the map phase their second stage   */
                                /* rides in the scatter's staging launch: counted there)                               */
#define RFX_BA_EV_SCATTER   6   /* table scatter of ray samples + TV lattice (map gradients only)                      */
#define RFX_BA_EV_DX_TABLE  7   /* pose phase: d loss / d x01 through the table                                        */
#define RFX_BA_EV_DX        8   /* pose phase: + OneBlob / GBV part                                                    */
#define RFX_BA_EV_POSE      9   /* pose phase: ray reduction, pose gradient, pose-MLP backward                         */
#define RFX_BA_STAGE_EVENTS 10
size_t rfx_ba_desc_bytes(void);          /* sizeof(rfx_ba_desc): lets a foreign binding verify its mirror of the struct */
size_t rfx_ba_workspace_bytes(int64_t n_rays, int S, int tv_P, int n_feat_total, int n_levels);
/* the same with the scatter's share sized by rfx_grid_encode_backward_workspace_bytes_for(hash, points): whatever lies behind
 * the minimum is the scatter's (its region is the workspace's last) */
size_t rfx_ba_workspace_bytes_for(int64_t n_rays, int S, int tv_P, const rfx_grid_desc* hash);
/* workspace: dev, 256-byte aligned, >= rfx_ba_workspace_bytes(n_kf_samples + n_cur, S, tv_P, L*F, L). */
int rfx_ba_forward_backward(const rfx_ba_desc* b, void* workspace, size_t workspace_bytes, rfx_stream stream);
/* Where the iteration's intermediates live inside the workspace after a call (byte offsets from its base), so that a
 * caller -- the parity tests -- can read the ray batch the call drew and every stage's result:
 *   [0] rays_o [n,3]  [1] rays_d [n,3]  [2] target rgb [n,3]  [3] target depth [n]  [4] d_cam [n,3]  [5] pose index int32 [n]
 *   [6] z_vals [n,S]  [7] x01 [n*S,3]   [8] raw [n*S,4]       [9] rgb_map [n,3]     [10] depth_map [n]
 *   [11] TV lattice points [P^3,3]      [12] TV features [P^3, n_feat_total]         [13] d_raw [n*S,4]  [14] dx01 [n*S,3]
 * Returns the number of offsets written (<= count), or RFX_ERR_ARG. */
#define RFX_BA_LAYOUT_FIELDS 15
int rfx_ba_workspace_layout(int64_t n_rays, int S, int tv_P, int n_feat_total, int n_levels, size_t* offsets, int count);

/* ---- the same iteration with the hash table partitioned by level over `world` GPUs (ABI 6) ----------------------------
 * Every rank fills the same rfx_ba_desc (same seeds: the same ray batch and lattice on every rank; field.hash describes
 * the FULL grid, field.hash_table / d_hash are full-size buffers of which only the own levels' part is read / written) and
 * calls the phases in turn; between them the caller moves the exchange buffers with its collective library
 * (torch.distributed: RCCL over xGMI):
 *   rfx_ba_shard_lookup   ray batch of all n rays (+ TV lattice and its own-level features, zero-fill of the own part of
 *                         d_hash), own levels' features of all n*S points -> feat_send [n*S, 2k]
 *      all-to-all: rank q's rays' rows of feat_send -> q's feat_recv
 *   rfx_ba_shard_render   stash <- feat_recv; decoder forward, compositing, losses and their gradient, backward chain and
 *                         weight gradients on the OWN rays -> demb_send, d_w (partial), loss_sums8 (partial)
 *      all-to-all: demb_send block q -> q's demb_recv;  all-reduce: d_w, loss_sums8
 *   rfx_ba_shard_scatter  map gradients: own levels' table gradient from demb_recv (all points) + the TV term of the own
 *                         levels -> d_hash;  pose phase: d loss / d x01 through the own levels, all points -> dx_send
 *      all-to-all (pose phase): rank q's rows of dx_send -> q's dx_recv[rank]
 *   rfx_ba_shard_pose     (pose phase) sums dx_recv over the ranks, adds the OneBlob / GBV part, reduces to ray and pose
 *                         gradients of the own rays -> d_poses16 (partial: all-reduce, then rfx_rba_backward)
 * The loss coefficients need no exchange: they are made of counts over target and sample depths (rfx_ba_forward_backward),
 * which every rank has for the whole batch.  Results equal the single-GPU iteration's up to the order of the sums. */
typedef struct rfx_ba_shard {
    int32_t      rank, world;                          /* 1 <= world <= RFX_MAX_LEVELS                                  */
    int32_t      level_start[RFX_MAX_LEVELS + 1];      /* rank q owns hash levels [level_start[q], level_start[q+1])    */
    int64_t      ray_start[RFX_MAX_LEVELS + 1];        /* rank q renders rays [ray_start[q], ray_start[q+1]) of the n   */
    float*       feat_send;                            /* dev [n*S, 2k]: k = own levels                                 */
    const float* feat_recv;                            /* dev: for q = 0..world-1 in turn [n_own*S, 2k_q]               */
    float*       demb_send;                            /* dev: for q = 0..world-1 in turn [n_own*S, 2k_q]               */
    const float* demb_recv;                            /* dev [n*S, 2k]                                                 */
    double*      loss_sums8;                           /* dev double[8]: the own rays' loss sums                        */
    float*       dx_send;                              /* dev [n*S, 3] (pose phase)                                     */
    const float* dx_recv;                              /* dev [world, n_own*S, 3] (pose phase)                          */
} rfx_ba_shard;
size_t rfx_ba_shard_bytes(void);        /* sizeof(rfx_ba_shard), for foreign bindings */
/* workspace for all four: the one rfx_ba_forward_backward takes (rfx_ba_workspace_bytes[_for] of the WHOLE batch). */
int rfx_ba_shard_lookup(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream);
/* rfx_ba_shard_lookup in two launches (ABI 9), for a caller that overlaps the feature all-to-all with what the exchange does
 * not need: _rays = the ray batch + the own levels' features of its points -> feat_send (start the all-to-all after it);
 * _tv = the TV lattice with its own-level features + the zero-fill of the own range of d_hash (issue it while the exchange is
 * in flight; it must precede rfx_ba_shard_render on the stream).  _rays followed by _tv == rfx_ba_shard_lookup, bit for bit.
 * The reference has no counterpart (single GPU): the iteration is mp_slam/mapper.py:392-423 / :470-505. */
int rfx_ba_shard_lookup_rays(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_ba_shard_lookup_tv(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_ba_shard_render(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_ba_shard_scatter(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream);
int rfx_ba_shard_pose(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream);

/* ---- optimizer step (M1) ------------------------------------------------------------------------------------
 * torch.optim.Adam as the reference builds it (mp_slam/slam.py:271-286; betas (0.9, 0.99), per-group lr / eps /
 * L2 weight decay, no amsgrad) and steps it (mp_slam/mapper.py:416-418, 497-499): all tensors of one optimizer in
 * one launch.  The caller keeps the state (exp_avg, exp_avg_sq, step count) and passes the step-dependent scalars:
 *   neg_step_size = -lr / (1 - beta1^step), bias_correction2_sqrt = sqrt(1 - beta2^step).                          */
#define RFX_ADAM_MAX_TENSORS 16
typedef struct rfx_adam_tensor {
    float*       param;                 /* dev [n], updated in place                                         */
    const float* grad;                  /* dev [n]                                                           */
    float*       exp_avg;               /* dev [n]                                                           */
    float*       exp_avg_sq;            /* dev [n]                                                           */
    int64_t      n;
    float        beta1, beta2, one_minus_beta1, one_minus_beta2;
    float        eps, weight_decay, neg_step_size, bias_correction2_sqrt;
} rfx_adam_tensor;
size_t rfx_adam_tensor_bytes(void);      /* sizeof(rfx_adam_tensor), for foreign bindings                     */
/* tensors: host array of `count` <= RFX_ADAM_MAX_TENSORS descriptors (read before the call returns). */
int rfx_adam_step(const rfx_adam_tensor* tensors, int count, rfx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* RFX_H */
