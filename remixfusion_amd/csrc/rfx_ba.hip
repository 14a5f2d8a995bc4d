// rfx_ba.hip -- one bundle-adjustment iteration (forward + backward) issued by ONE host call.
// Reference: the loop bodies of Mapper.global_mapping / global_pose (mp_slam/mapper.py:394-420, 470-505):
// ray batch -> JointEncoding.mapping (S1 sampler, points, Q1 field, R1 compositing, L1 losses) ->
// get_loss_from_ret(smooth=True) (TV1) -> loss.backward().  Every stage is an existing librfx entry point; this
// file only sequences them on one stream from a carved workspace, so the host pays one foreign call per
// iteration instead of ~20 (the Python side keeps the optimizers and the random draws).
#include "rfx_common.h"
#include <algorithm>

namespace rfx {

// d rays_o = sum_s dx / extent, d rays_d = sum_s z * dx / extent (reference: autograd through
// pts = o + d z, then the bound normalisation; scene_rep.py:388,443).  One wave per ray.
__global__ __launch_bounds__(256) void ray_grad_reduce_kernel(const float* __restrict__ dx, const float* __restrict__ z,
                                                              int64_t n, int S, float ex, float ey, float ez,
                                                              float* __restrict__ go, float* __restrict__ gd) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n) return;
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = lane; s < S; s += 64) {
        const float* g = dx + (ray * S + s) * 3;
        const float zz = z[ray * S + s];
        const float px = g[0] / ex, py = g[1] / ey, pz = g[2] / ez;
        a[0] += px; a[1] += py; a[2] += pz;
        a[3] += px * zz; a[4] += py * zz; a[5] += pz * zz;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[q] += __shfl_xor(a[q], o);
    }
    if (lane == 0) {
        go[ray * 3] = a[0]; go[ray * 3 + 1] = a[1]; go[ray * 3 + 2] = a[2];
        gd[ray * 3] = a[3]; gd[ray * 3 + 1] = a[4]; gd[ray * 3 + 2] = a[5];
    }
}

struct BaWs {
    float *o, *d, *tgt, *d_cam, *td, *z, *x01, *raw, *rgb_map, *depth_map, *lc, *pts, *feat, *d_raw, *dx, *go, *gd, *dfeat, *ones;
    int *pidx, *ray_cnt;
    double *sums, *cnt, *lsum;
    void *bwd_ws, *scat_ws;
    size_t bwd_bytes, scat_bytes, total;
};

static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

// scat_bytes_given: bytes the caller's workspace has for the scatter region (the last one); 0 = the minimum
static BaWs carve_ba(void* base, int64_t n, int S, int P, int n_feat, int n_levels, size_t scat_bytes_given = 0) {
    BaWs w;
    size_t off = 0;
    char* b0 = reinterpret_cast<char*>(base);
    auto take = [&](size_t bytes) { char* r = b0 ? b0 + off : nullptr; off += al(bytes); return r; };
    const size_t nS = (size_t)n * S, nt = (size_t)P * P * P;
    w.o = (float*)take(n * 12); w.d = (float*)take(n * 12); w.tgt = (float*)take(n * 12); w.d_cam = (float*)take(n * 12);
    w.td = (float*)take(n * 4); w.pidx = (int*)take(n * 4); w.ray_cnt = (int*)take(n * 4);
    w.z = (float*)take(nS * 4); w.x01 = (float*)take(nS * 12); w.raw = (float*)take(nS * 16);
    w.rgb_map = (float*)take(n * 12); w.depth_map = (float*)take(n * 4);
    w.sums = (double*)take(RFX_LOSS_WS_DOUBLES * 8); w.cnt = (double*)take(2048 * 4 * 8); w.lsum = (double*)take(1024 * 8 * 8); w.lc = (float*)take(8 * 4); w.ones = (float*)take(4);
    w.pts = (float*)take(nt * 12); w.feat = (float*)take(nt * n_feat * 4); w.dfeat = (float*)take(nt * n_feat * 4);
    w.d_raw = (float*)take(nS * 16); w.dx = (float*)take(nS * 12); w.go = (float*)take(n * 12); w.gd = (float*)take(n * 12);
    w.bwd_bytes = rfx_field_backward_workspace_bytes((int64_t)nS);
    w.bwd_ws = take(w.bwd_bytes);
    w.scat_bytes = std::max(rfx_grid_encode_backward_workspace_bytes((int64_t)(nS + nt), n_levels), scat_bytes_given);
    w.scat_ws = take(w.scat_bytes);
    w.total = off;
    return w;
}

// dst[i] = sum over q of src[q * stride + i] (in rank order): the ranks' partial d loss / d x01 of the own points
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ src, int parts, int64_t stride, int64_t count,
                                                        float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float a = 0.f;
    for (int q = 0; q < parts; ++q) a += src[q * stride + i];
    dst[i] = a;
}

// what a rank of a level-partitioned iteration owns (rfx_ba_shard): levels [l0, l1) as a sub-grid with the true table
// offsets, rays [r0, r1) of the batch
struct ShardGeom {
    int l0, l1, k;
    int64_t r0, r1, n_own;
    rfx_grid_desc own;
};

static int shard_geom(const rfx_ba_desc* b, const rfx_ba_shard* s, int64_t n, ShardGeom* g) {
    if (!s || s->world < 1 || s->world > RFX_MAX_LEVELS || s->rank < 0 || s->rank >= s->world) return RFX_ERR_ARG;
    const int L = b->field.hash.n_levels;
    if (s->level_start[0] != 0 || s->level_start[s->world] != L || s->ray_start[0] != 0 || s->ray_start[s->world] != n) return RFX_ERR_ARG;
    for (int q = 0; q < s->world; ++q)
        if (s->level_start[q + 1] <= s->level_start[q] || s->ray_start[q + 1] < s->ray_start[q]) return RFX_ERR_ARG;
    g->l0 = s->level_start[s->rank]; g->l1 = s->level_start[s->rank + 1]; g->k = g->l1 - g->l0;
    g->r0 = s->ray_start[s->rank]; g->r1 = s->ray_start[s->rank + 1]; g->n_own = g->r1 - g->r0;
    g->own = b->field.hash;
    g->own.n_levels = g->k;
    for (int i = 0; i < RFX_MAX_LEVELS; ++i) {
        const int l = i < g->k ? g->l0 + i : g->l1 - 1;        // (unused slots repeat the last level: never read)
        g->own.scale[i] = b->field.hash.scale[l]; g->own.res[i] = b->field.hash.res[l]; g->own.size[i] = b->field.hash.size[l];
        g->own.offset[i] = b->field.hash.offset[l]; g->own.hashed[i] = b->field.hash.hashed[l];
    }
    return RFX_OK;
}

// the level rows of the blocks [points, 2 k_q], q = 0..world-1 in turn, at `base`
static void shard_rows(const rfx_ba_shard* s, const float* base, int64_t points, rfx_level_rows* r) {
    for (int q = 0; q < s->world; ++q) {
        const int a = s->level_start[q], e = s->level_start[q + 1];
        for (int l = a; l < e; ++l) {
            r->rows[l] = base + (size_t)points * 2 * a;
            r->ld[l] = 2 * (e - a);
            r->col[l] = 2 * (l - a);
        }
    }
}

struct ShardCall {
    int64_t n, nS, nt;
    int S, P, L, F;
    bool map_grads;
    BaWs w;
    ShardGeom g;
};

static int shard_call(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, ShardCall* c) {
    if (!b || !workspace) return RFX_ERR_ARG;
    c->n = b->n_kf_samples + b->n_cur;
    c->S = b->sampler.n_range_d + b->sampler.n_samples_d; c->P = b->tv_P;
    c->L = b->field.hash.n_levels; c->F = b->field.hash.n_feat;
    if (c->n <= 0 || c->S <= 0 || c->P <= 0 || !b->poses16 || b->K <= 0 || !b->loss_w_dev || b->hash_entries <= 0) return RFX_ERR_ARG;
    if (c->L != RFX_MAX_LEVELS || c->F != 2) return RFX_ERR_UNSUPPORTED;
    c->map_grads = b->d_hash != nullptr;
    if ((b->d_hash != nullptr) != (b->d_w != nullptr) || (!c->map_grads && !b->d_poses16)) return RFX_ERR_ARG;
    if ((uintptr_t)workspace & 255) return RFX_ERR_ARG;
    const size_t min_total = rfx_ba_workspace_bytes(c->n, c->S, c->P, c->L * c->F, c->L);
    if (workspace_bytes < min_total) return RFX_ERR_WORKSPACE;
    const size_t min_scat = rfx_grid_encode_backward_workspace_bytes(c->n * c->S + (int64_t)c->P * c->P * c->P, c->L);
    c->w = carve_ba(workspace, c->n, c->S, c->P, c->L * c->F, c->L, min_scat + ((workspace_bytes - min_total) & ~(size_t)255));
    c->nS = c->n * c->S; c->nt = (int64_t)c->P * c->P * c->P;
    int rc = shard_geom(b, s, c->n, &c->g);
    if (rc) return rc;
    if (!s->feat_send || !s->feat_recv || !s->demb_send || !s->demb_recv || !s->loss_sums8) return RFX_ERR_ARG;
    if (b->d_poses16 && (!s->dx_send || !s->dx_recv)) return RFX_ERR_ARG;
    return RFX_OK;
}

// level of the own sub-grid from which rfx_ba_shard_scatter writes the gradient (and rfx_ba_shard_lookup[_tv] skips the zero-fill):
// both calls see the same descriptor, so both get the same answer
static int shard_overwrite_level(const rfx_ba_desc* b, const ShardCall& c) {
    if (!c.map_grads || b->d_poses16) return c.g.k;
    return scatter_overwrite_from_level(c.g.own, c.nS + c.nt, true, b->d_hash);
}

}  // namespace rfx

using namespace rfx;

extern "C" {

size_t rfx_ba_desc_bytes(void) { return sizeof(rfx_ba_desc); }

size_t rfx_ba_workspace_bytes(int64_t n_rays, int S, int tv_P, int n_feat, int n_levels) {
    if (n_rays <= 0 || S <= 0 || tv_P <= 0 || n_feat <= 0 || n_levels <= 0) return 0;
    return carve_ba(nullptr, n_rays, S, tv_P, n_feat, n_levels).total;
}

size_t rfx_ba_workspace_bytes_for(int64_t n_rays, int S, int tv_P, const rfx_grid_desc* hash) {
    if (n_rays <= 0 || S <= 0 || tv_P <= 0 || !hash || hash->n_levels <= 0 || hash->n_feat <= 0) return 0;
    const int64_t pts = n_rays * S + (int64_t)tv_P * tv_P * tv_P;
    return carve_ba(nullptr, n_rays, S, tv_P, hash->n_levels * hash->n_feat, hash->n_levels,
                    rfx_grid_encode_backward_workspace_bytes_for(hash, pts)).total;
}

int rfx_ba_workspace_layout(int64_t n_rays, int S, int tv_P, int n_feat, int n_levels, size_t* offsets, int count) {
    if (n_rays <= 0 || S <= 0 || tv_P <= 0 || n_feat <= 0 || n_levels <= 0 || !offsets || count <= 0) return RFX_ERR_ARG;
    const BaWs w = carve_ba(reinterpret_cast<void*>((uintptr_t)256), n_rays, S, tv_P, n_feat, n_levels);   // any non-null base
    const void* f[RFX_BA_LAYOUT_FIELDS] = {w.o, w.d, w.tgt, w.td, w.d_cam, w.pidx, w.z, w.x01, w.raw, w.rgb_map, w.depth_map,
                                           w.pts, w.feat, w.d_raw, w.dx};
    const int k = std::min(count, (int)RFX_BA_LAYOUT_FIELDS);
    for (int i = 0; i < k; ++i) offsets[i] = (size_t)((uintptr_t)f[i] - 256);
    return k;
}

#define RFX_TRY(call)            \
    do {                         \
        int _rc = (call);        \
        if (_rc) return _rc;     \
    } while (0)

// rfx_ba_desc.stage_events: entry i recorded behind the last launch of stage i
#define RFX_MARK(i)                                                                                          \
    do {                                                                                                     \
        if (b->stage_events && b->stage_events[i])                                                           \
            RFX_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(b->stage_events[i]), st));               \
    } while (0)

int rfx_ba_forward_backward(const rfx_ba_desc* b, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    if (!b || !workspace) return RFX_ERR_ARG;
    const int64_t n = b->n_kf_samples + b->n_cur;
    const int S = b->sampler.n_range_d + b->sampler.n_samples_d, P = b->tv_P;
    const int L = b->field.hash.n_levels, F = b->field.hash.n_feat;
    if (n <= 0 || S <= 0 || P <= 0 || !b->poses16 || b->K <= 0 || !b->loss_w_dev || b->hash_entries <= 0) return RFX_ERR_ARG;
    // map gradients are optional as a pair: without them (pose phase, where no optimizer consumes them) only the
    // ray/pose gradients are produced, so there must be somewhere to put those
    const bool map_grads = b->d_hash != nullptr;
    if ((b->d_hash != nullptr) != (b->d_w != nullptr) || (!map_grads && !b->d_poses16)) return RFX_ERR_ARG;
    if ((uintptr_t)workspace & 255) return RFX_ERR_ARG;
    if (workspace_bytes < rfx_ba_workspace_bytes(n, S, P, L * F, L)) return RFX_ERR_WORKSPACE;
    // everything behind the minimum belongs to the scatter (its region is the last): more binned levels per group of launches
    const size_t min_total = rfx_ba_workspace_bytes(n, S, P, L * F, L);
    const size_t min_scat = rfx_grid_encode_backward_workspace_bytes(n * S + (int64_t)P * P * P, L);
    const BaWs w = carve_ba(workspace, n, S, P, L * F, L, min_scat + ((workspace_bytes - min_total) & ~(size_t)255));     // (whole 256-B units)
    hipStream_t st = as_stream(stream);
    const int64_t nS = n * S, nt = (int64_t)P * P * P;
    // ---- ray batch (rays, S1, points); beside it in the same launch: decoder weights -> MFMA operand image (the optimizers
    //      update the weights in place between calls), the TV lattice with its table lookups and the zero-fill of the hash gradient:
    //      none of them depends on anything computed here
    const bool tv_on = map_grads || b->tv_sum;
    if (!b->seed_u && tv_on && !b->u6) return RFX_ERR_ARG;
    const float trunc_loss = b->trunc * b->sc_factor;
    // map phase: the trailing hashed levels of a large table are WRITTEN by the scatter (one block per segment), so their part of
    // d_hash is neither zero-filled here nor read back there (round 6: 300 MB per iteration at T = 2^21)
    const int ow_level = (map_grads && !b->d_poses16) ? scatter_overwrite_from_level(b->field.hash, nS + nt, true, b->d_hash) : L;
    const int64_t zero_floats = !map_grads ? 0 : ow_level < L ? (int64_t)b->field.hash.offset[ow_level] * F : (int64_t)b->hash_entries * F;
    int n_cnt = 0;
    RFX_MARK(RFX_BA_EV_START);
    RFX_TRY(ba_prologue(b->kf_rays, b->rays_per_kf, b->num_kf, b->kf_frame_ids, b->keyframe_every, b->cur_rays, b->cur_population,
                        b->n_kf_samples, b->n_cur, b->seed_kf, b->seed_cur, b->poses16, b->K, &b->sampler, b->u_z, b->seed_u, b->bbox,
                        b->bbox_f64, w.o, w.d, w.tgt, w.td, w.d_cam, w.pidx, w.z, w.x01, &b->field, b->u6, P, b->tv_voxel,
                        b->tv_margin, b->tv_normalise, tv_on ? w.pts : nullptr, tv_on ? w.feat : nullptr,
                        map_grads ? b->d_hash : nullptr, zero_floats, trunc_loss, b->depth_trunc,
                        w.cnt, &n_cnt, stream));            // ... and counts what the loss coefficients are made of
    RFX_MARK(RFX_BA_EV_PROLOGUE);
    // ---- forward
    // ... which leaves its hash features in the backward workspace: the chain below does not look the table up again
    RFX_TRY(rfx_field_forward_stash(&b->field, w.x01, nS, w.raw, w.bwd_ws, w.bwd_bytes, stream));
    RFX_MARK(RFX_BA_EV_FORWARD);
    float* lc = b->losses8 ? b->losses8 : w.lc;       // out: the four losses, then their coefficients
    // ---- R1 + L1 forward and L1 backward in one launch (the coefficients come from the prologue's counts), TV1 backward beside
    //      them; leaves the loss partial sums, the rows with a gradient per ray and d_raw
    int n_partials = 0;
    RFX_TRY(composite_loss_grad(w.raw, w.z, w.tgt, w.td, n, S, b->trunc, b->sc_factor, trunc_loss, b->depth_trunc, b->rgb_missing_on,
                                w.rgb_map, w.depth_map, w.lsum, &n_partials, w.cnt, n_cnt, b->loss_w_dev, w.d_raw, w.ray_cnt,
                                map_grads ? w.feat : nullptr, P, L * F, b->tv_scale, map_grads ? w.dfeat : nullptr, stream));
    // the TV term depends on the hash table only: without map gradients it is evaluated just for its value, if asked
    if (b->tv_sum) RFX_TRY(rfx_tv_forward(w.feat, P, L * F, b->tv_sum, stream));      // (its features: the prologue's)
    RFX_MARK(RFX_BA_EV_LOSS);
    // ---- backward: the chain variant that produces exactly what the following stages read; its selection launch also
    //      finishes the losses
    int finalized = 0;
    RFX_TRY(field_backward_chain_stashed_counted(&b->field, w.x01, nS, w.d_raw, w.bwd_ws, w.bwd_bytes,
                                                 map_grads && b->d_poses16 ? 0 : map_grads ? 2 : 1, w.ray_cnt, S, w.lsum, n_partials, lc,
                                                 &finalized, stream));
    if (!finalized) RFX_TRY(loss_finalize_launch(w.lsum, n_partials, n, S, lc, stream));
    RFX_MARK(RFX_BA_EV_CHAIN);
    float* dw1 = b->d_w; float* dw2 = dw1 ? dw1 + 32 * 81 : nullptr; float* dw3 = dw1 ? dw2 + 16 * 32 : nullptr;
    float* dw4 = dw1 ? dw3 + 32 * 66 : nullptr;
    if (map_grads && !b->d_poses16) {     // map phase: weight gradients and table scatter back to back (they share a launch)
        RFX_TRY(field_backward_weights_scatter(&b->field, w.x01, nS, w.d_raw, dw1, dw2, dw3, dw4, w.pts, w.dfeat, nt, b->d_hash, w.bwd_ws,
                                               w.bwd_bytes, w.scat_ws, w.scat_bytes, stream, ow_level < L ? ow_level : RFX_MAX_LEVELS + 1,
                                               b->stage_events ? b->stage_events[RFX_BA_EV_WEIGHTS] : nullptr));
        RFX_MARK(RFX_BA_EV_SCATTER);
        return RFX_OK;
    }
    if (map_grads) {
        RFX_TRY(field_backward_weights_overwrite(nS, w.d_raw, dw1, dw2, dw3, dw4, w.bwd_ws, w.bwd_bytes, stream));
        RFX_MARK(RFX_BA_EV_WEIGHTS);
    }
    if (b->d_poses16) {
        RFX_TRY(rfx_field_backward_scatter(&b->field, w.x01, nS, nullptr, w.dx, w.bwd_ws, w.bwd_bytes, stream));
        RFX_MARK(RFX_BA_EV_DX_TABLE);
        RFX_TRY(rfx_field_backward_dx(&b->field, w.x01, nS, w.d_raw, w.dx, w.bwd_ws, w.bwd_bytes, stream));
        RFX_MARK(RFX_BA_EV_DX);
        int chained = 0;      // d rays -> d poses -> pose-MLP backward in one launch, when the caller hands the MLP over
        if (b->rba) {
            if (!b->rba_acts || !b->rba_grads || !b->rba_ws) return RFX_ERR_ARG;
            // one block per camera walks that camera's rays, 64 at a time: with few cameras (the first keyframes of a stream:
            // 2 048 keyframe rays over 1-5 poses, 400-2 000 current-frame rays on the last) the separate, ray-parallel
            // stages are faster; from ~800 rays on the heaviest camera downwards the single launch wins (30 -> 20 us at 13)
            const int64_t heaviest = b->n_cur + (b->n_kf_samples + std::max(b->K - 1, 1) - 1) / std::max(b->K - 1, 1);
            if (heaviest <= 800) RFX_TRY(pose_chain_backward(w.dx, w.z, w.d_cam, w.pidx, n, S, b->bbox, b->K, b->d_poses16, b->rba, b->rba_acts, b->rba_scale,
                                        b->rba_grads, b->rba_ws, stream, &chained));
        }
        if (!chained) {
            const float ex = (float)(b->bbox[1] - b->bbox[0]), ey = (float)(b->bbox[3] - b->bbox[2]), ez = (float)(b->bbox[5] - b->bbox[4]);
            hipLaunchKernelGGL(ray_grad_reduce_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, w.dx, w.z, n, S, ex, ey, ez, w.go, w.gd);
            RFX_LAUNCH_CHECK();
            RFX_TRY(rfx_pose_grad(w.go, w.gd, w.d_cam, w.pidx, n, b->K, b->d_poses16, stream));
            if (b->rba) RFX_TRY(rfx_rba_backward(b->rba, b->rba_acts, b->K, b->d_poses16, b->rba_scale, b->rba_grads, b->rba_ws, stream));
        }
        RFX_MARK(RFX_BA_EV_POSE);
    }
    if (map_grads) {
        RFX_TRY(rfx_field_backward_scatter_merged(&b->field, w.x01, nS, w.pts, w.dfeat, nt, b->d_hash, w.bwd_ws, w.bwd_bytes, w.scat_ws,
                                                  w.scat_bytes, stream));
        RFX_MARK(RFX_BA_EV_SCATTER);
    }
    return RFX_OK;
}

// ---- the same iteration on a level-partitioned table (rfx.h: rfx_ba_shard) ------------------------------------------------
size_t rfx_ba_shard_bytes(void) { return sizeof(rfx_ba_shard); }

// parts: 1 = the ray batch and the own levels' features of its points (what the feature all-to-all sends), 2 = the TV lattice
// with its own-level lookups and the zero-fill of the own range of d_hash (nothing the exchange needs: the caller issues it
// while the all-to-all is in flight), 3 = both in one launch (rfx_ba_shard_lookup)
static int shard_lookup_parts(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream, int parts) {
    ShardCall c;
    RFX_TRY(shard_call(b, s, workspace, workspace_bytes, &c));
    const BaWs& w = c.w;
    const bool tv_on = c.map_grads || b->tv_sum;
    if (!b->seed_u && tv_on && !b->u6) return RFX_ERR_ARG;
    // the part of the gradient buffer that belongs to the own levels (a contiguous range of levels: one contiguous range)
    const int64_t z0 = (int64_t)c.g.own.offset[0] * c.F;
    int64_t z1 = ((int64_t)c.g.own.offset[c.g.k - 1] + c.g.own.size[c.g.k - 1]) * c.F;
    // (map phase: the scatter writes the trailing hashed levels of the own range itself -- see rfx_ba_forward_backward)
    const int ow_level = shard_overwrite_level(b, c);
    if (ow_level < c.g.k) z1 = (int64_t)c.g.own.offset[ow_level] * c.F;
    int n_cnt = 0;
    RFX_TRY(ba_prologue(b->kf_rays, b->rays_per_kf, b->num_kf, b->kf_frame_ids, b->keyframe_every, b->cur_rays, b->cur_population,
                        b->n_kf_samples, b->n_cur, b->seed_kf, b->seed_cur, b->poses16, b->K, &b->sampler, b->u_z, b->seed_u, b->bbox,
                        b->bbox_f64, w.o, w.d, w.tgt, w.td, w.d_cam, w.pidx, w.z, w.x01, &b->field, b->u6, c.P, b->tv_voxel,
                        b->tv_margin, b->tv_normalise, tv_on ? w.pts : nullptr, tv_on ? w.feat : nullptr,
                        c.map_grads ? b->d_hash + z0 : nullptr, c.map_grads ? z1 - z0 : 0, b->trunc * b->sc_factor, b->depth_trunc,
                        w.cnt, &n_cnt, stream, &c.g.own, parts));
    if (!(parts & 1)) return RFX_OK;
    return rfx_grid_encode_forward(&c.g.own, b->field.hash_table, w.x01, c.nS, s->feat_send, stream);
}

int rfx_ba_shard_lookup(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return shard_lookup_parts(b, s, workspace, workspace_bytes, stream, 3);
}
int rfx_ba_shard_lookup_rays(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return shard_lookup_parts(b, s, workspace, workspace_bytes, stream, 1);
}
int rfx_ba_shard_lookup_tv(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    return shard_lookup_parts(b, s, workspace, workspace_bytes, stream, 2);
}

int rfx_ba_shard_render(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    ShardCall c;
    RFX_TRY(shard_call(b, s, workspace, workspace_bytes, &c));
    const BaWs& w = c.w;
    const int S = c.S;
    const int64_t m = c.g.n_own, mS = m * S;
    hipStream_t st = as_stream(stream);
    const float* x01 = w.x01 + c.g.r0 * S * 3;
    const float* z = w.z + c.g.r0 * S;
    int n_partials = 0;
    if (m > 0) {
        rfx_level_rows rows;
        shard_rows(s, s->feat_recv, mS, &rows);
        RFX_TRY(rfx_field_stash_put(&rows, mS, w.bwd_ws, w.bwd_bytes, stream));
        RFX_TRY(rfx_field_forward_stashed(&b->field, x01, mS, w.raw, w.bwd_ws, w.bwd_bytes, stream));
    }
    // R1 + L1 forward and backward of the own rays with the WHOLE batch's coefficients (the prologue counted all n rays on
    // every rank); the TV backward of the own levels beside them
    RFX_TRY(composite_loss_grad(w.raw, z, w.tgt + c.g.r0 * 3, w.td + c.g.r0, m, S, b->trunc, b->sc_factor, b->trunc * b->sc_factor,
                                b->depth_trunc, b->rgb_missing_on, w.rgb_map, w.depth_map, w.lsum, &n_partials, w.cnt,
                                ba_count_partials(c.n), b->loss_w_dev, w.d_raw, w.ray_cnt, c.map_grads ? w.feat : nullptr, c.P,
                                c.g.k * c.F, b->tv_scale, c.map_grads ? w.dfeat : nullptr, stream, c.n));
    if (m == 0 && c.map_grads) RFX_TRY(rfx_tv_backward(w.feat, c.P, c.g.k * c.F, b->tv_scale, nullptr, w.dfeat, stream));
    if (b->tv_sum) RFX_TRY(rfx_tv_forward(w.feat, c.P, c.g.k * c.F, b->tv_sum, stream));     // the own levels' share of the sum
    int finalized = 0;
    RFX_TRY(field_backward_chain_stashed_counted(&b->field, x01, mS, w.d_raw, w.bwd_ws, w.bwd_bytes,
                                                 c.map_grads && b->d_poses16 ? 0 : c.map_grads ? 2 : 1, w.ray_cnt, S, nullptr, 0, nullptr,
                                                 &finalized, stream));
    if (c.map_grads) {
        float* dw1 = b->d_w; float* dw2 = dw1 + 32 * 81; float* dw3 = dw2 + 16 * 32; float* dw4 = dw3 + 32 * 66;
        RFX_TRY(field_backward_weights_overwrite(mS, w.d_raw, dw1, dw2, dw3, dw4, w.bwd_ws, w.bwd_bytes, stream));
    }
    rfx_level_rows out;
    shard_rows(s, s->demb_send, mS, &out);
    if (n_partials == 0) RFX_HIP_TRY(hipMemsetAsync(s->loss_sums8, 0, 8 * sizeof(double), st));
    return rfx_field_backward_demb_rows(mS, &out, w.lsum, n_partials, s->loss_sums8, w.bwd_ws, w.bwd_bytes, stream);
}

int rfx_ba_shard_scatter(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    ShardCall c;
    RFX_TRY(shard_call(b, s, workspace, workspace_bytes, &c));
    const BaWs& w = c.w;
    if (c.map_grads) {
        const int ow_level = shard_overwrite_level(b, c);
        RFX_TRY(grid_encode_backward_merged_from(&c.g.own, b->field.hash_table, w.x01, c.nS, s->demb_recv, w.pts, c.nt, w.dfeat,
                                                 b->d_hash, w.scat_ws, w.scat_bytes, stream, ow_level < c.g.k ? ow_level : RFX_MAX_LEVELS + 1));
    }
    if (b->d_poses16)
        RFX_TRY(rfx_grid_encode_backward(&c.g.own, b->field.hash_table, w.x01, c.nS, s->demb_recv, nullptr, s->dx_send, nullptr, 0,
                                         stream));
    return RFX_OK;
}

int rfx_ba_shard_pose(const rfx_ba_desc* b, const rfx_ba_shard* s, void* workspace, size_t workspace_bytes, rfx_stream stream) {
    ShardCall c;
    RFX_TRY(shard_call(b, s, workspace, workspace_bytes, &c));
    if (!b->d_poses16) return RFX_ERR_ARG;
    const BaWs& w = c.w;
    const int S = c.S;
    const int64_t m = c.g.n_own, mS = m * S;
    hipStream_t st = as_stream(stream);
    if (m == 0) {
        RFX_HIP_TRY(hipMemsetAsync(b->d_poses16, 0, (size_t)b->K * 16 * sizeof(float), st));
        return RFX_OK;
    }
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((mS * 3 + 255) / 256)), dim3(256), 0, st, s->dx_recv, s->world, mS * 3, mS * 3,
                       w.dx);
    RFX_LAUNCH_CHECK();
    RFX_TRY(rfx_field_backward_dx(&b->field, w.x01 + c.g.r0 * S * 3, mS, w.d_raw, w.dx, w.bwd_ws, w.bwd_bytes, stream));
    const float ex = (float)(b->bbox[1] - b->bbox[0]), ey = (float)(b->bbox[3] - b->bbox[2]), ez = (float)(b->bbox[5] - b->bbox[4]);
    hipLaunchKernelGGL(ray_grad_reduce_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, w.dx, w.z + c.g.r0 * S, m, S, ex, ey, ez,
                       w.go, w.gd);
    RFX_LAUNCH_CHECK();
    return rfx_pose_grad(w.go, w.gd, w.d_cam + c.g.r0 * 3, w.pidx + c.g.r0, m, b->K, b->d_poses16, stream);
}

}  // extern "C"
