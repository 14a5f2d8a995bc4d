"""GPU parity: librfx TSDF / GBV kernels vs the C oracle, bit for bit (through the C ABI)."""
import numpy as np
import pytest

from conftest import look_at, small_frame

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _assert_bit_equal(got, ref, what):
    g, r = _bits(got), _bits(ref)
    if not np.array_equal(g, r):
        bad = np.flatnonzero(g != r)
        raise AssertionError(f"{what}: {bad.size} of {g.size} differ; first idx {bad[:5]}, "
                             f"got {np.asarray(got).ravel()[bad[:5]]}, ref {np.asarray(ref).ravel()[bad[:5]]}")


class Vol:
    """thin ctypes driver over the C ABI (mirrors what moving_volume does)."""

    def __init__(self, dims, origin, voxel, trunc, weight_clamp=1, decode=0):
        import torch
        from remixfusion_amd import _lib
        self.torch, self.L, self.lib = torch, _lib, _lib.load()
        self.dims, self.origin, self.voxel, self.trunc = tuple(int(d) for d in dims), np.asarray(origin, np.float32), float(voxel), float(trunc)
        n = int(np.prod(self.dims))
        dev = "cuda:0"
        self.t = torch.ones(n, device=dev)
        self.w = torch.zeros(n, device=dev)
        self.c = torch.zeros(n, device=dev)
        self.weight_clamp, self.decode = weight_clamp, decode
        self.ws = None

    def integrate(self, K, c2w, rgb255, depth, obs_weight=1.0, reintegrate=0, old_bnd=None):
        torch, L = self.torch, self.L
        H, W = depth.shape
        if self.ws is None:
            nb = self.lib.rfx_tsdf_integrate_workspace_bytes(*self.dims, H, W)
            self.ws = torch.empty((nb + 3) // 4, device="cuda:0")
            self.cpk = torch.empty(H * W, device="cuda:0")
        d_rgb = torch.from_numpy(np.ascontiguousarray(rgb255, np.float32)).cuda().reshape(-1, 3)
        d_dep = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).cuda().reshape(-1)
        st = L.stream_ptr()
        L.check(self.lib.rfx_pack_color(L.ptr(d_rgb), L.ptr(self.cpk), H * W, st), "pack")
        ob = np.zeros(6, np.float32) if old_bnd is None else np.asarray(old_bnd, np.float32).reshape(-1)
        L.check(self.lib.rfx_tsdf_integrate(L.ptr(self.t), L.ptr(self.w), L.ptr(self.c), *self.dims,
                                            L.farr(L._F3, self.origin), self.voxel, L.farr(L._F9, K.reshape(-1)),
                                            L.farr(L._F16, c2w.reshape(-1)), L.ptr(self.cpk), L.ptr(d_dep), H, W,
                                            self.trunc, float(obs_weight), int(self.weight_clamp), int(reintegrate),
                                            L.farr(L._F6, ob), self.decode, L.ptr(self.ws), self.ws.numel() * 4, st),
                "integrate")
        torch.cuda.synchronize()

    def host(self):
        return self.t.cpu().numpy(), self.w.cpu().numpy(), self.c.cpu().numpy()


def _oracle_vol(dims):
    n = int(np.prod(dims))
    return np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)


def _run_pair(dims, origin, voxel, trunc, frames, weight_clamp=1, decode="reference", fma=True, **kw):
    from oracle import tsdf as O
    orc = O.load(fma)
    v = Vol(dims, origin, voxel, trunc, weight_clamp, 0 if decode == "reference" else 1)
    ot, ow, oc = _oracle_vol(dims)
    counts = []
    for (K, c2w, rgb255, depth) in frames:
        v.integrate(K, c2w, rgb255, depth, **kw)
        counts.append(orc.mv_integrate(ot, ow, oc, dims, origin, voxel, K, c2w, O.pack_color(rgb255), depth, trunc,
                                       obs_weight=kw.get("obs_weight", 1.0), weight_clamp=float(weight_clamp),
                                       reintegrate=float(kw.get("reintegrate", 0)), old_bnd=kw.get("old_bnd"),
                                       decode=decode))
    gt, gw, gc = v.host()
    _assert_bit_equal(gw, ow, "weight")
    _assert_bit_equal(gt, ot, "tsdf")
    _assert_bit_equal(gc, oc, "colour")
    return counts


def test_integrate_three_frames_bit_exact():
    frames = [small_frame(frame=f)[:4] for f in (0, 7, 19)]
    counts = _run_pair((200, 200, 150), (-4, -5, -3), 0.04, 0.15, frames)
    assert all(u > 10000 and c > 100 for u, c in counts)


@pytest.mark.parametrize("dims", [(37, 41, 53), (64, 64, 64), (5, 300, 7), (130, 3, 129)])
def test_integrate_ragged_dims(dims):
    K, c2w, rgb, depth, _ = small_frame(H=60, W=80)
    c2w = look_at((0.1, -0.2, 0.05), (1.0, 0.3, -0.1))
    _run_pair(dims, (-1, -2, -1), 0.05, 0.2, [(K, c2w, rgb, depth)])


def test_integrate_above_2pow24_reference_vs_exact_decode():
    # 22.5e6 voxels: (float)idx is inexact, the literal decode aliases a few voxels
    K, c2w, rgb, depth, _ = small_frame()
    dims = (300, 300, 250)
    for decode in ("reference", "exact"):
        _run_pair(dims, (-3, -4, -2), 0.02, 0.06, [(K, c2w, rgb, depth)], decode=decode)


def test_integrate_every_voxel_in_the_truncation_band_and_tiny_grids():
    """round 5's chunk kernel keeps the exact-path voxels of a wave in a 192-record LDS list and evaluates them after the item
    loop; with a truncation band as wide as the room nearly EVERY lane is such a voxel, so a list fills within three items and the
    in-loop drain (list about to overflow) runs all the time, over several frames (stored values re-read by later rounds).  The
    small volumes give queues of a few items on grids of fewer than eight workgroups (the XCD dealing's degenerate cases)."""
    frames = [small_frame(H=120, W=160, frame=f)[:4] for f in (0, 5, 11)]
    _run_pair((200, 200, 150), (-4, -5, -3), 0.04, 3.0, frames)             # trunc 3 m: the band covers the whole frustum
    K, c2w, rgb, depth, _ = small_frame(H=60, W=80)
    for dims, voxel in (((6, 5, 70), 0.5), ((3, 3, 3), 1.0), ((9, 9, 130), 0.1), ((16, 12, 64), 0.3)):
        _run_pair(dims, (-2, -2, -2), voxel, 0.6, [(K, c2w, rgb, depth)] * 2)


def test_integrate_weight_clamp_saturates_at_40():
    K, c2w, rgb, depth, _ = small_frame(H=60, W=80)
    _run_pair((80, 80, 60), (-2, -3, -2), 0.05, 0.2, [(K, c2w, rgb, depth)] * 43, weight_clamp=1)
    _run_pair((40, 40, 30), (-2, -3, -2), 0.1, 0.2, [(K, c2w, rgb, depth)] * 3, weight_clamp=0)


def test_integrate_reintegrate_and_deintegrate():
    K, c2w, rgb, depth, _ = small_frame(H=60, W=80)
    dims, origin = (80, 80, 60), (-2, -3, -2)
    old = np.array([-1.0, 1.5, -2.0, 0.5, -1.0, 1.0], np.float32)
    _run_pair(dims, origin, 0.05, 0.2, [(K, c2w, rgb, depth)], reintegrate=1, old_bnd=old)
    # integrate once, then remove the observation again inside old_bnd (obs_weight = -1 resets w<=1 voxels)
    from oracle import tsdf as O
    orc = O.load(True)
    v = Vol(dims, origin, 0.05, 0.2)
    ot, ow, oc = _oracle_vol(dims)
    for obs, re in ((1.0, 0), (-1.0, 1)):
        v.integrate(K, c2w, rgb, depth, obs_weight=obs, reintegrate=re, old_bnd=old)
        orc.mv_integrate(ot, ow, oc, dims, origin, 0.05, K, c2w, O.pack_color(rgb), depth, 0.2, obs_weight=obs,
                         reintegrate=float(re), old_bnd=old)
    gt, gw, gc = v.host()
    # w_old + (-1) == 0 outside the reset region gives 0/0 = NaN in both; compare bit patterns
    _assert_bit_equal(gw, ow, "weight")
    _assert_bit_equal(gt, ot, "tsdf")
    _assert_bit_equal(gc, oc, "colour")


@pytest.mark.parametrize("dims,cuts,decode", [((200, 200, 150), (0, 37, 100, 101, 200), "reference"), ((300, 300, 250), (0, 150, 300), "reference"),
                                              ((300, 300, 250), (0, 75, 150, 225, 300), "exact"), ((64, 48, 40), (0, 1, 63, 64), "reference")])
def test_integrate_slabs_equal_the_whole_volume(dims, cuts, decode):
    """multi-GPU form (rfx_tsdf_integrate_slab): the volume cut into x-slabs, each slab integrating the same frames into
    its own arrays, is bit-identical to the whole volume -- including above 2^24 voxels, where the reference's fp32 index
    decode aliases voxels next to the x-slab boundaries of the GLOBAL index (300x300x250: 22.5e6 voxels)."""
    import torch
    from remixfusion_amd import _lib as L
    lib = L.load()
    frames = [small_frame(frame=f)[:4] for f in (0, 11)]
    origin, voxel, trunc = (-3, -4, -2), 0.02 if dims[0] == 300 else 0.04, 0.06 if dims[0] == 300 else 0.15
    if dims[0] == 64:
        origin, voxel, trunc = (-2, -3, -2), 0.08, 0.24
    dec = 0 if decode == "reference" else 1
    whole = Vol(dims, origin, voxel, trunc, 1, dec)
    for (K, c2w, rgb, depth) in frames:
        whole.integrate(K, c2w, rgb, depth)
    wt, ww, wc = whole.host()
    H, W = frames[0][3].shape
    plane = dims[1] * dims[2]
    st = L.stream_ptr()
    for x0, x1 in zip(cuts[:-1], cuts[1:]):
        n = (x1 - x0) * plane
        t, w, c = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        nb = lib.rfx_tsdf_integrate_workspace_bytes(x1 - x0, dims[1], dims[2], H, W)
        ws, cpk = torch.empty((nb + 3) // 4, device="cuda"), torch.empty(H * W, device="cuda")
        for (K, c2w, rgb, depth) in frames:
            d_rgb = torch.from_numpy(np.ascontiguousarray(rgb, np.float32)).cuda().reshape(-1, 3)
            d_dep = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).cuda().reshape(-1)
            L.check(lib.rfx_pack_color(L.ptr(d_rgb), L.ptr(cpk), H * W, st), "pack")
            L.check(lib.rfx_tsdf_integrate_slab(L.ptr(t), L.ptr(w), L.ptr(c), *dims, x0, x1, L.farr(L._F3, np.asarray(origin, np.float32)),
                                                voxel, L.farr(L._F9, K.reshape(-1)), L.farr(L._F16, c2w.reshape(-1)), L.ptr(cpk), L.ptr(d_dep),
                                                H, W, trunc, 1.0, 1, 0, L.farr(L._F6, np.zeros(6, np.float32)), dec, L.ptr(ws),
                                                ws.numel() * 4, st), "slab")
        torch.cuda.synchronize()
        sl = slice(x0 * plane, x1 * plane)
        _assert_bit_equal(w.cpu().numpy(), ww[sl], f"weight slab [{x0},{x1})")
        _assert_bit_equal(t.cpu().numpy(), wt[sl], f"tsdf slab [{x0},{x1})")
        _assert_bit_equal(c.cpu().numpy(), wc[sl], f"colour slab [{x0},{x1})")
    assert float((ww > 0).sum()) > 1000
    # argument checks
    t = torch.ones(plane, device="cuda")
    rc = lib.rfx_tsdf_integrate_slab(L.ptr(t), L.ptr(t), L.ptr(t), *dims, 5, 5, L.farr(L._F3, np.asarray(origin, np.float32)), voxel,
                                     L.farr(L._F9, K.reshape(-1)), L.farr(L._F16, c2w.reshape(-1)), L.ptr(cpk), L.ptr(d_dep), H, W, trunc, 1.0,
                                     1, 0, L.farr(L._F6, np.zeros(6, np.float32)), dec, L.ptr(ws), ws.numel() * 4, st)
    assert rc != 0


def test_integrate_no_valid_depth_is_noop_and_camera_outside():
    K, c2w, rgb, depth, _ = small_frame(H=60, W=80)
    _run_pair((40, 40, 30), (-2, -3, -2), 0.1, 0.2, [(K, c2w, rgb, np.zeros_like(depth))])
    far = look_at((30.0, 0.0, 0.0), (1.0, 0.0, 0.0))      # looks away from the volume
    _run_pair((40, 40, 30), (-2, -3, -2), 0.1, 0.2, [(K, far, rgb, depth)])
    back = look_at((6.0, 0.0, 0.0), (-1.0, 0.05, 0.0))    # outside, looking in
    _run_pair((40, 40, 30), (-2, -3, -2), 0.1, 0.2, [(K, back, rgb, depth + 4.0)])


# ------------------------------------------------------------------ the other MV kernels
def _rand_vol(dims, seed=0):
    rng = np.random.default_rng(seed)
    n = int(np.prod(dims))
    t = rng.uniform(-1, 1, n).astype(np.float32)
    w = rng.integers(0, 5, n).astype(np.float32)
    c = (rng.integers(0, 256, n) * 65536 + rng.integers(0, 256, n) * 256 + rng.integers(0, 256, n)).astype(np.float32)
    return t, w, c


def _cuda(*arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


def test_fill_copy_filter():
    import torch
    from oracle import tsdf as O
    from remixfusion_amd import _lib as L
    lib, orc = L.load(), O.load()
    for n_side in ((13, 7, 5), (64, 64, 33)):
        t, w, c = _rand_vol(n_side, 1)
        n = t.size
        dt, dw, dc = _cuda(t, w, c)
        bt, bw, bc = (torch.empty_like(dt) for _ in range(3))
        L.check(lib.rfx_tsdf_copy(L.ptr(dt), L.ptr(dw), L.ptr(dc), L.ptr(bt), L.ptr(bw), L.ptr(bc), n, L.stream_ptr()), "copy")
        for g, r in zip((bt, bw, bc), (t, w, c)):
            _assert_bit_equal(g.cpu().numpy(), r, "copy")
        ft, fw, fc = t.copy(), w.copy(), c.copy()
        orc.mv_filter(ft, fw, fc, 2.0)
        L.check(lib.rfx_tsdf_filter(L.ptr(dt), L.ptr(dw), L.ptr(dc), n, 2.0, L.stream_ptr()), "filter")
        for g, r in zip((dt, dw, dc), (ft, fw, fc)):
            _assert_bit_equal(g.cpu().numpy(), r, "filter")
        L.check(lib.rfx_tsdf_fill(L.ptr(dt), L.ptr(dw), L.ptr(dc), n, L.stream_ptr()), "fill")
        assert (dt == 1).all() and (dw == 0).all() and (dc == 0).all()


@pytest.mark.parametrize("shift", [(1.0, 0.0, 0.0), (-1.0, 2.0, 0.0), (0.0, 0.0, -1.0), (3.0, -3.0, 1.0), (0.37, 0.0, 0.0)])
def test_shift_matches_oracle_and_roundtrips(shift):
    from oracle import tsdf as O
    from remixfusion_amd import _lib as L
    lib, orc = L.load(), O.load()
    dims, voxel = (50, 40, 30), 0.1
    origin = np.array([-2.0, -2.0, -1.0], np.float32)
    new_origin = origin + np.asarray(shift, np.float32)
    src = _rand_vol(dims, 2)
    ref = [np.empty_like(a) for a in src]
    orc.mv_shift(ref, src, dims, new_origin, dims, origin, voxel)
    d_src = _cuda(*src)
    d_dst = _cuda(*[np.zeros_like(a) for a in src])
    L.check(lib.rfx_tsdf_shift(*[L.ptr(a) for a in d_dst], *dims, L.farr(L._F3, new_origin),
                               *[L.ptr(a) for a in d_src], *dims, L.farr(L._F3, origin), voxel, 0, L.stream_ptr()), "shift")
    for g, r, nm in zip(d_dst, ref, "twc"):
        _assert_bit_equal(g.cpu().numpy(), r, "shift " + nm)
    if all(float(s).is_integer() for s in shift):
        # shift(+s) then shift(-s) restores the overlap region exactly, (1,0,0) elsewhere
        back = _cuda(*[np.zeros_like(a) for a in src])
        L.check(lib.rfx_tsdf_shift(*[L.ptr(a) for a in back], *dims, L.farr(L._F3, origin),
                                   *[L.ptr(a) for a in d_dst], *dims, L.farr(L._F3, new_origin), voxel, 0, L.stream_ptr()), "shift")
        sv = np.round(np.asarray(shift) / voxel).astype(int)
        x, y, z = np.meshgrid(*[np.arange(d) for d in dims], indexing="ij")
        inside = np.ones(dims, bool)
        for ax, g in zip(range(3), (x, y, z)):
            inside &= (g - sv[ax] >= 0) & (g - sv[ax] < dims[ax])
        got = back[0].cpu().numpy().reshape(dims)
        assert np.array_equal(got[inside], src[0].reshape(dims)[inside])
        assert (got[~inside] == 1).all()


def test_trilerp_matches_oracle():
    from oracle import tsdf as O
    from remixfusion_amd import _lib as L
    import torch
    lib, orc = L.load(), O.load()
    dims, voxel = (30, 25, 20), 0.05
    origin = np.array([-0.7, -0.6, -0.5], np.float32)
    t, w, c = _rand_vol(dims, 3)
    rng = np.random.default_rng(4)
    pts = rng.uniform(-0.9, 1.0, (5000, 3)).astype(np.float32)         # includes out-of-volume points
    pts[:50] = origin + np.floor(rng.uniform(0, 19, (50, 3))) * voxel  # exactly on vertices
    ref = orc.mv_trilerp(t, w, c, dims, origin, voxel, pts)
    dt, dw, dc, dp = _cuda(t, w, c, pts)
    out = torch.empty((pts.shape[0], 5), device="cuda")
    L.check(lib.rfx_tsdf_trilerp(L.ptr(dt), L.ptr(dw), L.ptr(dc), *dims, L.farr(L._F3, origin), voxel, L.ptr(dp),
                                 pts.shape[0], L.ptr(out), L.stream_ptr()), "trilerp")
    _assert_bit_equal(out.cpu().numpy(), ref, "trilerp")
    # empty input is a no-op
    L.check(lib.rfx_tsdf_trilerp(L.ptr(dt), L.ptr(dw), L.ptr(dc), *dims, L.farr(L._F3, origin), voxel, L.ptr(dp), 0,
                                 L.ptr(out), L.stream_ptr()), "trilerp0")


def test_truncated_pc_matches_oracle():
    from oracle import tsdf as O
    from remixfusion_amd import _lib as L
    import torch
    lib, orc = L.load(), O.load()
    dims, voxel = (40, 30, 20), 0.05
    origin = np.array([-1.0, -0.7, -0.5], np.float32)
    t, w, c = _rand_vol(dims, 5)
    for pc_num in (1000, 24000, 50000):
        ref, n_ref = orc.mv_truncated_pc(t, c, dims, origin, voxel, 0.05, pc_num, 0.5)
        dt, dc = _cuda(t, c)
        pc = torch.zeros((pc_num, 7), device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        L.check(lib.rfx_tsdf_truncated_pc(L.ptr(dt), L.ptr(dc), *dims, L.farr(L._F3, origin), voxel, 0.05, pc_num, 0.5,
                                          L.ptr(pc), cnt.data_ptr(), 0, L.stream_ptr()), "pc")
        _assert_bit_equal(pc.cpu().numpy(), ref, "pc")
        assert int(cnt.item()) == n_ref


# ------------------------------------------------------------------ GBV
def test_gbv_integrate_and_clear_bit_exact():
    from oracle import tsdf as O
    from remixfusion_amd import _lib as L
    import torch
    lib, orc = L.load(), O.load()
    box = np.array([-3, 3, -4, 2.5, -2, 2.5], np.float32)
    for R in (48, 200):
        n = R ** 3
        trgb = np.zeros((n, 4), np.float32)
        wv = np.zeros(n, np.float32)
        orc.gbv_clear(trgb)
        d_trgb = torch.zeros((n, 4), device="cuda")
        d_w = torch.zeros(n, device="cuda")
        L.check(lib.rfx_gbv_clear(L.ptr(d_trgb), n, L.stream_ptr()), "clear")
        _assert_bit_equal(d_trgb.cpu().numpy(), trgb, "clear")
        for f in (0, 5, 10):
            K, c2w, _, depth, rgb01 = small_frame(frame=f)
            upd = orc.gbv_integrate(trgb, wv, R, box, K, c2w, rgb01, depth, 0.1)
            assert upd > 1000
            d_pose, d_rgb, d_dep = _cuda(c2w.reshape(-1), rgb01, depth)
            L.check(lib.rfx_gbv_integrate(L.ptr(d_trgb), L.ptr(d_w), R, L.farr(L._F6, box), L.farr(L._F9, K.reshape(-1)),
                                          L.ptr(d_pose), L.ptr(d_rgb), L.ptr(d_dep), depth.shape[0], depth.shape[1],
                                          0.1, 1.0, L.stream_ptr()), "gbv")
        _assert_bit_equal(d_w.cpu().numpy(), wv, "gbw")
        _assert_bit_equal(d_trgb.cpu().numpy(), trgb, "gbv")
        # de-integration of the last frame (obs_weight = -1) follows the same branch structure
        orc.gbv_integrate(trgb, wv, R, box, K, c2w, rgb01, depth, 0.1, obs_weight=-1.0)
        L.check(lib.rfx_gbv_integrate(L.ptr(d_trgb), L.ptr(d_w), R, L.farr(L._F6, box), L.farr(L._F9, K.reshape(-1)),
                                      L.ptr(d_pose), L.ptr(d_rgb), L.ptr(d_dep), depth.shape[0], depth.shape[1],
                                      0.1, -1.0, L.stream_ptr()), "gbv-")
        _assert_bit_equal(d_w.cpu().numpy(), wv, "gbw-")
        _assert_bit_equal(d_trgb.cpu().numpy(), trgb, "gbv-")


def test_moving_volume_class_follows_camera():
    """moving_volume through its reference-shaped API: integrate, move > t_treshold, swap."""
    import torch
    from oracle import tsdf as O
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.Volume import moving_volume

    class T:  # the four anchor fields of model/traj.py:28-31
        kfx = kfy = kfz = 0.0
        first = 0

    cfg = synthetic_config("office0")
    cfg["volume"].update({"voxel_size": 0.05, "trunc": 0.15})
    K, c2w, rgb255, depth, _ = small_frame()
    traj = T()
    mv = moving_volume(cfg, traj, c2w.astype(np.float64))
    assert tuple(mv.vol_dim) == (160, 160, 120)
    orc = O.load()
    ot, ow, oc = _oracle_vol(mv.vol_dim)
    mv.integrate(rgb255, depth, K, c2w, mv.vol_bnds)
    orc.mv_integrate(ot, ow, oc, mv.vol_dim, mv.vol_origin, 0.05, K, c2w, O.pack_color(rgb255), depth, 0.15)
    moved = c2w.copy().astype(np.float64)
    moved[0, 3] += 1.3                                      # > t_treshold along x
    flag, old_bnds = mv.check_move_volume_new(1, moved, traj)
    assert flag and mv.vol_bnds[0, 0] == old_bnds[0, 0] + 1.0
    nt, nw, nc = _oracle_vol(mv.vol_dim)
    orc.mv_shift((nt, nw, nc), (ot, ow, oc), mv.vol_dim, mv.vol_origin, mv.vol_dim, old_bnds[:, 0].astype(np.float32), 0.05)
    gt, gw, gc = mv.get_volume_all()
    _assert_bit_equal(gt, nt, "mv tsdf")
    _assert_bit_equal(gw, nw, "mv weight")
    _assert_bit_equal(gc, nc, "mv colour")
    res, mask = mv.tri_interpolate(np.array([[0.5, 0.2, 0.1], [100.0, 0, 0]], np.float32))
    assert res.shape == (2, 5) and mask.all() and res[1, 0] == 1.0


@pytest.mark.timeout(900)
def test_integrate_full_size_800x800x600_bit_exact():
    """BASELINE config 2 size (3.84e8 voxels @ 1 cm, 640x480): GPU vs C oracle, every voxel."""
    import torch
    from oracle import tsdf as O
    dims, origin, voxel, trunc = (800, 800, 600), (-4, -5, -3), 0.01, 0.05
    K, c2w, rgb255, depth, _ = small_frame(H=480, W=640, frame=3)
    v = Vol(dims, origin, voxel, trunc)
    v.integrate(K, c2w, rgb255, depth)
    ot, ow, oc = _oracle_vol(dims)
    u, c = O.load().mv_integrate(ot, ow, oc, dims, origin, voxel, K, c2w, O.pack_color(rgb255), depth, trunc)
    assert u > 1e6
    for g, r, nm in ((v.w, ow, "weight"), (v.t, ot, "tsdf"), (v.c, oc, "colour")):
        _assert_bit_equal(g.cpu().numpy(), r, nm)


def test_pipeline_volume_stream_gives_the_same_volume():
    """MappingPipeline integrates the moving volume on its own HIP stream, overlapping the mapper (pipeline.mv_stream);
    the volume after a stretch of frames must be bit-identical to the one-stream run."""
    import torch
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline

    def run(side_stream):
        cfg = synthetic_config("office0")
        cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
        cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
        cfg["mapping"].update({"first_iters": 5, "sample": 512, "iters": 2, "BA_iters": 2})
        cfg["training"].update({"smooth_pts": 16})
        cfg["pipeline"] = {"mv_stream": side_stream}
        pipe = MappingPipeline(cfg, n_frames=40, seed=1)
        assert (pipe.mv_stream is not None) == side_stream
        frames = pipe.prefetch(list(range(31)))
        pipe.start(frames[0])
        for i in range(1, 31):
            pipe.step(i, frames[i])
        # no synchronize: get_volume_all itself waits for the stream that integrates (moving_volume.producer_stream)
        return [v.copy() for v in pipe.mv.get_volume_all()], np.array(pipe.mv.vol_bnds, copy=True)

    (t0, w0, c0), b0 = run(False)
    (t1, w1, c1), b1 = run(True)
    assert np.array_equal(b0, b1)
    assert float((w0 > 0).mean()) > 0.01
    _assert_bit_equal(t1, t0, "tsdf")
    _assert_bit_equal(w1, w0, "weight")
    _assert_bit_equal(c1, c0, "colour")
