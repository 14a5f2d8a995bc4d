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
            nb = self.lib.rfx_tsdf_integrate_workspace_bytes(H, W)
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


def test_integrate_no_valid_depth_is_noop_and_camera_outside():
    K, c2w, rgb, depth, _ = small_frame(H=60, W=80)
    _run_pair((40, 40, 30), (-2, -3, -2), 0.1, 0.2, [(K, c2w, rgb, np.zeros_like(depth))])
    far = look_at((30.0, 0.0, 0.0), (1.0, 0.0, 0.0))      # looks away from the volume
    _run_pair((40, 40, 30), (-2, -3, -2), 0.1, 0.2, [(K, far, rgb, depth)])
    back = look_at((6.0, 0.0, 0.0), (-1.0, 0.05, 0.0))    # outside, looking in
    _run_pair((40, 40, 30), (-2, -3, -2), 0.1, 0.2, [(K, back, rgb, depth + 4.0)])


def test_integrate_matches_nofma_oracle_within_tolerance():
    """the uncontracted variant of the oracle only moves round-off (SURVEY 8d tolerances)."""
    from oracle import tsdf as O
    K, c2w, rgb, depth, _ = small_frame()
    dims, origin = (200, 200, 150), (-4, -5, -3)
    a, b = _oracle_vol(dims), _oracle_vol(dims)
    O.load(True).mv_integrate(*a, dims, origin, 0.04, K, c2w, O.pack_color(rgb), depth, 0.15)
    O.load(False).mv_integrate(*b, dims, origin, 0.04, K, c2w, O.pack_color(rgb), depth, 0.15)
    touched = (a[1] != b[1])
    assert touched.mean() <= 1e-4
    same = ~touched
    assert np.max(np.abs(a[0][same] - b[0][same])) <= 2e-5
