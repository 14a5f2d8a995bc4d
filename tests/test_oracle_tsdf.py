"""CPU tests pinning the C oracle with analytic known-answer cases (the reference ships no
vectors for its PyCUDA kernels: SURVEY.md 8c)."""
import numpy as np

from conftest import look_at, small_frame
from oracle import tsdf as O


def _vol(dims):
    n = int(np.prod(dims))
    return np.ones(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)


def _plane_frame(H, W, d0, rgb=(10, 20, 30)):
    K = np.array([[0.9 * W, 0, (W - 1) / 2], [0, 0.9 * W, (H - 1) / 2], [0, 0, 1]], np.float32)
    depth = np.full((H, W), d0, np.float32)
    col = np.broadcast_to(np.asarray(rgb, np.float32), (H, W, 3)).copy()
    return K, depth, col


def test_kat_fronto_parallel_plane_closed_form():
    """identity pose, plane at z-depth d0: on the optical axis sdf = d0 - z, tsdf = min(1, sdf/trunc)."""
    H, W, d0, trunc, voxel = 60, 80, 1.0, 0.1, 0.02
    K, depth, col = _plane_frame(H, W, d0)
    K[0, 2], K[1, 2] = 40.0, 30.0
    dims, origin = (20, 20, 80), (0.0, 0.0, 0.0)     # x,y >= 0, z up to 1.6 m; camera at origin looking +z
    t, w, c = _vol(dims)
    c2w = np.eye(4, dtype=np.float32)
    u, cb = O.load().mv_integrate(t, w, c, dims, origin, voxel, K, c2w, O.pack_color(col), depth, trunc)
    T = t.reshape(dims); Wt = w.reshape(dims); C = c.reshape(dims)
    z = np.arange(80) * voxel
    axis_t, axis_w = T[0, 0, :], Wt[0, 0, :]
    exp = np.minimum(1.0, (d0 - z) / trunc)
    seen = (z > 0) & (d0 - z >= -trunc + 1e-4)
    assert np.allclose(axis_t[seen], exp[seen], atol=2e-6)
    unseen = (z == 0) | (d0 - z < -trunc - 1e-4)
    assert (axis_w[seen] == 1).all() and (axis_w[unseen] == 0).all() and (axis_t[unseen] == 1).all()
    edge = np.abs(np.abs(d0 - z) - trunc) < 1e-4       # exactly on the band edge: fp32 decides
    band = seen & (np.abs(d0 - z) <= trunc) & ~edge
    assert (C[0, 0, band] == 30 * 65536 + 20 * 256 + 10).all() and (C[0, 0, seen & ~band & ~edge] == 0).all()
    assert u == int((Wt > 0).sum()) and cb == int((C > 0).sum())


def test_kat_running_average_weight_clamp_and_colour_rounding():
    H, W, voxel, trunc = 30, 40, 0.05, 0.2
    dims, origin = (8, 8, 40), (0, 0, 0)
    c2w = np.eye(4, dtype=np.float32)
    t, w, c = _vol(dims)
    orc = O.load()
    K, d1, col1 = _plane_frame(H, W, 1.00, (0, 0, 100))
    _, d2, col2 = _plane_frame(H, W, 1.10, (0, 0, 101))
    K[0, 2], K[1, 2] = 20.0, 15.0
    orc.mv_integrate(t, w, c, dims, origin, voxel, K, c2w, O.pack_color(col1), d1, trunc)
    orc.mv_integrate(t, w, c, dims, origin, voxel, K, c2w, O.pack_color(col2), d2, trunc)
    k = 19                                            # z = 0.95
    T = t.reshape(dims); Wt = w.reshape(dims); C = c.reshape(dims)
    a, b = min(1, (1.0 - 0.95) / trunc), min(1, (1.10 - 0.95) / trunc)
    assert abs(T[0, 0, k] - (a + b) / 2) < 1e-6 and Wt[0, 0, k] == 2
    assert C[0, 0, k] == np.float32(round((100 + 101) / 2 + 1e-9) * 65536)   # roundf(100.5) = 101 (half away)
    for _ in range(45):
        orc.mv_integrate(t, w, c, dims, origin, voxel, K, c2w, O.pack_color(col2), d2, trunc)
    assert w.max() == 40                              # fmin(w,128) then >40 -> 40 (Volume.py:302-306)
    t2, w2, c2 = _vol(dims)
    for _ in range(45):
        orc.mv_integrate(t2, w2, c2, dims, origin, voxel, K, c2w, O.pack_color(col2), d2, trunc, weight_clamp=0.0)
    assert w2.max() == 45


def test_kat_deintegration_resets_single_observations():
    H, W, voxel, trunc = 30, 40, 0.05, 0.2
    dims, origin = (8, 8, 40), (0, 0, 0)
    K, d1, col1 = _plane_frame(H, W, 1.0)
    K[0, 2], K[1, 2] = 20.0, 15.0
    c2w = np.eye(4, dtype=np.float32)
    t, w, c = _vol(dims)
    orc = O.load()
    orc.mv_integrate(t, w, c, dims, origin, voxel, K, c2w, O.pack_color(col1), d1, trunc)
    old = np.array([0, 1, 0, 1, 0, 3], np.float32)
    orc.mv_integrate(t, w, c, dims, origin, voxel, K, c2w, O.pack_color(col1), d1, trunc, obs_weight=-1.0,
                     reintegrate=1.0, old_bnd=old)
    assert (t == 1).all() and (w == 0).all() and (c == 0).all()


def test_kat_pixel_rounding_is_half_even():
    """a voxel projecting exactly onto x.5 picks the even pixel (__float2int_rn, Volume.py:261)."""
    H, W = 4, 8
    K = np.array([[2.0, 0, 0.5], [0, 2.0, 0.0], [0, 0, 1]], np.float32)   # u = 2*x/z + 0.5
    depth = np.zeros((H, W), np.float32)
    depth[0, 2] = 1.0    # pixel 2 valid, pixel 3 invalid
    dims, origin, voxel = (3, 1, 3), (0, 0, 0), 1.0
    t, w, c = _vol(dims)
    O.load().mv_integrate(t, w, c, dims, origin, voxel, K, np.eye(4, dtype=np.float32),
                          np.zeros((H, W), np.float32), depth, 5.0)
    # voxel (x=1,y=0,z=1): u = 2*1/1+0.5 = 2.5 -> rint -> 2 (even) -> valid depth -> updated
    assert w.reshape(dims)[1, 0, 1] == 1
    # voxel (x=2,z=1... ) u = 4.5 -> 4 (even) -> depth 0 -> skipped
    assert w.reshape(dims)[2, 0, 1] == 0


def test_literal_decode_aliases_only_above_2pow24_and_only_at_slab_edges():
    from ctypes import c_float
    dims = (300, 300, 250)
    K, c2w, rgb, depth, _ = small_frame()
    a, b = _vol(dims), _vol(dims)
    orc = O.load()
    orc.mv_integrate(*a, dims, (-3, -4, -2), 0.02, K, c2w, O.pack_color(rgb), depth, 0.06, decode="reference")
    orc.mv_integrate(*b, dims, (-3, -4, -2), 0.02, K, c2w, O.pack_color(rgb), depth, 0.06, decode="exact")
    diff = np.flatnonzero((a[0] != b[0]) | (a[1] != b[1]))
    r = diff % (300 * 250)
    assert ((r < 64) | (r >= 300 * 250 - 64)).all()      # only next to an x-slab boundary
    dims2 = (100, 100, 80)                               # < 2^24 voxels: decodes agree everywhere
    a, b = _vol(dims2), _vol(dims2)
    orc.mv_integrate(*a, dims2, (-3, -4, -2), 0.06, K, c2w, O.pack_color(rgb), depth, 0.2, decode="reference")
    orc.mv_integrate(*b, dims2, (-3, -4, -2), 0.06, K, c2w, O.pack_color(rgb), depth, 0.2, decode="exact")
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_fma_and_nofma_oracles_agree_within_survey_tolerance():
    """the contraction model only moves round-off: on a generic pose <=1e-5 of the voxels change by
    more than 1e-5 (pixel-rounding ties / truncation-edge flips), the rest by <=1e-5.
    (An axis-aligned camera such as frame 0 makes exact ties common, hence frame 7.)"""
    K, c2w, rgb, depth, _ = small_frame(frame=7)
    dims, origin = (200, 200, 150), (-4, -5, -3)
    a, b = _vol(dims), _vol(dims)
    O.load(True).mv_integrate(*a, dims, origin, 0.04, K, c2w, O.pack_color(rgb), depth, 0.15)
    O.load(False).mv_integrate(*b, dims, origin, 0.04, K, c2w, O.pack_color(rgb), depth, 0.15)
    bad = (np.abs(a[0] - b[0]) > 1e-5) | (a[1] != b[1])
    assert bad.mean() <= 1e-5
    assert np.abs(a[0] - b[0])[~bad].max() <= 1e-5


def test_shift_roundtrip_and_fill_copy():
    orc = O.load()
    dims, voxel = (20, 16, 12), 0.1
    rng = np.random.default_rng(0)
    src = [rng.uniform(-1, 1, int(np.prod(dims))).astype(np.float32) for _ in range(3)]
    o0 = np.array([-1, -1, -1], np.float32)
    o1 = o0 + np.array([0.5, 0.0, -0.2], np.float32)
    mid = [np.empty_like(s) for s in src]
    back = [np.empty_like(s) for s in src]
    orc.mv_shift(mid, src, dims, o1, dims, o0, voxel)
    orc.mv_shift(back, mid, dims, o0, dims, o1, voxel)
    S, B = src[0].reshape(dims), back[0].reshape(dims)
    assert np.array_equal(B[5:, :, :10], S[5:, :, :10]) and (B[:5] == 1).all() and (B[:, :, 10:] == 1).all()
    cp = [np.empty_like(s) for s in src]
    orc.mv_copy(src, cp)
    assert all(np.array_equal(x, y) for x, y in zip(src, cp))
    orc.mv_fill(*cp)
    assert (cp[0] == 1).all() and (cp[1] == 0).all() and (cp[2] == 0).all()


def test_trilerp_on_vertices_returns_voxel_values():
    orc = O.load()
    dims, voxel = (6, 5, 4), 0.5
    origin = np.zeros(3, np.float32)
    rng = np.random.default_rng(1)
    t = rng.uniform(-1, 1, 120).astype(np.float32)
    w = np.ones(120, np.float32)
    c = (rng.integers(0, 256, 120) * 65536 + rng.integers(0, 256, 120) * 256 + rng.integers(0, 256, 120)).astype(np.float32)
    pts = np.array([[1.0, 0.5, 0.5], [1.25, 0.5, 0.5], [9, 9, 9], [2.5, 2.0, 1.5]], np.float32)
    out = orc.mv_trilerp(t, w, c, dims, origin, voxel, pts)
    T = t.reshape(dims)
    assert out[0, 0] == T[2, 1, 1] and out[0, 4] == T[2, 1, 1]
    assert abs(out[1, 0] - 0.5 * (T[2, 1, 1] + T[3, 1, 1])) < 1e-6
    assert tuple(out[2]) == (1, 0, 0, 0, 0)          # outside
    assert tuple(out[3]) == (1, 0, 0, 0, 0)          # low corner on the last vertex -> out of range (:382)


def test_gbv_kat_and_clear():
    orc = O.load()
    R = 40
    box = np.array([0, 2, 0, 2, 0, 2], np.float32)
    H, W = 30, 40
    K, depth, col = _plane_frame(H, W, 1.0)
    K[0, 2], K[1, 2] = 20.0, 15.0
    rgb01 = np.full((H, W, 3), 0.25, np.float32)
    trgb = np.zeros((R ** 3, 4), np.float32)
    wv = np.zeros(R ** 3, np.float32)
    orc.gbv_clear(trgb)
    assert (trgb[:, 0] == 1).all() and (trgb[:, 1:] == 0).all()
    n = orc.gbv_integrate(trgb, wv, R, box, K, np.eye(4, dtype=np.float32), rgb01, depth, 0.1)
    G = trgb.reshape(R, R, R, 4)       # [z][y][x]
    z = np.arange(R) / R * 2.0
    k = 19                              # z = 0.95 on the axis x=y=0
    assert abs(G[k, 0, 0, 0] - 0.5) < 1e-5 and abs(G[k, 0, 0, 1] - 0.25) < 1e-7
    assert n == int((wv > 0).sum())
    assert (G[z > 1.1 + 1e-3, 0, 0, 0] == 1).all()     # behind the surface beyond trunc: untouched
