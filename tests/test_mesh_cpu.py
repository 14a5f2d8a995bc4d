"""Marching-cubes case tables (remixfusion_amd/mesh.py) checked on the CPU through oracle/mesh_oracle.py."""
import numpy as np

from oracle import mesh_oracle as mo
from remixfusion_amd.mesh import EDGE_CORNERS, build_tables


def test_tables_shape_and_complement():
    n_tri, tab, max_tri = build_tables()
    assert max_tri == 5 and n_tri[0] == 0 and n_tri[255] == 0
    assert (n_tri[1:255] > 0).all()
    # every listed edge of a case is a cut edge of that case
    for case in range(256):
        for e in tab[case, :3 * n_tri[case]]:
            a, b = EDGE_CORNERS[e]
            assert ((case >> a) & 1) != ((case >> b) & 1)


def test_sphere_closed_oriented():
    n_tri, tab, _ = build_tables()
    N = 14
    g = np.stack(np.meshgrid(*[np.arange(N)] * 3, indexing="ij"), -1).astype(np.float64)
    r = 4.3
    f = np.linalg.norm(g - (N - 1) / 2, axis=-1) - r
    v, k = mo.polygonise(f, 0.0, None, n_tri, tab, EDGE_CORNERS)
    bad, chi, vol = mo.topology(*mo.weld(v, k))
    assert bad == 0 and chi == 2
    assert 0.93 < vol / (4 / 3 * np.pi * r ** 3) < 1.0        # inscribed polyhedron, outward orientation


def test_noise_closed_oriented():
    """random fields hit the ambiguous cases; positive padding closes the surface inside the volume."""
    n_tri, tab, _ = build_tables()
    rng = np.random.default_rng(3)
    for _ in range(2):
        pad = np.ones((11, 11, 11))
        pad[1:-1, 1:-1, 1:-1] = rng.standard_normal((9, 9, 9))
        v, k = mo.polygonise(pad, 0.0, None, n_tri, tab, EDGE_CORNERS)
        bad, chi, vol = mo.topology(*mo.weld(v, k))
        assert bad == 0 and vol > 0
