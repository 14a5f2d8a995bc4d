"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the marching-cubes sweep.

The reference meshes with ``skimage.measure.marching_cubes`` (utils.py:158), which is not vendored in
/root/reference and not installed here: parity with skimage's Lewiner tables is UNPINNED.  What this
oracle pins instead: given the same case tables, the device sweep must emit exactly these triangles, and
the tables themselves must give closed, consistently oriented surfaces (``topology``).
Pure-Python loops: small volumes only.
"""
import numpy as np

CORNERS = np.array([[(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], np.int64)


def polygonise(field, level, mask, n_tri, tab, edge_corners):
    """-> (tri_verts [3T,3] float32 in index coords, keys [3T] int64, cell-major / table order)."""
    field = np.asarray(field, np.float32)
    X, Y, Z = field.shape
    level = np.float32(level)
    V, K = [], []
    for x in range(X - 1):
        for y in range(Y - 1):
            for z in range(Z - 1):
                vals, code, ok = [], 0, True
                for c in range(8):
                    i = (x + CORNERS[c, 0], y + CORNERS[c, 1], z + CORNERS[c, 2])
                    v = field[i]
                    vals.append(v)
                    ok = ok and not np.isnan(v) and (mask is None or bool(mask[i]))
                    if v < level:
                        code |= 1 << c
                if not ok:
                    continue
                for k in range(3 * int(n_tri[code])):
                    a, b = edge_corners[tab[code, k]]
                    va, vb = vals[a], vals[b]
                    w = np.float32(level - va) / np.float32(vb - va)
                    pa = (np.array([x, y, z]) + CORNERS[a]).astype(np.float32)
                    axis = int(np.argmax(CORNERS[b] - CORNERS[a]))
                    pa[axis] = pa[axis] + w
                    V.append(pa)
                    ga = np.array([x, y, z]) + CORNERS[a]
                    K.append(((int(ga[0]) * Y + int(ga[1])) * Z + int(ga[2])) * 3 + axis)
    return np.array(V, np.float32).reshape(-1, 3), np.array(K, np.int64)


def weld(tri_verts, keys):
    u, inv = np.unique(keys, return_inverse=True)
    verts = np.zeros((len(u), 3), np.float64)
    verts[inv] = tri_verts
    return verts, inv.reshape(-1, 3)


def topology(verts, faces):
    """-> (number of directed edges not matched by exactly one opposite edge, Euler characteristic,
    signed volume).  A closed, consistently oriented 2-manifold has 0 bad edges."""
    E = {}
    for f in faces:
        for i in range(3):
            a, b = int(f[i]), int(f[(i + 1) % 3])
            E[(a, b)] = E.get((a, b), 0) + 1
    bad = sum(1 for k, v in E.items() if not (v == 1 and E.get((k[1], k[0]), 0) == 1))
    chi = len(verts) - len(E) // 2 + len(faces)
    vol = np.sum(np.einsum("ij,ij->i", verts[faces[:, 0]], np.cross(verts[faces[:, 1]], verts[faces[:, 2]]))) / 6
    return bad, chi, float(vol)
