"""GPU iso-surface extraction: dense field sweep + marching cubes on the device.

Replaces the CPU path of the reference's ``utils.extract_mesh_github`` (utils.py:121-212 there: 65 536-point
chunks through ``query_sdf_res`` / ``query_w_res``, then ``skimage.measure.marching_cubes(raw, level,
mask=weight>0)`` on the host, then ``query_color_residual`` at the vertices).

The polygonisation table is *generated* here (no copied tables): for each of the 256 corner-sign cases
the cut edges are linked face by face (on an ambiguous face the two inside corners are always kept
apart, a rule that depends only on the face's own corners, so neighbouring cells agree and the surface
is crack-free), the links are traced into closed loops and each loop is fan-triangulated with outward
orientation (from the inside (< level) towards the outside).  skimage uses Lewiner's tables; individual
triangles therefore differ from the reference's mesh while the surface is the same iso-surface of the
same trilinear samples -- parity with skimage is unpinned (it is not installed here).
Cells are skipped unless all 8 corners are inside ``mask``.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

# corner id = x + 2y + 4z ; edges = corner pairs differing in exactly one bit
_CORNERS = np.array([[(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], np.int64)
_EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]      # 12 edges
# faces: (fixed axis, value) -> corners in cyclic order
_FACES = []
for axis in range(3):
    u, v = [a for a in range(3) if a != axis]
    for val in (0, 1):
        cyc = []
        for (du, dv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
            p = [0, 0, 0]
            p[axis], p[u], p[v] = val, du, dv
            cyc.append(p[0] + 2 * p[1] + 4 * p[2])
        _FACES.append(cyc)
_EDGE_ID = {e: i for i, e in enumerate(_EDGES)}


def _edge_of(a: int, b: int) -> int:
    return _EDGE_ID[(min(a, b), max(a, b))]


def _coplanar(e0: int, e1: int) -> bool:
    """True when two cube edges lie on a common cube face."""
    c4 = set(_EDGES[e0]) | set(_EDGES[e1])
    return any(c4 <= set(cyc) for cyc in _FACES)


def _triangulate(loop):
    """Triangulate a closed loop of cut edges keeping its orientation.  A diagonal between two cut edges of
    one cube face would lie *in* that face, where the neighbouring cell may draw the same segment (a
    non-manifold edge after welding), so the triangulation with the fewest such diagonals is taken
    (zero for every case of the cube)."""
    n = len(loop)
    best = None

    def rec(poly):
        # all triangulations of the index polygon `poly` -> list of (cost, [tri...])
        if len(poly) < 3:
            return [(0, [])]
        if len(poly) == 3:
            return [(0, [tuple(poly)])]
        out = []
        a, b = poly[0], poly[-1]
        for k in range(1, len(poly) - 1):
            c = poly[k]
            cost = 0
            if k > 1:
                cost += _coplanar(loop[a], loop[c])
            if k < len(poly) - 2:
                cost += _coplanar(loop[c], loop[b])
            for cl, tl in rec(poly[:k + 1]):
                for cr, tr in rec(poly[k:]):
                    out.append((cost + cl + cr, tl + [(a, c, b)] + tr))
        m = min(o[0] for o in out)
        return [o for o in out if o[0] == m][:1]

    cost, tl = rec(list(range(n)))[0]
    # (a, c, b) with a < c < b keeps the loop's cyclic order
    return [(loop[i], loop[j], loop[k]) for (i, j, k) in tl], cost


def build_tables() -> Tuple[np.ndarray, np.ndarray, int]:
    """(n_tri [256] int32, tri_edges [256, max_tri*3] int32 (edge ids, -1 padded), max_tri)."""
    cases = []
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        links: Dict[int, list] = {}
        for cyc in _FACES:
            cut = []        # (edge id, index k of the edge cyc[k]-cyc[k+1])
            for k in range(4):
                a, b = cyc[k], cyc[(k + 1) % 4]
                if inside[a] != inside[b]:
                    cut.append((_edge_of(a, b), k))
            if len(cut) == 2:
                pairs = [(cut[0][0], cut[1][0])]
            elif len(cut) == 4:
                # ambiguous face: corners alternate in/out; link the two cut edges around each inside corner
                pairs = []
                for k in range(4):
                    if inside[cyc[k]]:
                        pairs.append((_edge_of(cyc[(k - 1) % 4], cyc[k]), _edge_of(cyc[k], cyc[(k + 1) % 4])))
            else:
                pairs = []
            for e0, e1 in pairs:
                links.setdefault(e0, []).append(e1)
                links.setdefault(e1, []).append(e0)
        assert all(len(v) == 2 for v in links.values()), case
        tris = []
        seen = set()
        mid = {i: (_CORNERS[a] + _CORNERS[b]) / 2.0 for i, (a, b) in enumerate(_EDGES)}
        for start in sorted(links):
            if start in seen:
                continue
            loop, prev, cur = [start], None, start
            seen.add(start)
            while True:
                nxt = [n for n in links[cur] if n != prev]
                nxt = nxt[0] if nxt else links[cur][0]
                if nxt == start:
                    break
                if nxt in seen:      # two-edge loop cannot happen on a cube; guard anyway
                    break
                loop.append(nxt)
                seen.add(nxt)
                prev, cur = cur, nxt
            # orientation from the first link alone (a purely local rule, so neighbouring cells agree):
            # the link e0 -> e1 lies on one cube face with outward normal n_f; with t = p1 - p0 and s the
            # in-face direction from the inside to the outside corners, the loop is counter-clockwise
            # around the outward (inside -> outside) surface normal iff  s . (n_f x t) > 0.
            e0, e1 = loop[0], loop[1]
            c4 = set(_EDGES[e0]) | set(_EDGES[e1])
            face = next(cyc for cyc in _FACES if c4 <= set(cyc))
            fc = _CORNERS[face].mean(0)
            n_f = np.sign(fc - 0.5) * (np.abs(fc - 0.5) > 0.25)
            t = mid[e1] - mid[e0]
            ends = [c for e in (e0, e1) for c in _EDGES[e]]
            s_in = np.mean([_CORNERS[c] for c in ends if inside[c]], 0)
            s_out = np.mean([_CORNERS[c] for c in ends if not inside[c]], 0)
            if np.dot(s_out - s_in, np.cross(n_f, t)) < 0:
                loop = loop[::-1]
            tl, cost = _triangulate(loop)
            assert cost == 0, (case, loop)
            tris.extend(tl)
        cases.append(tris)
    max_tri = max(len(t) for t in cases)
    n_tri = np.array([len(t) for t in cases], np.int32)
    tab = -np.ones((256, max_tri * 3), np.int32)
    for c, t in enumerate(cases):
        flat = [e for tri in t for e in tri]
        tab[c, :len(flat)] = flat
    return n_tri, tab, max_tri


_TABLES: Dict[str, Tuple] = {}
EDGE_CORNERS = np.array(_EDGES, np.int32)        # [12,2]


_LATTICE_CACHE: Dict = {}


_TABLES_LOCK = None


def device_tables(device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """case tables on the device, uploaded once per device.  Exports run on several streams and threads (AsyncMeshExporter):
    the first caller records an event behind the upload and every caller's stream waits for it."""
    global _TABLES_LOCK
    if _TABLES_LOCK is None:
        import threading
        _TABLES_LOCK = threading.Lock()
    key = str(device)
    with _TABLES_LOCK:
        if key not in _TABLES:
            n_tri, tab, max_tri = build_tables()
            a, b = torch.from_numpy(n_tri).to(device), torch.from_numpy(tab).to(device)
            ready = None
            if a.is_cuda:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(a.device))
            _TABLES[key] = (a, b, max_tri, ready)
        a, b, max_tri, ready = _TABLES[key]
    if ready is not None:
        torch.cuda.current_stream(a.device).wait_event(ready)
    return a, b, max_tri


def marching_cubes(volume: torch.Tensor, level: float = 0.0, mask: Optional[torch.Tensor] = None,
                   weld: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """volume [X,Y,Z] fp32 on the GPU -> (verts [nv,3] in index coordinates, faces [nf,3] int64).
    weld=True merges the vertices shared between cells (one vertex per cut grid edge), like skimage."""
    lib = _lib.load()
    vol = volume.to(torch.float32).contiguous()
    if not vol.is_cuda:
        raise _lib.RfxError("marching_cubes needs a device tensor")
    X, Y, Z = vol.shape
    dev = vol.device
    n_tri_t, tab_t, max_tri = device_tables(dev)
    m = mask.to(torch.uint8).contiguous() if mask is not None else None
    n_cells = (X - 1) * (Y - 1) * (Z - 1)
    if n_cells <= 0:
        return torch.zeros((0, 3), device=dev), torch.zeros((0, 3), dtype=torch.int64, device=dev)
    counts = torch.empty(n_cells, dtype=torch.int32, device=dev)
    st = stream_ptr(dev)
    check(lib.rfx_mc_count(ptr(vol), m.data_ptr() if m is not None else None, X, Y, Z, float(level), n_tri_t.data_ptr(),
                           counts.data_ptr(), st), "rfx_mc_count")
    offsets = torch.cumsum(counts, 0, dtype=torch.int64)
    total = int(offsets[-1].item()) if n_cells else 0
    if total == 0:
        return torch.zeros((0, 3), device=dev), torch.zeros((0, 3), dtype=torch.int64, device=dev)
    offsets = (offsets - counts).contiguous()
    tri_verts = torch.empty((total * 3, 3), dtype=torch.float32, device=dev)
    keys = torch.empty(total * 3, dtype=torch.int64, device=dev)
    check(lib.rfx_mc_emit(ptr(vol), m.data_ptr() if m is not None else None, X, Y, Z, float(level), tab_t.data_ptr(),
                          max_tri, counts.data_ptr(), offsets.data_ptr(), ptr(tri_verts), keys.data_ptr(), st), "rfx_mc_emit")
    if not weld:
        return tri_verts, torch.arange(total * 3, device=dev).view(-1, 3)
    uniq, inverse = torch.unique(keys, return_inverse=True)
    verts = torch.empty((uniq.shape[0], 3), dtype=torch.float32, device=dev)
    verts[inverse] = tri_verts            # every duplicate holds the same interpolated position
    return verts, inverse.view(-1, 3)


def get_voxels(x_max, x_min, y_max, y_min, z_max, z_min, voxel_size=None, resolution=None):
    """grid axes of the reference's getVoxels (utils.py:79-103)."""
    x_max, x_min, y_max, y_min, z_max, z_min = (float(v) for v in (x_max, x_min, y_max, y_min, z_max, z_min))
    if voxel_size is not None:
        n = [round((a - b) / voxel_size + 0.0005) for a, b in ((x_max, x_min), (y_max, y_min), (z_max, z_min))]
        return tuple(torch.linspace(lo, hi, k + 1) for (hi, lo), k in zip(((x_max, x_min), (y_max, y_min), (z_max, z_min)), n))
    return tuple(torch.linspace(lo, hi, resolution) for hi, lo in ((x_max, x_min), (y_max, y_min), (z_max, z_min)))


@torch.no_grad()
def extract_mesh(query_fn: Callable, query_w_fn: Callable, config: Dict, bounding_box: torch.Tensor,
                 marching_cube_bound: Optional[torch.Tensor] = None, color_func: Optional[Callable] = None,
                 voxel_size: Optional[float] = None, resolution: Optional[int] = None, isolevel: float = 0.0):
    """Counterpart of ``utils.extract_mesh_github``: returns dict(vertices [nv,3] world, faces [nf,3],
    colors [nv,3] uint8 or None).  Everything stays on the device; only the vertex count is synced."""
    dev = bounding_box.device
    mcb = bounding_box if marching_cube_bound is None else marching_cube_bound
    # the sample lattice is a function of the bounds and the voxel size only: built once per (bounds, size, device) -- the
    # reference rebuilds it on the host for every export (utils.py:79-103, :137-150), 28 MB and an upload at config 5's sizes
    key = (tuple(float(v) for v in mcb.reshape(-1)), tuple(float(v) for v in bounding_box.reshape(-1)), voxel_size, resolution,
           str(dev), bool(config["grid"]["tcnn_encoding"]))
    hit = _LATTICE_CACHE.get(key)
    if hit is None:
        tx, ty, tz = get_voxels(mcb[0, 1], mcb[0, 0], mcb[1, 1], mcb[1, 0], mcb[2, 1], mcb[2, 0], voxel_size, resolution)
        pts = torch.stack(torch.meshgrid(tx, ty, tz, indexing="ij"), -1).to(torch.float32).to(dev)
        flat = pts.reshape(-1, 3)
        if config["grid"]["tcnn_encoding"]:
            flat = (flat - bounding_box[:, 0]) / (bounding_box[:, 1] - bounding_box[:, 0])
        if len(_LATTICE_CACHE) >= 4:
            # (an export on another stream may still be reading an evicted lattice: its `hit` tuple keeps the tensor alive, and
            #  record_stream below keeps the allocator from reusing the block before that stream is done with it)
            _LATTICE_CACHE.clear()
        flat = flat.contiguous()
        ready = None
        if flat.is_cuda:                       # exports run on several streams (AsyncMeshExporter): readers wait for the producer once
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(flat.device))
        hit = _LATTICE_CACHE[key] = (tx, ty, tz, tuple(pts.shape), flat, ready)
    tx, ty, tz, sh, flat, ready = hit
    if ready is not None:
        cur = torch.cuda.current_stream(flat.device)
        cur.wait_event(ready)
        flat.record_stream(cur)
    sdf = query_fn(flat[:, None, :]).reshape(sh[:-1]).to(torch.float32)
    weight = query_w_fn(flat[:, None, :]).reshape(sh[:-1])
    verts, faces = marching_cubes(sdf, isolevel, mask=weight > 0)
    # index coordinates -> world (utils.py:171-184)
    scale = torch.tensor([tx[-1] - tx[0], ty[-1] - ty[0], tz[-1] - tz[0]], device=dev)
    offset = torch.tensor([tx[0], ty[0], tz[0]], device=dev)
    denom = torch.tensor([tx.shape[0] - 1, ty.shape[0] - 1, tz.shape[0] - 1], device=dev, dtype=torch.float32)
    world = scale * (verts / denom) + offset
    world = world / config["data"]["sc_factor"] - config["data"]["translation"]
    colors = None
    if color_func is not None and world.shape[0] > 0:
        v01 = (world.to(bounding_box) - bounding_box[:, 0]) / (bounding_box[:, 1] - bounding_box[:, 0])
        rgb = color_func(v01.to(torch.float32)[:, None, :]).reshape(-1, 3)
        colors = (torch.clamp(rgb, 0, 1) * 255).to(torch.uint8)
    return {"vertices": world, "faces": faces, "colors": colors}


def write_ply(path: str, mesh: Dict) -> None:
    """minimal binary PLY writer (the reference exports through trimesh)."""
    v = mesh["vertices"].detach().cpu().numpy().astype("<f4")
    f = mesh["faces"].detach().cpu().numpy().astype("<i4")
    c = mesh["colors"].detach().cpu().numpy().astype("u1") if mesh.get("colors") is not None else None
    with open(path, "wb") as fh:
        hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {v.shape[0]}", "property float x", "property float y",
               "property float z"]
        if c is not None:
            hdr += ["property uchar red", "property uchar green", "property uchar blue"]
        hdr += [f"element face {f.shape[0]}", "property list uchar int vertex_indices", "end_header"]
        fh.write(("\n".join(hdr) + "\n").encode())
        if c is not None:
            rec = np.zeros(v.shape[0], dtype=[("p", "<f4", 3), ("c", "u1", 3)])
            rec["p"], rec["c"] = v, c
            fh.write(rec.tobytes())
        else:
            fh.write(v.tobytes())
        frec = np.zeros(f.shape[0], dtype=[("n", "u1"), ("i", "<i4", 3)])
        frec["n"], frec["i"] = 3, f
        fh.write(frec.tobytes())


# ---- in-loop exports off the mapper's critical path ------------------------------------------------------------------
class FieldSnapshot:
    """A frozen copy of what a mesh export reads -- hash table, global volume (GBV / GBW), decoder weights -- with the three
    point queries of ``JointEncoding`` on it (``query_sdf_res`` / ``query_w_res`` / ``query_color_residual``, reference
    model/scene_rep.py:212-298): same kernels, same arithmetic, other buffers.  The reference meshes inside the mapper's loop
    and blocks it (mp_slam/mapper.py:908-918); here the loop only takes the copy (D2D, ~0.1 ms for 320 MB) and a worker thread
    sweeps the copy on a stream of its own while the next frames are mapped (``AsyncMeshExporter``)."""

    def __init__(self, model):
        self.model = model
        enc = model.embed_res_fn
        self.table = torch.empty_like(enc.params.detach())
        self.gbv = torch.empty_like(model.GBV.params.detach())
        self.gbw = torch.empty_like(model.GBW.params.detach())
        self.weights = [torch.empty_like(w.detach()) for w in model.decoder_res.fused_weights()]
        self.staged = torch.empty(int(_lib.load().rfx_field_staged_floats()), dtype=torch.float32, device=self.table.device)

    def capture(self):
        """on the caller's current stream: after it, the model may go on changing"""
        m = self.model
        with torch.no_grad():
            self.table.copy_(m.embed_res_fn.params.detach())
            self.gbv.copy_(m.GBV.params.detach())
            self.gbw.copy_(m.GBW.params.detach())
            for dst, src in zip(self.weights, m.decoder_res.fused_weights()):
                dst.copy_(src.detach())

    def _desc(self):
        import ctypes as C
        m = self.model
        tr = m.config["training"]
        d = _lib.FieldDesc()
        d.hash = m.embed_res_fn.desc
        d.hash_table, d.gbv = ptr(self.table), ptr(self.gbv)
        d.gbv_res = int(m.config["globalV"]["base_resolution"])
        d.w1, d.w2, d.w3, d.w4 = (ptr(w) for w in self.weights)
        d.c_trunc, d.trunc = float(tr["c_trunc"]), float(tr["trunc"])
        d.tsdf_scale = d.c_trunc / d.trunc
        d.clamp_mode, d.clamp_hi = 0, 1.0
        d.pos_fp16 = 1 if getattr(m.embedpos_fn, "fp16", False) else 0
        d.staged = None
        check(_lib.load().rfx_field_stage_weights(C.byref(d), ptr(self.staged), stream_ptr(self.staged.device)), "rfx_field_stage_weights")
        d.staged = self.staged.data_ptr()
        return d

    @staticmethod
    def _flat(q):
        return torch.reshape(q, [-1, q.shape[-1]]).to(torch.float32).contiguous()

    def query_sdf_res(self, query_points):
        import ctypes as C
        x = self._flat(query_points)
        out = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
        d = self._desc()
        check(_lib.load().rfx_field_query_sdf(C.byref(d), ptr(x), x.shape[0], ptr(out), stream_ptr(x.device)), "rfx_field_query_sdf")
        return torch.reshape(out, list(query_points.shape[:-1]))

    def query_w_res(self, query_points):
        x = self._flat(query_points)
        out = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device)
        check(_lib.load().rfx_grid_encode_forward(self.model.GBW.desc, ptr(self.gbw), ptr(x), x.shape[0], ptr(out), stream_ptr(x.device)),
              "rfx_grid_encode_forward")
        return torch.reshape(out, list(query_points.shape[:-1]))

    def query_color_residual(self, query_points):
        import ctypes as C
        x = self._flat(query_points)
        out = torch.empty((x.shape[0], 3), dtype=torch.float32, device=x.device)
        d = self._desc()
        check(_lib.load().rfx_field_query_color(C.byref(d), ptr(x), x.shape[0], ptr(out), stream_ptr(x.device)), "rfx_field_query_color")
        return out


class AsyncMeshExporter:
    """``submit`` copies the field on the caller's stream into a snapshot slot and hands the sweep + marching cubes + PLY write
    to a worker thread that runs them on the slot's side stream behind an event; a slot is re-used only after its export has
    finished (``submit`` waits for it).  An export can only start once the GPU has reached its snapshot, i.e. after the whole
    keyframe's iterations queued before it, so with ONE slot the next ``submit`` usually finds the previous export still waiting
    for the GPU and blocks: that is back-pressure, not a loss -- at config 5's sizes the GPU is busy throughout (15 iterations
    = 20 ms per keyframe), and two slots (DEPTH = 2, measured) give the same 221 frames/s while holding a second 320 MB copy.
    ``result()`` joins everything and returns the mesh of the LATEST submit (or re-raises a worker's exception).  An exporter
    still working when the interpreter exits is drained first (a worker inside a HIP call at exit aborts the process)."""

    DEPTH = 1

    def __init__(self, model, config, bounding_box, marching_cube_bound):
        import threading
        self._threading = threading
        self.model, self.config, self.bb, self.mcb = model, config, bounding_box, marching_cube_bound
        self._slots = [{"snapshot": None, "stream": torch.cuda.Stream(device=bounding_box.device), "thread": None}
                       for _ in range(self.DEPTH)]
        self._seq, self._mesh, self._mesh_seq, self._error = 0, None, -1, None
        self._lock = threading.Lock()
        import atexit
        import weakref
        me = weakref.ref(self)

        def _drain():                       # an interpreter that exits under a running export aborts (worker inside a HIP call)
            obj = me()
            if obj is not None:
                try:
                    obj.wait()
                except BaseException as e:  # noqa: BLE001 -- cannot be raised at exit; a lost mesh must not be silent
                    import sys
                    sys.stderr.write(f"remixfusion_amd.mesh: an asynchronous mesh export failed and its file is missing: {e!r}\n")
        atexit.register(_drain)

    def _work(self, slot, seq, path, voxel_size, ev):
        try:
            with torch.cuda.stream(slot["stream"]):
                slot["stream"].wait_event(ev)
                s = slot["snapshot"]
                mesh = extract_mesh(s.query_sdf_res, s.query_w_res, self.config, self.bb, color_func=s.query_color_residual,
                                    marching_cube_bound=self.mcb, voxel_size=voxel_size)
                import os
                os.makedirs(os.path.dirname(path), exist_ok=True)
                write_ply(path, mesh)
                slot["stream"].synchronize()
            with self._lock:
                if seq > self._mesh_seq:
                    self._mesh, self._mesh_seq = mesh, seq
        except BaseException as e:          # noqa: BLE001 -- handed to the thread that asks for the result
            with self._lock:
                self._error = e

    def _join(self, slot):
        if slot["thread"] is not None:
            slot["thread"].join()
            slot["thread"] = None
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def submit(self, path, voxel_size):
        slot = self._slots[self._seq % self.DEPTH]
        self._join(slot)                                # the export that used this slot's buffers, two submits ago
        if slot["snapshot"] is None:
            slot["snapshot"] = FieldSnapshot(self.model)
        slot["snapshot"].capture()
        ev = torch.cuda.Event()
        ev.record()
        slot["thread"] = self._threading.Thread(target=self._work, args=(slot, self._seq, path, voxel_size, ev), daemon=True)
        slot["thread"].start()
        self._seq += 1

    def wait(self):
        for k in range(self.DEPTH):                     # oldest first
            self._join(self._slots[(self._seq + k) % self.DEPTH])

    def result(self):
        self.wait()
        return self._mesh
