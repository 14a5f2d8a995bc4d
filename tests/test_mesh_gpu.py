"""GPU tests of the device marching cubes: triangle-exact against oracle/mesh_oracle.py with the
same case tables, topology at sizes the Python oracle cannot reach, and the mesh export of a mapped scene."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sphere(N, r, dtype=np.float32):
    g = np.stack(np.meshgrid(*[np.arange(N)] * 3, indexing="ij"), -1).astype(np.float64)
    return (np.linalg.norm(g - (N - 1) / 2, axis=-1) - r).astype(dtype)


def _topology_gpu(verts, faces):
    """closedness / orientation / Euler characteristic with torch ops (large meshes)."""
    nv = verts.shape[0]
    e = torch.cat([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    fwd = e[:, 0] * nv + e[:, 1]
    rev = e[:, 1] * nv + e[:, 0]
    uf, cf = torch.unique(fwd, return_counts=True)
    closed = bool((cf == 1).all()) and torch.equal(uf, torch.unique(rev))
    chi = nv - uf.shape[0] // 2 + faces.shape[0]
    v = verts.double()
    vol = (v[faces[:, 0]] * torch.cross(v[faces[:, 1]], v[faces[:, 2]], dim=-1)).sum() / 6
    return closed, int(chi), float(vol)


def test_emit_matches_oracle_triangles():
    from oracle import mesh_oracle as mo
    from remixfusion_amd.mesh import EDGE_CORNERS, build_tables, marching_cubes
    n_tri, tab, _ = build_tables()
    rng = np.random.default_rng(5)
    noise = rng.standard_normal((9, 12, 7)).astype(np.float32)
    noise[2, 3, 4] = np.nan                                    # NaN samples disable their 8 cells
    mask = rng.uniform(size=noise.shape) > 0.1
    for field, level, m in ((noise, 0.0, None), (noise, 0.3, mask), (_sphere(12, 3.7), 0.0, None)):
        ref_v, ref_k = mo.polygonise(field, level, m, n_tri, tab, EDGE_CORNERS)
        tv, faces = marching_cubes(torch.from_numpy(field).cuda(), level,
                                   None if m is None else torch.from_numpy(m).cuda(), weld=False)
        assert faces.shape[0] * 3 == ref_v.shape[0]
        # same triangles in the same (cell-major, table) order, bit-identical corner positions
        assert np.array_equal(tv.cpu().numpy().view(np.uint32), ref_v.view(np.uint32))
        v, f = marching_cubes(torch.from_numpy(field).cuda(), level, None if m is None else torch.from_numpy(m).cuda())
        rv, rf = mo.weld(ref_v, ref_k)
        assert v.shape[0] == rv.shape[0] and np.array_equal(f.cpu().numpy(), rf)
        assert np.array_equal(v.cpu().numpy(), rv.astype(np.float32))


def test_large_volumes_closed_and_oriented():
    from remixfusion_amd.mesh import marching_cubes
    N, r = 160, 51.3
    v, f = marching_cubes(torch.from_numpy(_sphere(N, r)).cuda(), 0.0)
    closed, chi, vol = _topology_gpu(v, f)
    assert closed and chi == 2
    assert abs(vol / (4 / 3 * np.pi * r ** 3) - 1) < 2e-3
    # torus (genus 1) and a random field (every ambiguous configuration), padded so the surface closes
    g = torch.stack(torch.meshgrid(*[torch.arange(96.0)] * 3, indexing="ij"), -1).cuda() - 47.5
    tor = (torch.sqrt(g[..., 0] ** 2 + g[..., 1] ** 2) - 28.0) ** 2 + g[..., 2] ** 2 - 9.5 ** 2
    closed, chi, vol = _topology_gpu(*marching_cubes(tor, 0.0))
    assert closed and chi == 0 and vol > 0
    gen = torch.Generator(device="cuda").manual_seed(0)
    pad = torch.ones((130, 120, 110), device="cuda")
    pad[1:-1, 1:-1, 1:-1] = torch.randn((128, 118, 108), device="cuda", generator=gen)
    closed, chi, vol = _topology_gpu(*marching_cubes(pad, 0.0))
    assert closed and vol > 0


def test_empty_and_masked():
    from remixfusion_amd.mesh import marching_cubes
    vol = torch.ones((8, 8, 8), device="cuda")
    v, f = marching_cubes(vol, 0.0)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    s = torch.from_numpy(_sphere(20, 6.0)).cuda()
    v, f = marching_cubes(s, 0.0, mask=torch.zeros_like(s, dtype=torch.bool))
    assert f.shape[0] == 0
    half = torch.zeros_like(s, dtype=torch.bool)
    half[:10] = True                                           # open surface: only the lower half is meshed
    v, f = marching_cubes(s, 0.0, mask=half)
    assert f.shape[0] > 0 and float(v[:, 0].max()) <= 9.0


def test_mesh_of_a_mapped_scene(tmp_path):
    """save_mesh after a short mapping run: surface near the synthetic room's walls, colours attached."""
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.pipeline import MappingPipeline
    cfg = synthetic_config("office0")
    cfg["cam"].update({"H": 120, "W": 160, "fx": 144.0, "fy": 144.0, "cx": 79.5, "cy": 59.5})
    cfg["volume"].update({"voxel_size": 0.04, "trunc": 0.15})
    cfg["mapping"].update({"first_iters": 50, "sample": 512})
    cfg["synthetic"].update({"depth_noise": 0.0, "dropout": 0.0})
    cfg["data"].update({"output": str(tmp_path), "exp_name": "t"})
    cfg["mapping"]["marching_cubes_bound"] = cfg["mapping"]["bound"]      # the synthetic room's walls lie outside office0's
    pipe = MappingPipeline(cfg, n_frames=20, seed=1)
    frames = pipe.prefetch(list(range(11)))
    pipe.track_frame(0, frames[0])
    pipe.mapper.init_mapvolume()
    pipe.mapper.first_frame_mapping({k: v for k, v in frames[0].items() if k != "rgb255"}, cfg["mapping"]["first_iters"])
    for i in range(1, 11):
        pipe.step(i, frames[i])
    ds = pipe.dataset
    lo, hi = ds.room[:, 0].cuda(), ds.room[:, 1].cuda()

    def surface_distance(p):                                   # analytic scene: box walls + two spheres
        d = torch.minimum((p - lo).abs(), (p - hi).abs()).min(-1).values
        for c, r in ds.spheres:
            d = torch.minimum(d, ((p - c.cuda()).norm(dim=-1) - r).abs())
        return d

    bb = torch.tensor(cfg["mapping"]["marching_cubes_bound"], device="cuda", dtype=torch.float32)
    for mesh in (pipe.slam.save_mesh(10), pipe.slam.save_mesh_explicit(10)):
        v, f, c = mesh["vertices"], mesh["faces"], mesh["colors"]
        assert f.shape[0] > 5000 and c.shape == (v.shape[0], 3) and c.dtype == torch.uint8
        assert int(f.min()) >= 0 and int(f.max()) < v.shape[0]
        assert bool(((v >= bb[:, 0] - 1e-4) & (v <= bb[:, 1] + 1e-4)).all())
        d = surface_distance(v.float())
        assert float(d.median()) < 0.03 and float((d < 0.1).float().mean()) > 0.9, (float(d.median()), float((d < 0.1).float().mean()))
        assert float(c.float().std()) > 5.0                    # colours vary over the checkerboard
    # the export the mapper's loop uses (SLAM.save_mesh_async): the field is copied at the call, swept by a worker thread on a
    # side stream -- the same mesh as the blocking export of the same state, even though the field moves on meanwhile
    ref = pipe.slam.save_mesh(10)
    ex = pipe.slam.save_mesh_async(11)
    with torch.no_grad():
        saved = pipe.model.embed_res_fn.params.detach().clone()
        pipe.model.embed_res_fn.params.mul_(0.5)                # the mapper's next step, as far as the export is concerned
    got = ex.result()
    with torch.no_grad():
        pipe.model.embed_res_fn.params.copy_(saved)
    for k in ("vertices", "faces", "colors"):
        assert torch.equal(got[k], ref[k]), k
    import os
    assert os.path.getsize(os.path.join(str(tmp_path), "t", "mesh_track11.ply")) == os.path.getsize(os.path.join(str(tmp_path), "t", "mesh_track10.ply"))
    assert os.path.getsize(os.path.join(str(tmp_path), "t", "mesh_track10.ply")) > 10000
    with open(os.path.join(str(tmp_path), "t", "mesh_track10_ex.ply"), "rb") as fh:
        assert fh.readline().strip() == b"ply"
