"""CPU tests: the oracle (and the torch/numpy host logic of the product) against golden vectors
generated from the reference's own code (tests/golden/make_golden.py)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import field_oracle as FO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def L(name):
    return {k: v for k, v in np.load(os.path.join(G, name)).items()}


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_decoder_forward_and_grads_match_reference():
    g = L("decoder.npz")
    W = [T(g[k]).clone().requires_grad_(True) for k in ("W1", "W2", "W3", "W4")]
    ins = [T(g[k]).clone().requires_grad_(True) for k in ("emb", "pos", "tsdf", "ex_rgb")]
    out = FO.mlp_forward(*ins, *W)
    assert torch.allclose(out, T(g["out"]), rtol=1e-6, atol=1e-7)
    out.backward(T(g["gout"]))
    for t, k in zip(W + ins, ("dW1", "dW2", "dW3", "dW4", "d_emb", "d_pos", "d_tsdf", "d_rgb")):
        assert torch.allclose(t.grad, T(g[k]), rtol=1e-5, atol=1e-6), k
    # the product's ColorSDFNet has the same parameter layout / concat order
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.decoder import ColorSDFNet
    net = ColorSDFNet(synthetic_config("office0"), input_ch=32, input_ch_pos=48)
    with torch.no_grad():
        for p, k in zip(net.fused_weights(), ("W1", "W2", "W3", "W4")):
            p.copy_(T(g[k]))
    assert torch.allclose(net(T(g["emb"]), T(g["pos"]), T(g["tsdf"]), T(g["ex_rgb"])), T(g["out"]), rtol=1e-6, atol=1e-7)
    assert [k for k, _ in net.state_dict().items()] == ["color_net.model.0.weight", "color_net.model.2.weight",
                                                        "sdf_net.model.0.weight", "sdf_net.model.2.weight"]


def test_render_matches_reference():
    g = L("render.npz")
    rgb, depth = FO.raw2outputs(T(g["raw"]), T(g["z"]), float(g["trunc"]), float(g["sc_factor"]))
    assert torch.allclose(rgb, T(g["rgb"]), rtol=1e-6, atol=1e-7) and torch.allclose(depth, T(g["depth"]), rtol=1e-6, atol=1e-7)
    w = FO.sdf2weights(T(g["raw"])[..., 3], T(g["z"]), float(g["trunc"]), float(g["sc_factor"]))
    assert torch.allclose(w, T(g["weights"]), rtol=1e-6, atol=1e-8)


def test_losses_match_reference():
    g = L("losses.npz")
    from remixfusion_amd.model import utils as PU
    for mod in (FO, PU):
        front, sdfm, fw, sw = mod.get_masks(T(g["z"]), T(g["td"]), 0.05)
        assert torch.equal(front, T(g["front"])) and torch.equal(sdfm, T(g["sdf_mask"]))
        assert abs(float(fw) - float(g["fs_w"])) < 1e-7 and abs(float(sw) - float(g["sdf_w"])) < 1e-7
    fs, sl = FO.get_sdf_loss(T(g["z"]), T(g["td"]), T(g["sdf"]), 0.05, middle_mask=T(g["mid"]))
    assert abs(float(fs) - float(g["fs"])) < 1e-6 * abs(float(g["fs"])) + 1e-9 and abs(float(sl) - float(g["sl"])) < 1e-6 * abs(float(g["sl"])) + 1e-9
    fs, sl = PU.get_sdf_loss(T(g["z"]), T(g["td"]), T(g["sdf"]), 0.05, "l2", middle_mask=T(g["mid"]))
    assert abs(float(fs) - float(g["fs"])) < 1e-6 * abs(float(g["fs"])) + 1e-9 and abs(float(sl) - float(g["sl"])) < 1e-6 * abs(float(g["sl"])) + 1e-9
    fs0, sl0 = PU.get_sdf_loss(T(g["z"]), T(g["td"]), T(g["sdf"]), 0.05, "l2")
    assert abs(float(fs0) - float(g["fs0"])) < 1e-6 and abs(float(sl0) - float(g["sl0"])) < 1e-6
    assert abs(float(PU.compute_loss(T(g["sdf"]), T(g["z"]))) - float(g["l2"])) < 1e-5
    assert abs(float(PU.compute_loss(T(g["sdf"]), T(g["z"]), "l1")) - float(g["l1"])) < 1e-5


def oracle_mapping(g, cfg, clamp):
    meta = FO.hashgrid_meta_from_config(int(g["hash_T"]), int(g["hash_R"]))
    fp = FO.FieldParams(hash_meta=meta, hash_table=T(g["table"]), gbv=T(g["gbv"]), gbw=torch.zeros(1), gbv_res=int(g["gbv_res"]),
                        W1=T(g["W1"]), W2=T(g["W2"]), W3=T(g["W3"]), W4=T(g["W4"]), c_trunc=cfg["training"]["c_trunc"],
                        trunc=cfg["training"]["trunc"], map_clamp=cfg["mapping"]["clamp"])
    tr, cam = cfg["training"], cfg["cam"]
    z = FO.sample_z_vals(T(g["td"]), cam["near"], cam["far"], tr["range_d"], tr["n_range_d"], tr["n_samples_d"], 0)
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    rend = FO.render_rays(fp, bb, T(g["o"]), T(g["d"]), z, clamp=clamp, sc_factor=cfg["data"]["sc_factor"])
    loss = FO.mapping_losses(rend["rgb_res_map"], rend["depth_res_map"], rend["raw"], z, T(g["tgt"]), T(g["td"]),
                             depth_trunc=cam["depth_trunc"], rgb_missing=tr["rgb_missing"], trunc=tr["trunc"],
                             sc_factor=cfg["data"]["sc_factor"])
    return rend, loss


@pytest.mark.parametrize("name", ["office0", "scene0000"])
def test_mapping_pipeline_matches_reference(name):
    """z sampler, normalisation, tsdf rescale/clamp, residual add, rendering and the four losses as
    executed by the reference's JointEncoding.mapping() (with the oracle's encoders plugged in)."""
    from remixfusion_amd.config import synthetic_config
    cfg = synthetic_config(name)
    g = L(f"mapping_{name}.npz")
    for clamp, tag in ((False, "c0"), (True, "c1")):
        rend, loss = oracle_mapping(g, cfg, clamp)
        assert torch.allclose(rend["z_vals"], T(g[f"{tag}_z_vals"]), rtol=0, atol=1e-6)
        assert torch.allclose(rend["raw"], T(g[f"{tag}_raw"]), rtol=1e-5, atol=1e-6)
        assert torch.allclose(rend["rgb_res_map"], T(g[f"{tag}_rgb_res_map"]), rtol=1e-5, atol=1e-6)
        assert torch.allclose(rend["depth_res_map"], T(g[f"{tag}_depth_res_map"]), rtol=1e-5, atol=1e-6)
        for k in ("rgb_res_loss", "depth_res_loss", "sdf_res_loss", "fs_res_loss"):
            ref = float(g[f"{tag}_{k}"])
            assert abs(float(loss[k]) - ref) <= 1e-5 * abs(ref) + 1e-9, (k, float(loss[k]), ref)


def test_host_helpers_match_reference(tmp_path):
    g = L("host.npz")
    from remixfusion_amd.config import load_config, synthetic_config
    from remixfusion_amd.datasets import get_camera_rays
    from remixfusion_amd.model.keyframe import KeyFrameDatabase
    from remixfusion_amd.model.utils import batchify
    assert torch.equal(get_camera_rays(12, 16, 14.4, 14.0, 7.5, 5.5), T(g["rays"]))
    assert torch.equal(batchify(lambda x: x * 2 + 1, 7)(torch.arange(23.0)[:, None]), T(g["batchified"]))
    c2, c3 = synthetic_config("office0"), synthetic_config("scene0000")
    assert np.array_equal(np.array(c2["mapping"]["bound"], float), g["office0_bound"])
    assert c2["mapping"]["iters"] == int(g["office0_iters"]) and c2["grid"]["hash_size"] == int(g["office0_hash"])
    assert c2["volume"]["voxel_size"] == float(g["office0_voxel"])
    assert c3["grid"]["hash_size"] == int(g["scene0000_hash"]) and c3["mapping"]["clamp"] == float(g["scene0000_clamp"])
    assert c3["training"]["n_range_d"] == int(g["scene0000_n_range_d"]) and c3["volume"]["voxel_size"] == float(g["scene0000_vox"])
    assert c3["volume"]["x_config"]["len"] == float(g["scene0000_xlen"])
    # inherit_from + deep merge
    base = tmp_path / "base.yaml"
    base.write_text("a: {b: 1, c: {d: 2, e: 3}}\nf: 4\n")
    child = tmp_path / "child.yaml"
    child.write_text(f"inherit_from: {base}\na: {{c: {{d: 20}}, g: 7}}\nh: [1, 2]\n")
    cfg = load_config(str(child))
    assert cfg["a"] == {"b": 1, "c": {"d": 20, "e": 3}, "g": 7} and cfg["f"] == 4 and cfg["h"] == [1, 2]
    # keyframe store: same python RNG stream -> same rays as the reference
    rays = get_camera_rays(12, 16, 14.4, 14.0, 7.5, 5.5)
    c2["mapping"]["device_sampling"] = False          # reference behaviour: python's random stream
    kf = KeyFrameDatabase(c2, 12, 16, 4, 30, "cpu")
    random.seed(1234)
    gen = torch.Generator().manual_seed(1)
    for fid in (0, 5, 10):
        kf.add_keyframe({"frame_id": fid, "direction": rays[None], "rgb": torch.rand(1, 12, 16, 3, generator=gen),
                         "depth": torch.rand(1, 12, 16, generator=gen)})
    sr, ids = kf.sample_global_rays(40)
    assert torch.equal(kf.rays, T(g["kf_rays"])) and torch.equal(kf.frame_ids, T(g["kf_ids"]))
    assert torch.equal(sr, T(g["sample_rays"])) and torch.equal(ids, T(g["sample_ids"]))


def test_moving_volume_bound_logic_matches_reference():
    g = L("volume_bounds.npz")
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.Volume import moving_volume

    class Traj:
        kfx = kfy = kfz = 0.0
        first = 0

    def bare(version="center", fix_z=0):
        v = synthetic_config("office0")["volume"]
        mv = moving_volume.__new__(moving_volume)
        mv.voxel_size = 0.05
        mv.first_len, mv.second_len, mv.third_len, mv.more_angel_t = v["first_len"], v["second_len"], v["third_len"], v["more_angel_t"]
        mv.fix_x, mv.fix_y, mv.fix_z = 0, 0, fix_z
        mv.x_len, mv.y_len, mv.z_len = 4, 4, 3
        mv.x_range, mv.y_range, mv.z_range = [0, 1], [0, 1], [-1.5, 2.5]
        mv.version, mv.t_treshold = version, 1
        mv.last_pcid, mv.surface_pc = 0, None
        return mv

    def pose(t, yaw):
        c, s = np.cos(yaw), np.sin(yaw)
        P = np.eye(4)
        P[:3, :3] = np.array([[-s, 0, c], [c, 0, s], [0, -1, 0]])
        P[:3, 3] = t
        return P

    mv, tr = bare(), Traj()
    assert np.array_equal(mv.center_volbnd(None, pose([0.4, -1.6, 0.2], 0.3), tr), g["center_bnds"])
    assert np.array_equal([tr.kfx, tr.kfy, tr.kfz], g["center_anchor"])
    angs = []
    for v in ([1, 0, 0], [0.3, -0.8, 0.1], [-1, 0.2, 0.5], [0, 0, 1]):
        x = np.asarray(v, np.float32)
        for ax in np.eye(3, dtype=np.float32):
            a, f = mv.require_angle(x, ax)
            angs.append([a, f, mv.require_angle(x, ax, True)])
            for fixed in ("x", "y", "z"):
                a, f = mv.require_angle_projection(x, ax, fixed=fixed)
                angs.append([a, f, mv.require_angle_projection(x, ax, True, fixed=fixed)])
    assert np.allclose(np.array(angs), g["angles"], rtol=0, atol=1e-12, equal_nan=True)
    mvm, trm = bare("more", fix_z=1), Traj()
    more = np.stack([mvm.more_volbnd(None, pose([0.4, -1.6, 0.2], yaw), trm) for yaw in (0.1, 1.4, 2.9, -1.7)])
    assert np.array_equal(more, g["more_bnds"]) and trm.first == int(g["more_first"])
    assert np.array_equal(mvm.more_calculations(np.zeros((3, 2)), [1, 0, 2], [1, -1, 1], np.array([2.0, -3.0, 1.0])), g["more_calc"])
    # scripted camera walk: which moves trigger copy + swap, and the bounds / anchors after each
    mv, tr = bare(), Traj()
    mv.vol_bnds = np.asarray(mv.center_volbnd(None, pose([0.2, 0.1, 0.0], 0.0), tr))
    log = []
    mv.copy_volume = lambda: log.append("copy")

    def swap(new, old):
        log.append("swap")
        mv.vol_bnds = new
    mv.update_tsdf_swap_rot_trans = swap
    for i, t in enumerate(g["walk"]):
        f, old = mv.check_move_volume_new(i, pose(t, 0.2 * i), tr)
        assert bool(f) == bool(g["walk_flags"][i])
        assert np.array_equal(old, g["walk_old"][i]) and np.array_equal(mv.vol_bnds, g["walk_new"][i])
        assert np.array_equal([tr.kfx, tr.kfy, tr.kfz], g["walk_anchor"][i])
    assert log.count("copy") == int(g["walk_n_copy"]) and log.count("swap") == int(g["walk_n_swap"])


def test_tracker_host_logic_matches_reference():
    """tests/golden/tracker_host.npz: the reference's own cal_transform / update_PST / random_optimization (model/ROtracker.py
    :606-709, :493-534, :713-831) run in the authoring container on injected fitness arrays (numpy 2.x arithmetic: a Python scalar
    does not promote a float32).  oracle/tracker_host_oracle.py in mode="numpy2" reproduces every output BIT FOR BIT (R of the
    whole loops to one float32 ulp: np.matmul may fuse): candidate selection in index order with the count_search cut, ties,
    unevaluated candidates, weights, the mean and its normalisation, failure returns, the box update, template / pixel-offset
    schedule, smoothing and flags are the reference's.  (mode="numpy1", which the product follows, differs in the documented
    casts only; tests/test_oracle_tracker_host.py.)"""
    from oracle import tracker_host_oracle as O
    g = np.load(os.path.join(G, "tracker_host.npz"))
    assert str(g["numpy_version"]).startswith("2.")
    templates = {c: g[f"template_{c}"] for c in range(3)}
    depth_level, tiff_index = [int(v) for v in g["depth_level"]], [int(v) for v in g["tiff_index"]]
    n_fail = n_ok = 0
    for k in range(int(g["n_cal_transform"])):
        cls, slot, count_search = (int(v) for v in g[f"ct{k}_template"])
        sv, ss = g[f"ct{k}_search_value"], g[f"ct{k}_search_size"].copy()
        ok, m, mt, bad = O.cal_transform(sv, templates[cls][slot], ss, count_search, mode="numpy2")
        assert not bad and ok == bool(g[f"ct{k}_success"]), k
        assert not bool(g[f"ct{k}_min_tsdf_is_f64"]) and np.float32(m) == np.float32(g[f"ct{k}_min_tsdf"]), k
        assert np.array_equal(mt, g[f"ct{k}_mean_transform"]), (k, mt, g[f"ct{k}_mean_transform"])
        O.update_PST(ss, m, mt, scale=0.09 if k % 2 else 0.12, mode="numpy2")
        assert np.array_equal(ss, g[f"ct{k}_search_size_after"]), (k, ss, g[f"ct{k}_search_size_after"])
        n_ok += ok
        n_fail += not ok
    assert n_ok >= 6 and n_fail >= 2
    n_loop_fail = 0
    for k in range(int(g["n_loops"])):
        count_search, fix, it_scale = (int(v) for v in g[f"loop{k}_config"])
        scaling = float(g[f"loop{k}_scaling"])
        pose = g[f"loop{k}_pose_in"]
        st = O.SearchState(pose[:3, :3], pose[:3, 3], np.full(6, 0.02, np.float32))
        for i in range(20):
            cp = st.template()
            tiff = tiff_index[cp]
            cand = templates[tiff // 20][(tiff % 20) // 3]
            n = int(g[f"loop{k}_fed_n"][i])
            assert cand.shape[0] == n, (k, i)
            assert (depth_level[cp], st.level_index) == tuple(int(v) for v in g[f"loop{k}_levels"][i]), (k, i)   # what evaluate_tsdf was asked for
            bad = O.search_step(st, i, g[f"loop{k}_fed"][i, :n], cand, depth_level, count_search, scaling, bool(fix), bool(it_scale), 0.9,
                                mode="numpy2")
            assert not bad
            n_loop_fail += not st.success
        out = g[f"loop{k}_pose_out"]
        assert np.array_equal(st.T, out[:3, 3]), (k, st.T, out[:3, 3])
        assert np.abs(st.R - out[:3, :3]).max() <= 2.4e-7, k                # 20 float32 3x3 products: BLAS may fuse
        assert np.array_equal(st.search_size, g[f"loop{k}_search_size"]), (k, st.search_size, g[f"loop{k}_search_size"])
        assert np.array_equal(st.previous_search_size, g[f"loop{k}_previous_search_size"]), k
        assert st.first_success == bool(g[f"loop{k}_previous_frame_success"]), k
    assert n_loop_fail >= 10


def test_constant_velocity_prediction_matches_reference():
    """mp_slam/tracker.py:55-72 (`predict_current_pose`): frame 1 starts from the previous estimate; from frame 2 on the motion
    between the tracker's last two results repeats -- float32 inverse and products on the host, rotation re-orthogonalised by a
    float32 SVD (model/utils.py:63-70).  The product's `constant_velocity` gives the reference's own output bit for bit."""
    from remixfusion_amd.mp_slam.tracker import constant_velocity
    g = np.load(os.path.join(G, "tracker_host.npz"))
    ro, out = torch.from_numpy(g["pred_ro"]), g["pred_out"]
    assert np.array_equal(out[0], ro[0].numpy())
    for f in range(2, out.shape[0] + 1):
        mine = constant_velocity(ro[f - 2].clone(), ro[f - 1].clone())
        assert mine.dtype == torch.float32 and np.array_equal(mine.numpy(), out[f - 1]), f
        R = mine[:3, :3].double()
        assert float((R @ R.T - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-6


def test_smoothness_matches_reference():
    """mp_slam/slam.py:193-217 `SLAM.smoothness` run from the reference's own code (tests/golden/smoothness.npz): the lattice the
    two torch.rand draws place (integer and fractional bounds), its normalisation, and the TV sum over the oracle encoder's
    features.  The oracle's restatement gives the same points and the same value, bit for bit, from the same draws."""
    g = L("smoothness.npz")
    meta = FO.hashgrid_meta_from_config(int(g["hash_T"]), int(g["hash_R"]))
    table = T(g["table"])
    for name in ("int_bounds", "frac_bounds"):
        bbox = T(g[f"{name}_bbox"])
        sp, vox, margin = int(g[f"{name}_args"][0]), float(g[f"{name}_args"][1]), float(g[f"{name}_args"][2])
        torch.manual_seed(int(g[f"{name}_seed"]))
        r_off = torch.rand(3)
        r_jit = torch.rand((1, 1, 1, 3))
        pts = FO.smoothness_points(bbox, sp, vox, margin, r_off, r_jit)
        assert pts.shape == (sp - 1, sp - 1, sp - 1, 3)
        assert torch.equal(pts[::3, ::3, ::3], T(g[f"{name}_pts"])), name
        assert torch.equal(pts.double().sum(dim=(0, 1, 2)), T(g[f"{name}_pts_sum"])), name
        feat = FO.grid_encode(pts.reshape(-1, 3), table, meta).reshape(sp - 1, sp - 1, sp - 1, -1)
        assert torch.equal(FO.smoothness_from_features(feat, sp), T(g[f"{name}_loss"])), name
