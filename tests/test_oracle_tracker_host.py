"""The tracker's HOST logic (reference model/ROtracker.py: cal_transform :606-709, update_PST :493-534, the bookkeeping of
random_optimization :745-826): the scalar-loop oracle against closed forms, and the product's vectorised host code against the
oracle bit for bit.  (The device-side search is compared with the product's host code on the GPU: tests/test_tracker_gpu.py.)"""
import numpy as np

from oracle import tracker_host_oracle as O

DEPTH_LEVEL = [32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16, 8, 32, 16]


def _template(rng, n, scale=1.0):
    t = rng.uniform(-1, 1, (n, 6)).astype(np.float32) * np.float32(scale)
    t[0] = 0
    return t


def test_oracle_cal_transform_closed_forms():
    rng = np.random.default_rng(0)
    cand = _template(rng, 64)
    ss = np.array([0.02, 0.03, 0.01, 0.02, 0.015, 0.01], np.float32)
    # nothing beats the null candidate (ties do not count): failure, min_tsdf = the null candidate's value, zeros
    sv = np.full(64, 0.5, np.float32)
    ok, m, mt, bad = O.cal_transform(sv, cand, ss, 200)
    assert not ok and m == np.float32(0.5) and not mt.any() and not bad
    # one better candidate: the mean IS that candidate (scaled by the box), the quaternion normalised
    sv[17] = 0.25
    ok, m, mt, bad = O.cal_transform(sv, cand, ss, 200)
    q = cand[17, 3:6].astype(np.float64) * ss[3:6]
    quat = np.concatenate([[np.sqrt(1 - (q ** 2).sum())], q])
    assert ok and abs(m - 0.25) < 1e-7
    assert np.allclose(mt[:3], cand[17, :3] * ss[:3], rtol=3e-7) and np.allclose(mt[3:], quat / np.linalg.norm(quat), rtol=3e-7)
    # weights are (origin - fit): two candidates 3:1
    sv[40] = 0.4375                                           # weight 0.0625 against 0.25
    ok, m, mt, bad = O.cal_transform(sv, cand, ss, 200)
    assert np.allclose(mt[:3], (0.8 * cand[17, :3] + 0.2 * cand[40, :3]) * ss[:3], rtol=1e-6)
    assert abs(m - (0.8 * 0.25 + 0.2 * 0.4375)) < 1e-7
    # only the first count_search better candidates, in index order
    sv[5] = 0.1
    ok, m, mt, bad = O.cal_transform(sv, cand, ss, 2)         # 5 and 17; 40 is cut
    w5, w17 = 0.4, 0.25
    assert np.allclose(mt[:3], (w5 * cand[5, :3] + w17 * cand[17, :3]) / (w5 + w17) * ss[:3], rtol=1e-6)
    # a selected candidate whose scaled vector part is longer than 1: the reference exits
    big = cand.copy()
    big[17, 3:6] = 60.0
    assert O.cal_transform(sv, big, ss, 200)[3]
    # update_PST: the box follows |mean| + 1e-3, normalised over the six, times scale * tsdf, + 1e-3
    box = ss.copy()
    mt = np.array([0.003, 0, 0, 1, 0, 0.004, 0], np.float32)
    O.update_PST(box, 0.2, mt, scale=0.09)
    s = np.array([0.004, 0.001, 0.001, 0.001, 0.005, 0.001])
    assert np.allclose(box, 0.09 * 0.2 * s / np.linalg.norm(s) + 1e-3, rtol=1e-6)


def _product_tracker(cand_by_step, count_search=200, fix_level_index=0, iterative_scale=True, scaling=0.09):
    from remixfusion_amd.model.ROtracker import ROTracker
    tr = ROTracker.__new__(ROTracker)
    tr.count_search, tr.fix_level_index, tr.iterative_scale, tr.scaling_coefficient = count_search, fix_level_index, iterative_scale, scaling
    tr.depth_level = DEPTH_LEVEL
    tr.init_size = 0.02
    tr.init_searchsize()
    tr.previous_frame_success = False
    tr.initialize_search_size = np.zeros(6)
    return tr


def test_product_host_search_step_is_the_oracle_bit_for_bit():
    """20-iteration searches on synthetic fitness values (successes, failures, count_search cuts, both level-index modes): after
    every iteration pose, box, previous box, template index, pixel offset and flags equal the oracle's; everything bit for bit
    except R, where np.matmul may fuse (<= 1 float32 ulp of 1)."""
    rng = np.random.default_rng(1)
    templates = [_template(rng, n, 1.0) for n in (1024, 3072, 10240)]
    n_fail = n_ok = n_cut = 0
    for trial in range(12):
        count_search = (200, 200, 7, 512)[trial % 4]
        fix = trial % 3 == 0
        it_scale = trial % 5 != 0
        tr = _product_tracker(None, count_search, int(fix), it_scale, scaling=(0.09, 0.12)[trial % 2])
        R0 = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        T0 = rng.normal(size=3).astype(np.float32)
        tr.current_global_R, tr.current_global_T = R0.copy(), T0.copy()
        st = O.SearchState(R0, T0, tr.search_size)
        host = {"previous_success": False, "success": False, "count_particle": 0, "level_index": 5}
        for i in range(20):
            cp = st.template()
            if not host["success"]:
                host["count_particle"] = 0
            assert host["count_particle"] == cp
            cand = templates[(2, 1, 0)[cp % 3]]
            n = cand.shape[0]
            mode = rng.integers(0, 4)
            sv = rng.uniform(0.2, 0.6, n).astype(np.float32)
            if mode == 0:
                sv[0] = 0.1                                   # nothing better: a failed iteration
            elif mode == 1:
                sv[0] = 0.21                                  # a handful better
            # evaluated hit counts of zero give exactly 0 (value 0 / 1e-6): better than anything positive
            sv[rng.integers(1, n, 3)] = 0.0 if mode == 3 else sv[1]
            tr.transform_candidate = cand
            tr._search_step(i, host, sv.copy(), 0.9)
            bad = O.search_step(st, i, sv, cand, DEPTH_LEVEL, count_search, tr.scaling_coefficient, fix, it_scale, 0.9)
            assert not bad
            n_ok += st.success
            n_fail += not st.success
            n_cut += int((sv[1:] < sv[0]).sum() > count_search)
            tag = (trial, i, mode)
            assert host["success"] == st.success and host["previous_success"] == st.previous_success, tag
            assert host["count_particle"] == st.count_particle and host["level_index"] == st.level_index, tag
            assert np.array_equal(tr.search_size, st.search_size) and tr.search_size.dtype == np.float32, tag
            assert np.array_equal(tr.previous_search_size, st.previous_search_size), tag
            assert np.array_equal(tr.current_global_T, st.T), tag
            assert np.float64(host["min_tsdf"]) == np.float64(st.min_tsdf), tag
            assert np.abs(tr.current_global_R - st.R).max() <= 1.2e-7, tag
            tr.current_global_R = st.R.copy()                 # (so that a fused product does not accumulate)
        assert tr.previous_frame_success == st.first_success
    assert n_ok > 60 and n_fail > 30 and n_cut > 60, (n_ok, n_fail, n_cut)
