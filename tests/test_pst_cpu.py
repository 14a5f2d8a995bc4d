"""PST template loader (model/pst.py; reference model/ROtracker.py:834-866): the baseline-TIFF reader, the ALL_PST
container layout, the explicit generated fallback, and -- where a directory with the reference's 60 TIFFs is available
(RFX_PST_PATH, or the reference checkout in the authoring container) -- identity with the committed digest
tests/golden/pst_fixture.npz of those files."""
import hashlib
import importlib.util
import os
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _pst():
    # loaded by path: importing remixfusion_amd.model pulls in the GPU library binding
    spec = importlib.util.spec_from_file_location("rfx_pst", os.path.join(ROOT, "remixfusion_amd", "model", "pst.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


TIFF_INDEX = [0, 21, 42, 3, 24, 45, 6, 27, 48, 9, 30, 51, 12, 33, 54, 15, 36, 57, 18, 39]   # ROtracker.py:116-124
SIZES = [10240, 3072, 1024]


def _ref_dir():
    for d in (os.environ.get("RFX_PST_PATH"), "/root/reference/PFO/fps_uniform_sphere"):
        if d and os.path.isdir(d):
            return d
    return None


def test_tiff_reader_round_trip_and_errors(tmp_path):
    P = _pst()
    rng = np.random.default_rng(0)
    for shape in ((1024, 6), (1, 6), (7, 3)):
        a = rng.standard_normal(shape).astype(np.float32)
        f = str(tmp_path / "a.tiff")
        P.write_float_tiff(f, a)
        b = P.read_float_tiff(f)
        assert b.dtype == np.float32 and np.array_equal(a, b)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:       # an independent writer (multi-strip) and an independent reader
        a = rng.standard_normal((3072, 6)).astype(np.float32)
        f = str(tmp_path / "pil.tiff")
        Image.fromarray(a, mode="F").save(f)
        assert np.array_equal(P.read_float_tiff(f), a)
        P.write_float_tiff(f, a)
        assert np.array_equal(np.array(Image.open(f)), a)
        Image.fromarray((a * 10).astype(np.uint8)).save(str(tmp_path / "u8.tiff"))
        with pytest.raises(ValueError):
            P.read_float_tiff(str(tmp_path / "u8.tiff"))
    bad = tmp_path / "bad.tiff"
    bad.write_bytes(b"not a tiff at all")
    with pytest.raises(ValueError):
        P.read_float_tiff(str(bad))
    good = open(str(tmp_path / "a.tiff"), "rb").read()
    (tmp_path / "short.tiff").write_bytes(good[:-5])
    with pytest.raises(ValueError):
        P.read_float_tiff(str(tmp_path / "short.tiff"))


def test_container_layout_from_a_directory(tmp_path):
    """load_pst puts pst_{size}_{num}.tiff into ALL_PST[class][num // 3] (reference :853-866)."""
    P = _pst()
    small = [64, 32, 16]
    for ti in TIFF_INDEX:
        cls, num, _ = P.pst_slot(ti)
        a = np.full((small[cls], 6), float(ti), np.float32)
        a[0] = 0
        P.write_float_tiff(str(tmp_path / f"pst_{small[cls]}_{num}.tiff"), a)
    A = P.load_pst(str(tmp_path), small, TIFF_INDEX)
    assert {k: v.shape for k, v in A.items()} == {0: (7, 64, 6), 1: (7, 32, 6), 2: (6, 16, 6)}
    for ti in TIFF_INDEX:
        cls, num, slot = P.pst_slot(ti)
        assert slot == num // 3 and float(A[cls][slot][1, 0]) == float(ti)
    with pytest.raises(FileNotFoundError):
        P.load_pst(str(tmp_path / "nope"), small, TIFF_INDEX)
    P.write_float_tiff(str(tmp_path / "pst_16_2.tiff"), np.zeros((15, 6), np.float32))
    with pytest.raises(ValueError):
        P.load_pst(str(tmp_path), small, TIFF_INDEX)


def test_generated_fallback_has_the_template_structure():
    P = _pst()
    A = P.generated_pst(20251205, SIZES, TIFF_INDEX)
    assert {k: v.shape for k, v in A.items()} == {0: (7, 10240, 6), 1: (7, 3072, 6), 2: (6, 1024, 6)}
    t = A[2][0]
    assert not t[0].any() and float(np.linalg.norm(t, axis=1).max()) <= 1.0 + 1e-6
    assert (np.diff(np.linalg.norm(t[1:], axis=1)) <= 1e-7).all()


def test_fixture_is_self_consistent():
    g = np.load(os.path.join(HERE, "golden", "pst_fixture.npz"))
    assert len(g["names"]) == 60 and g["heads"].shape == (60, 4, 6)
    assert sorted(set(int(s[0]) for s in g["shapes"])) == [1024, 3072, 10240] and all(int(s[1]) == 6 for s in g["shapes"])
    assert not g["heads"][:, 0].any()            # row 0 of every template is the null perturbation


@pytest.mark.skipif(_ref_dir() is None, reason="no directory with the reference's PST TIFFs (set RFX_PST_PATH)")
def test_reference_templates_match_the_committed_digest():
    P = _pst()
    d = _ref_dir()
    g = np.load(os.path.join(HERE, "golden", "pst_fixture.npz"))
    for i, name in enumerate(g["names"]):
        a = P.read_float_tiff(os.path.join(d, str(name)))
        assert a.shape == tuple(g["shapes"][i]) and np.array_equal(a[:4], g["heads"][i])
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g["sha256"][i])
        assert abs(float(a.astype(np.float64).sum()) - float(g["sums"][i])) < 1e-9
    A = P.load_pst(d, SIZES, TIFF_INDEX)
    k = list(g["names"]).index("pst_3072_7.tiff")
    assert np.array_equal(A[1][7 // 3][:4], g["heads"][k])


def test_template_file_of_every_search_step_matches_reference(tmp_path):
    """tests/golden/tracker_host.npz `pst_file_at_step`: the file whose particles the reference's search reads at step k
    (its own readpst + get_PST, model/ROtracker.py:834-866, :474-492, run with cv2.imread handing out arrays that carry their
    file name).  The product's loader + `pst_slot` serve the same file at every step, from the same container shapes."""
    P = _pst()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracker_host.npz"))
    small = [64, 32, 16]
    for size in small:
        for num in range(20):
            P.write_float_tiff(str(tmp_path / f"pst_{size}_{num}.tiff"), np.full((size, 6), float(num + 100 * small.index(size)), np.float32))
    A = P.load_pst(str(tmp_path), small, TIFF_INDEX)
    assert [list(A[c].shape) for c in range(3)] == g["pst_container_shapes"].tolist()
    for k, ti in enumerate(TIFF_INDEX):
        cls, _, slot = P.pst_slot(ti)
        code = int(A[cls][slot][0, 0])
        assert f"pst_{small[code // 100]}_{code % 100}.tiff" == str(g["pst_file_at_step"][k]), k


def test_packaged_archive_holds_the_reference_templates():
    """tests/golden/pst_templates.npz (round 5): the 60 arrays themselves, identical -- SHA-256 of the sample bytes, shape,
    first rows, float64 sums -- to what the committed digest recorded from the reference's TIFFs (PFO/fps_uniform_sphere,
    read by model/ROtracker.py:834-866), and `load_pst` / `resolve_pst_source` serve them where no TIFF directory exists."""
    P = _pst()
    g = np.load(os.path.join(HERE, "golden", "pst_fixture.npz"))
    arc = os.path.join(HERE, "golden", "pst_templates.npz")
    assert os.path.isfile(arc) and not hasattr(P, "PACKAGED_ARCHIVE")          # a fixture of the tests, not a file the product looks for
    z = np.load(arc)
    assert sorted(z.files) == sorted(str(n)[:-len(".tiff")] for n in g["names"])
    for i, name in enumerate(g["names"]):
        a = z[str(name)[:-len(".tiff")]]
        assert a.dtype == np.float32 and a.shape == tuple(g["shapes"][i]) and np.array_equal(a[:4], g["heads"][i])
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(g["sha256"][i])
        assert abs(float(a.astype(np.float64).sum()) - float(g["sums"][i])) < 1e-9
    A = P.load_pst(arc, SIZES, TIFF_INDEX)
    for ti in TIFF_INDEX:
        cls, num, slot = P.pst_slot(ti)
        assert np.array_equal(A[cls][slot], z[f"pst_{SIZES[cls]}_{num}"])
    # resolution order (round 6): RFX_PST_PATH must exist when set; a configured path that exists is used; one that does not is
    # skipped WITH a warning and nothing else is searched (the tracker then generates templates, or raises: RO.PST_fallback)
    old = os.environ.pop("RFX_PST_PATH", None)
    try:
        with pytest.warns(UserWarning, match="does not exist"):
            assert P.resolve_pst_source("PFO/fps_uniform_sphere_that_does_not_exist") is None
        assert P.resolve_pst_source(None) is None
        assert P.resolve_pst_source(arc) == arc
        assert P.resolve_pst_source(HERE) == HERE
        os.environ["RFX_PST_PATH"] = arc
        assert P.resolve_pst_source("PFO/fps_uniform_sphere_that_does_not_exist") == arc
        os.environ["RFX_PST_PATH"] = os.path.join(HERE, "no_such_templates.npz")
        with pytest.raises(FileNotFoundError, match="RFX_PST_PATH"):
            P.resolve_pst_source(arc)
    finally:
        os.environ.pop("RFX_PST_PATH", None)
        if old is not None:
            os.environ["RFX_PST_PATH"] = old
    d = _ref_dir()
    if d is not None:         # the authoring container: directory and archive give the same container, bit for bit
        B = P.load_pst(d, SIZES, TIFF_INDEX)
        assert all(np.array_equal(A[c], B[c]) for c in range(3))
