"""Pre-sampled search templates ("PST") of the random-optimisation tracker.

The reference ships 60 float32 TIFFs under ``PFO/fps_uniform_sphere`` (``pst_{10240,3072,1024}_{0..19}.tiff``,
each [P, 6]: row 0 is the null perturbation, the rest are farthest-point samples of the 6-D unit ball) and reads
them with ``cv2.imread(path, -1)`` into ``ALL_PST[class][index]`` (reference model/ROtracker.py:834-866).
``load_pst`` builds the same container from the same files (``RO.PST_path`` exactly as in the reference YAMLs: a user of the
reference has that directory) or from an ``.npz`` archive of the same 60 arrays.  The package itself ships NO template data
(round 6; the reference repository carries no licence file, so its data is not redistributed as part of the product): the
repository's test suite keeps one such archive as a FIXTURE (``tests/golden/pst_templates.npz``, provenance in
``tests/golden/README.md``) and points ``RFX_PST_PATH`` at it (``tests/conftest.py``).

``read_float_tiff`` is a baseline-TIFF reader for exactly that file kind (uncompressed, one float32 sample per
pixel, strips): neither cv2 nor Pillow is needed.  ``make_pst`` -- seeded templates of the same structure (origin first, then
points of the 6-D unit ball, farthest first) -- is what the tracker searches with when no template directory / archive is
configured (``RO.PST_fallback: "generated"``, the default): it runs, with a warning, but a tracker that searches with
different particles does not retrace the reference's poses.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Sequence

import numpy as np

_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 8: "h", 9: "i", 16: "Q"}


def read_float_tiff(path: str) -> np.ndarray:
    """[rows, cols] float32 array of an uncompressed single-channel 32-bit-float TIFF (classic or BigTIFF)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 8 or buf[:2] not in (b"II", b"MM"):
        raise ValueError(f"{path}: not a TIFF file")
    bo = "<" if buf[:2] == b"II" else ">"
    magic = struct.unpack_from(bo + "H", buf, 2)[0]
    if magic == 42:
        big, off = False, struct.unpack_from(bo + "I", buf, 4)[0]
    elif magic == 43:
        big, off = True, struct.unpack_from(bo + "Q", buf, 8)[0]
    else:
        raise ValueError(f"{path}: bad TIFF magic {magic}")
    n = struct.unpack_from(bo + ("Q" if big else "H"), buf, off)[0]
    off += 8 if big else 2
    esz, inl = (20, 8) if big else (12, 4)
    tags: Dict[int, Sequence[int]] = {}
    for i in range(n):
        e = off + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        cnt = struct.unpack_from(bo + ("Q" if big else "I"), buf, e + 4)[0]
        if typ not in _TYPE_FMT:
            continue
        nbytes = _TYPE_SIZE[typ] * cnt
        vo = e + (12 if big else 8)
        if nbytes > inl:
            vo = struct.unpack_from(bo + ("Q" if big else "I"), buf, vo)[0]
        tags[tag] = struct.unpack_from(bo + _TYPE_FMT[typ] * cnt, buf, vo)
    try:
        width, height = tags[256][0], tags[257][0]
        offsets = tags[273]
    except KeyError as exc:
        raise ValueError(f"{path}: missing TIFF tag {exc}") from None
    if tags.get(259, (1,))[0] != 1:
        raise ValueError(f"{path}: compressed TIFFs are not supported (reference PST files are raw)")
    if tags.get(258, (1,))[0] != 32 or tags.get(339, (1,))[0] != 3 or tags.get(277, (1,))[0] != 1:
        raise ValueError(f"{path}: expected one 32-bit float sample per pixel")
    rps = min(tags.get(278, (height,))[0], height)
    counts = tags.get(279)
    out = np.empty((height, width), np.float32)
    row = 0
    for s, so in enumerate(offsets):
        rows = min(rps, height - row)
        nb = rows * width * 4
        if counts is not None and counts[s] < nb:
            raise ValueError(f"{path}: short strip {s}")
        if so + nb > len(buf):
            raise ValueError(f"{path}: strip {s} runs past the end of the file")
        out[row:row + rows] = np.frombuffer(buf, dtype=bo + "f4", count=rows * width, offset=so).reshape(rows, width)
        row += rows
    if row != height:
        raise ValueError(f"{path}: {row} of {height} rows present")
    return out


def write_float_tiff(path: str, a: np.ndarray) -> None:
    """inverse of read_float_tiff (little-endian classic TIFF, one strip): for tests and for exporting templates."""
    a = np.ascontiguousarray(a, np.float32)
    h, w = a.shape
    entries = [(256, 4, w), (257, 4, h), (258, 3, 32), (259, 3, 1), (262, 3, 1), (273, 4, 8 + 2 + 10 * 12 + 4),
               (277, 3, 1), (278, 4, h), (279, 4, a.nbytes), (339, 3, 3)]
    with open(path, "wb") as fh:
        fh.write(b"II" + struct.pack("<HI", 42, 8) + struct.pack("<H", len(entries)))
        for tag, typ, val in entries:
            fh.write(struct.pack("<HHI", tag, typ, 1) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val)))
        fh.write(struct.pack("<I", 0) + a.tobytes())


def make_pst(n: int, seed: int) -> np.ndarray:
    """[n,6] generated template: origin first, then points uniform in the 6-D unit ball, farthest first.
    NOT the reference's particles -- explicit fallback only (see module docstring)."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n - 1, 6))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    r = rng.uniform(0.0, 1.0, (n - 1, 1)) ** (1.0 / 6.0)
    pts = g * r
    pts = pts[np.argsort(-np.linalg.norm(pts, axis=1))]
    return np.concatenate([np.zeros((1, 6)), pts], 0).astype(np.float32)


def pst_slot(tiff_index: int):
    """(class, file number, slot in ALL_PST[class]) of one entry of ROTracker.tiff_index (reference :854-858)."""
    cls = tiff_index // 20
    num = tiff_index - cls * 20
    return cls, num, num // 3


def _empty(tiff_index: Sequence[int], PST_size: Sequence[int]) -> Dict[int, np.ndarray]:
    n = len(tiff_index)           # container shapes of the reference (:847-850)
    return {0: np.zeros((n // 3 + 1, PST_size[0], 6), np.float32), 1: np.zeros((n // 3 + 1, PST_size[1], 6), np.float32),
            2: np.zeros((n // 3, PST_size[2], 6), np.float32)}


def resolve_pst_source(PST_path):
    """Where the templates come from: ``RFX_PST_PATH`` when set (it MUST then name a directory of the reference's TIFFs or an
    ``.npz`` archive of them: a typo or an unmounted directory raises instead of silently searching with other particles), else
    ``RO.PST_path`` when it exists (a configured path that does not exist is skipped WITH a warning); ``None`` when neither
    names templates (the caller then generates them, or raises: ``RO.PST_fallback``)."""
    import warnings

    def usable(c):
        return bool(c) and (os.path.isdir(c) or (os.path.isfile(c) and c.endswith(".npz")))

    env = os.environ.get("RFX_PST_PATH")
    if env:
        if not usable(env):
            raise FileNotFoundError(f"RFX_PST_PATH={env!r} is neither a directory of PST TIFFs nor an .npz archive of them")
        return env
    if usable(PST_path):
        return PST_path
    if PST_path:
        warnings.warn(f"RO.PST_path {PST_path!r} does not exist: the tracker's particle templates are not the configured ones", stacklevel=3)
    return None


def load_pst(PST_path: str, PST_size: Sequence[int], tiff_index: Sequence[int]) -> Dict[int, np.ndarray]:
    """ALL_PST as the reference's readpst builds it (model/ROtracker.py:834-866) from ``pst_{size}_{num}.tiff`` in the
    directory ``PST_path``, or from the arrays ``pst_{size}_{num}`` of the ``.npz`` archive ``PST_path``."""
    archive = None
    if os.path.isfile(PST_path) and PST_path.endswith(".npz"):
        archive = np.load(PST_path)
    elif not os.path.isdir(PST_path):
        raise FileNotFoundError(
            f"RO.PST_path {PST_path!r} is neither a directory of the reference's 60 float32 TIFFs (PFO/fps_uniform_sphere) "
            "nor an .npz archive of them.  Point RO.PST_path / RFX_PST_PATH at the reference's directory, or set RO.PST_fallback: "
            "'generated' to search with seeded templates instead (poses will then differ from the reference's).")
    out = _empty(tiff_index, PST_size)
    for ti in tiff_index:
        cls, num, slot = pst_slot(ti)
        name = f"pst_{PST_size[cls]}_{num}"
        if archive is not None:
            if name not in archive:
                raise ValueError(f"{PST_path}: no array {name!r}")
            a = np.ascontiguousarray(archive[name], np.float32)
            path = f"{PST_path}:{name}"
        else:
            path = os.path.join(PST_path, name + ".tiff")
            a = read_float_tiff(path)
        if a.shape != (PST_size[cls], 6):
            raise ValueError(f"{path}: shape {a.shape}, expected {(PST_size[cls], 6)}")
        out[cls][slot] = a
    return out


def generated_pst(seed: int, PST_size: Sequence[int], tiff_index: Sequence[int]) -> Dict[int, np.ndarray]:
    out = _empty(tiff_index, PST_size)
    for ti in tiff_index:
        cls, _, slot = pst_slot(ti)
        out[cls][slot] = make_pst(PST_size[cls], seed + 97 * ti)
    return out
