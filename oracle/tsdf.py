"""ctypes front-end of oracle/tsdf_oracle.c -- TEST INFRASTRUCTURE ONLY.

``load(fma=True)`` returns a handle to ``_build/liborc_{fma,nofma}.so`` (built by
``make -C oracle``).  All arrays are C-contiguous float32 numpy arrays modified in place.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


class Counts(C.Structure):
    _fields_ = [("updated", C.c_int64), ("colour", C.c_int64)]


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


class TsdfOracle:
    def __init__(self, fma: bool = True):
        path = os.path.join(_HERE, "_build", "liborc_fma.so" if fma else "liborc_nofma.so")
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        L.orc_mv_integrate.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_int, _F, C.c_float, _F, _F, _F, _F,
                                       C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _F,
                                       C.c_int, C.POINTER(Counts)]
        L.orc_mv_integrate_range.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_int, _F, C.c_float, _F, _F, _F, _F,
                                             C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _F,
                                             C.c_int, C.c_int64, C.c_int64, C.POINTER(Counts)]
        L.orc_mv_shift.argtypes = [_F, _F, _F, _F, _F, _F, C.c_int, C.c_int, C.c_int, _F,
                                   C.c_int, C.c_int, C.c_int, _F, C.c_float, C.c_int]
        L.orc_mv_trilerp.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_int, _F, C.c_float, _F, C.c_int64, _F]
        L.orc_mv_filter.argtypes = [_F, _F, _F, C.c_int64, C.c_float]
        L.orc_mv_truncated_pc.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, _F, C.c_float, C.c_float,
                                          C.c_int, C.c_float, _F, C.c_int]
        L.orc_mv_truncated_pc.restype = C.c_int64
        L.orc_mv_fill.argtypes = [_F, _F, _F, C.c_int64]
        L.orc_mv_copy.argtypes = [_F, _F, _F, _F, _F, _F, C.c_int64]
        L.orc_gbv_integrate.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_float, _F, _F, _F, _F, _F,
                                        C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(Counts)]
        L.orc_gbv_clear.argtypes = [_F, C.c_int64]
        L.orc_tr_vertex.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _F, _F]
        L.orc_tr_normal.argtypes = [_F, _F, C.c_int, C.c_int]
        L.orc_tr_evaluate.argtypes = [_F, C.c_int, C.c_int, C.c_int, _F, C.c_float, _F, _F, _F, _F, _F, _F, C.c_int, _F,
                                      C.c_int, C.c_int, C.c_int, C.c_int, _F, _F, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
        L.orc_fma_mode.restype = C.c_int
        assert L.orc_fma_mode() == int(fma)

    # -- MV ----------------------------------------------------------------------------------
    def mv_integrate(self, tsdf, weight, color, dims, origin, voxel, K, c2w, color_packed, depth,
                     trunc, obs_weight=1.0, weight_clamp=1.0, reintegrate=0.0, old_bnd=None,
                     decode="reference") -> Tuple[int, int]:
        H, W = depth.shape
        ob = np.zeros(6, np.float32) if old_bnd is None else np.ascontiguousarray(old_bnd, np.float32).reshape(-1)
        cnt = Counts()
        self.lib.orc_mv_integrate(tsdf, weight, color, int(dims[0]), int(dims[1]), int(dims[2]),
                                  np.ascontiguousarray(origin, np.float32), float(voxel),
                                  np.ascontiguousarray(K, np.float32).reshape(-1),
                                  np.ascontiguousarray(c2w, np.float32).reshape(-1),
                                  np.ascontiguousarray(color_packed, np.float32).reshape(-1),
                                  np.ascontiguousarray(depth, np.float32).reshape(-1), H, W,
                                  float(trunc), float(obs_weight), float(weight_clamp), float(reintegrate),
                                  ob, 0 if decode == "reference" else 1, C.byref(cnt))
        return cnt.updated, cnt.colour

    def mv_integrate_threads(self, tsdf, weight, color, dims, origin, voxel, K, c2w, color_packed, depth, trunc,
                             threads: int, obs_weight=1.0, weight_clamp=1.0, decode="reference") -> Tuple[int, int]:
        """the same sweep with the voxel index range cut into ``threads`` pieces run by that many host threads (the C call
        releases the GIL; voxels are independent, so the result is the single-threaded one).  bench.py's cpu_baseline."""
        from concurrent.futures import ThreadPoolExecutor
        H, W = depth.shape
        n = int(dims[0]) * int(dims[1]) * int(dims[2])
        args = (np.ascontiguousarray(origin, np.float32), float(voxel), np.ascontiguousarray(K, np.float32).reshape(-1),
                np.ascontiguousarray(c2w, np.float32).reshape(-1), np.ascontiguousarray(color_packed, np.float32).reshape(-1),
                np.ascontiguousarray(depth, np.float32).reshape(-1), H, W, float(trunc), float(obs_weight), float(weight_clamp), 0.0,
                np.zeros(6, np.float32), 0 if decode == "reference" else 1)
        cuts = [n * k // threads for k in range(threads + 1)]

        def piece(k):
            cnt = Counts()
            self.lib.orc_mv_integrate_range(tsdf, weight, color, int(dims[0]), int(dims[1]), int(dims[2]), *args, cuts[k], cuts[k + 1],
                                            C.byref(cnt))
            return cnt.updated, cnt.colour

        with ThreadPoolExecutor(max_workers=threads) as ex:
            res = list(ex.map(piece, range(threads)))
        return sum(r[0] for r in res), sum(r[1] for r in res)

    def mv_shift(self, dst3, src3, dims, origin, odims, old_origin, voxel, decode="reference"):
        self.lib.orc_mv_shift(dst3[0], dst3[1], dst3[2], src3[0], src3[1], src3[2],
                              int(dims[0]), int(dims[1]), int(dims[2]), np.ascontiguousarray(origin, np.float32),
                              int(odims[0]), int(odims[1]), int(odims[2]),
                              np.ascontiguousarray(old_origin, np.float32), float(voxel),
                              0 if decode == "reference" else 1)

    def mv_trilerp(self, tsdf, weight, color, dims, origin, voxel, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        out = np.zeros((pts.shape[0], 5), np.float32)
        self.lib.orc_mv_trilerp(tsdf, weight, color, int(dims[0]), int(dims[1]), int(dims[2]),
                                np.ascontiguousarray(origin, np.float32), float(voxel), pts, pts.shape[0], out)
        return out

    def mv_filter(self, tsdf, weight, color, thr):
        self.lib.orc_mv_filter(tsdf, weight, color, tsdf.size, float(thr))

    def mv_truncated_pc(self, tsdf, color, dims, origin, voxel, trunc, pc_num, trunc_tsdf=0.5, decode="reference"):
        pc = np.zeros((pc_num, 7), np.float32)
        n = self.lib.orc_mv_truncated_pc(tsdf, color, int(dims[0]), int(dims[1]), int(dims[2]),
                                         np.ascontiguousarray(origin, np.float32), float(voxel), float(trunc),
                                         int(pc_num), float(trunc_tsdf), pc, 0 if decode == "reference" else 1)
        return pc, n

    def mv_fill(self, tsdf, weight, color):
        self.lib.orc_mv_fill(tsdf, weight, color, tsdf.size)

    def mv_copy(self, src3, dst3):
        self.lib.orc_mv_copy(src3[0], src3[1], src3[2], dst3[0], dst3[1], dst3[2], src3[0].size)

    # -- tracker -----------------------------------------------------------------------------
    def tr_vertex(self, depth, K, cut_dist, trunc, sample_range, u_rows):
        H, W = depth.shape
        out = np.zeros((H * W, 4), np.float32)
        u = np.ascontiguousarray(u_rows, np.float32)
        self.lib.orc_tr_vertex(np.ascontiguousarray(depth, np.float32).reshape(-1), out,
                               np.ascontiguousarray(K, np.float32).reshape(-1), H, W, float(cut_dist), float(trunc),
                               float(sample_range), np.ascontiguousarray(u[:, 0]), np.ascontiguousarray(u[:, 1]))
        return out

    def tr_normal(self, vertex4, H, W):
        out = np.zeros((H * W, 3), np.float32)
        self.lib.orc_tr_normal(np.ascontiguousarray(vertex4, np.float32), out, H, W)
        return out

    def tr_evaluate(self, tsdf, dims, origin, voxel, vertex4, normal3, R, T, q6, ss, K, H, W, level, level_index):
        """(float32 running sums in pixel order, hit counts, the sums in 2^-30 fixed point: order-independent)"""
        P = q6.shape[0]
        val, cnt, q30 = np.zeros(P, np.float32), np.zeros(P, np.float32), np.zeros(P, np.int64)
        self.lib.orc_tr_evaluate(tsdf, int(dims[0]), int(dims[1]), int(dims[2]), np.ascontiguousarray(origin, np.float32),
                                 float(voxel), np.ascontiguousarray(vertex4, np.float32), np.ascontiguousarray(normal3, np.float32),
                                 np.ascontiguousarray(R, np.float32).reshape(-1), np.ascontiguousarray(T, np.float32),
                                 np.ascontiguousarray(q6, np.float32), np.ascontiguousarray(ss, np.float32), P,
                                 np.ascontiguousarray(K, np.float32).reshape(-1), H, W, int(level), int(level_index), val, cnt, q30)
        return val, cnt, q30

    # -- GBV ---------------------------------------------------------------------------------
    def gbv_integrate(self, trgb, w, res, box, K, c2w, rgb01, depth, trunc, obs_weight=1.0,
                      decode="reference") -> int:
        H, W = depth.shape
        r = (res, res, res) if np.isscalar(res) else res
        cnt = Counts()
        self.lib.orc_gbv_integrate(trgb, w, int(r[0]), int(r[1]), int(r[2]), float(1.0 / r[0]),
                                   np.ascontiguousarray(box, np.float32).reshape(-1),
                                   np.ascontiguousarray(K, np.float32).reshape(-1),
                                   np.ascontiguousarray(c2w, np.float32).reshape(-1),
                                   np.ascontiguousarray(rgb01, np.float32).reshape(-1),
                                   np.ascontiguousarray(depth, np.float32).reshape(-1), H, W,
                                   float(trunc), float(obs_weight), 0 if decode == "reference" else 1,
                                   C.byref(cnt))
        return cnt.updated

    def gbv_clear(self, trgb):
        self.lib.orc_gbv_clear(trgb, trgb.size // 4)


_CACHE = {}


def load(fma: bool = True) -> TsdfOracle:
    if fma not in _CACHE:
        _CACHE[fma] = TsdfOracle(fma)
    return _CACHE[fma]


def pack_color(rgb255: np.ndarray) -> np.ndarray:
    """host packing of model/Volume.py:725-728: floor(B*65536 + G*256 + R), fp32."""
    c = rgb255.astype(np.float32)
    return np.floor(c[..., 2] * (256 * 256) + c[..., 1] * 256 + c[..., 0]).astype(np.float32)
