"""TEST INFRASTRUCTURE ONLY -- not part of the product (only tests/ may import this).

CPU restatement, as literal scalar Python loops, of the HOST logic of the reference's pose search
(/root/reference/model/ROtracker.py): `cal_transform` :606-709, `update_PST` :493-534 and the per-iteration bookkeeping of
`random_optimization` :745-826.  The product runs this logic vectorised on the host (remixfusion_amd/model/ROtracker.py) and,
by default, in one block on the device (rfx_track_search_update); both are compared with this file.

Arithmetic types.  The reference mixes numpy float32 scalars (template entries, fitness values, search sizes) with Python
floats; what each expression rounds to depends on numpy's scalar promotion rules, and the reference pins numpy==1.21.6
(requirements.txt:6), where scalar (op) scalar promotes like arrays of those types do: float32 (op) float32 -> float32,
float32 (op) Python float / int -> float64.  (numpy >= 2 keeps float32 there; this container has 2.2.)  So that the restatement
does not depend on the numpy that runs it, every operation below states its type with an explicit cast: `f32(...)` where 1.21
computes in float32, plain Python floats (= float64) elsewhere.

Parity status: PINNED ONLY BY CONSTRUCTION (line-by-line restatement + closed-form checks in tests/test_oracle_tracker_host.py);
the reference ships no vectors for the tracker and PyCUDA cannot run here.
"""
import math

import numpy as np

f32 = np.float32


def cal_transform(search_value, transform_candidate, search_size, count_search_max):
    """reference :606-709.  search_value float32 [n], transform_candidate float32 [n,6], search_size float32 [6].
    Returns (success, min_tsdf, mean_transform float32 [7], invalid) -- `invalid` where the reference prints and exits (:662-669)."""
    mean_transform = np.zeros(7, dtype=np.float32)
    origin_tsdf = f32(search_value[0])
    sum_tx = sum_ty = sum_tz = sum_qw = sum_qx = sum_qy = sum_qz = sum_weight = sum_tsdf = 0.0
    count_search = 0
    for j in range(1, len(search_value)):
        if f32(search_value[j]) < origin_tsdf:
            tx, ty, tz, qx, qy, qz = (f32(v) for v in transform_candidate[j])
            cur_fit = f32(search_value[j])
            weight = f32(origin_tsdf - cur_fit)                       # float32 - float32
            sum_tx += float(f32(tx * weight))                         # float32 product, float64 running sum
            sum_ty += float(f32(ty * weight))
            sum_tz += float(f32(tz * weight))
            sum_qx += float(f32(qx * weight))
            sum_qy += float(f32(qy * weight))
            sum_qz += float(f32(qz * weight))
            qx = f32(qx * f32(search_size[3]))
            qy = f32(qy * f32(search_size[4]))
            qz = f32(qz * f32(search_size[5]))
            rad = 1 - float(f32(qx * qx)) - float(f32(qy * qy)) - float(f32(qz * qz))     # int - float32 -> float64
            if rad < 0:
                return False, origin_tsdf, mean_transform, True
            qw = math.sqrt(rad)
            sum_qw += qw * float(weight)
            sum_weight += float(weight)
            sum_tsdf += float(f32(cur_fit * weight))
            count_search += 1
            if count_search == count_search_max:
                break
    if count_search <= 0:
        return False, origin_tsdf, mean_transform, False
    mean_tsdf = sum_tsdf / sum_weight
    mean_transform[0] = (sum_tx / sum_weight) * float(search_size[0])
    mean_transform[1] = (sum_ty / sum_weight) * float(search_size[1])
    mean_transform[2] = (sum_tz / sum_weight) * float(search_size[2])
    qww = sum_qw / sum_weight
    qxx = (sum_qx / sum_weight) * float(search_size[3])
    qyy = (sum_qy / sum_weight) * float(search_size[4])
    qzz = (sum_qz / sum_weight) * float(search_size[5])
    lens = 1 / math.sqrt(qww * qww + qxx * qxx + qyy * qyy + qzz * qzz)
    mean_transform[3] = qww * lens
    mean_transform[4] = qxx * lens
    mean_transform[5] = qyy * lens
    mean_transform[6] = qzz * lens
    return True, mean_tsdf, mean_transform, False


def update_PST(search_size, tsdf, mean_transform, min_scale=1e-3, scale=0.09):
    """reference :493-534, in place on search_size (float32 [6]).  tsdf: float64 (a success's mean) or float32 (a failure's origin
    value) -- either way `scale * tsdf` is a float64 product under numpy 1.21."""
    s_tx = abs(float(mean_transform[0])) + min_scale
    s_ty = abs(float(mean_transform[1])) + min_scale
    s_tz = abs(float(mean_transform[2])) + min_scale
    s_qx = abs(float(mean_transform[4])) + min_scale
    s_qy = abs(float(mean_transform[5])) + min_scale
    s_qz = abs(float(mean_transform[6])) + min_scale
    trans_norm = math.sqrt(s_tx ** 2 + s_ty ** 2 + s_tz ** 2 + s_qx ** 2 + s_qy ** 2 + s_qz ** 2)
    t = float(tsdf)
    search_size[3] = scale * t * (s_qx / trans_norm) + min_scale
    search_size[4] = scale * t * (s_qy / trans_norm) + min_scale
    search_size[5] = scale * t * (s_qz / trans_norm) + min_scale
    search_size[0] = scale * t * (s_tx / trans_norm) + min_scale
    search_size[1] = scale * t * (s_ty / trans_norm) + min_scale
    search_size[2] = scale * t * (s_tz / trans_norm) + min_scale


class SearchState:
    """the variables `random_optimization` carries from one iteration to the next (reference :724-742)"""

    def __init__(self, R, T, search_size):
        self.R = np.array(R, dtype=np.float32).reshape(3, 3)
        self.T = np.array(T, dtype=np.float32).reshape(3)
        self.search_size = np.array(search_size, dtype=np.float32).reshape(6)
        self.previous_search_size = np.zeros(6, dtype=np.float32)
        self.previous_success = False
        self.success = False
        self.count_particle = 0
        self.level_index = 5
        self.first_success = False
        self.min_tsdf = None

    def template(self):
        """`if not success: count_particle = 0` at the top of an iteration (:745-746); returns the template index to evaluate"""
        if not self.success:
            self.count_particle = 0
        return self.count_particle


def search_step(st, i, search_value, transform_candidate, depth_level, count_search, scaling_coefficient, fix_level_index,
                iterative_scale, beta=0.9):
    """one iteration of the loop after `evaluate_tsdf` (reference :757-826).  Returns `invalid` (the reference's exit)."""
    success, min_tsdf, mt, invalid = cal_transform(search_value, transform_candidate, st.search_size, count_search)
    if invalid:
        return True
    st.success, st.min_tsdf = success, min_tsdf
    qw, qx, qy, qz = (f32(v) for v in mt[3:7])
    if success:
        if st.count_particle < 19:
            st.count_particle += 1
        # float32 products and sums; `2 * (...)` and `1 - ...` are float64 under numpy 1.21 and the array constructor rounds the
        # result to float32 once: the same float32 a float32 evaluation gives (2 x is exact, one rounding either way)
        def e(a, b):
            return f32(float(2 * float(f32(a + b))))

        def d(a):
            return f32(1 - 2 * float(a))
        Rinc = np.array([[d(f32(f32(qy * qy) + f32(qz * qz))), e(f32(qx * qy), -f32(qz * qw)), e(f32(qx * qz), f32(qy * qw))],
                         [e(f32(qx * qy), f32(qz * qw)), d(f32(f32(qx * qx) + f32(qz * qz))), e(f32(qy * qz), -f32(qx * qw))],
                         [e(f32(qx * qz), -f32(qy * qw)), e(f32(qy * qz), f32(qx * qw)), d(f32(f32(qx * qx) + f32(qy * qy)))]],
                        dtype=np.float32)
        st.T = (st.T + mt[:3]).astype(np.float32)
        Rn = np.zeros((3, 3), dtype=np.float32)                   # np.matmul of two float32 3x3: products and sums in float32
        for r in range(3):                                        # (left to right here; a BLAS may fuse or reorder: tests allow 1 ulp)
            for c in range(3):
                Rn[r, c] = f32(f32(f32(Rinc[r, 0] * st.R[0, c]) + f32(Rinc[r, 1] * st.R[1, c])) + f32(Rinc[r, 2] * st.R[2, c]))
        st.R = Rn
    level_index = 1 if fix_level_index else st.level_index + 5
    st.level_index = level_index % depth_level[st.count_particle]
    update_PST(st.search_size, min_tsdf, mt, scale=scaling_coefficient)
    if st.previous_success and success:
        for k in range(6):                                        # Python float * float32 scalar -> float64, stored as float32
            st.search_size[k] = beta * float(st.search_size[k]) + (1 - beta) * float(st.previous_search_size[k])
    elif success:
        if iterative_scale:
            st.previous_success = True
        st.previous_search_size[:] = st.search_size
    if not success:
        st.previous_success = False
    if i == 0:
        st.first_success = bool(success)
    return False
