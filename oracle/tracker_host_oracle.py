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

Two modes.  mode="numpy1" (default; what the product follows): the types above.  mode="numpy2": the types numpy >= 2 gives the
very same expressions (a Python scalar is "weak": float32 (op) Python float stays float32, so EVERYTHING is float32, the running
sums included).  The second mode exists to pin the restatement: tests/golden/tracker_host.npz holds inputs and outputs of the
REFERENCE's own `cal_transform`, `update_PST` and `random_optimization` run in this container (numpy 2.2; PyCUDA and cv2 stubbed,
the fitness arrays injected in place of the CUDA evaluation: tests/golden/make_golden.py), and mode="numpy2" reproduces them bit
for bit (tests/test_golden.py) -- selection order, weights, formulas and bookkeeping are the reference's; the two modes differ only
in the casts spelled out below.

Parity status: structure PINNED against the reference's own code run here (numpy-2 arithmetic); the numpy-1.21 arithmetic the
product follows is pinned by construction (the documented promotion rules) + closed forms (tests/test_oracle_tracker_host.py).
"""
import math

import numpy as np

f32 = np.float32


class _Types:
    """up(x): what `float32 (op) Python scalar` gives -- float64 under numpy 1.21, float32 under numpy 2; sqrt likewise"""

    def __init__(self, mode):
        if mode not in ("numpy1", "numpy2"):
            raise ValueError(mode)
        self.two = mode == "numpy2"

    def up(self, x):
        return f32(x) if self.two else float(x)

    def sqrt(self, x):
        return np.sqrt(f32(x)) if self.two else math.sqrt(x)

    def zero(self):
        return 0.0                      # `sum = 0.0`: a Python float either way (weak under numpy 2: the first += makes it float32)


def cal_transform(search_value, transform_candidate, search_size, count_search_max, mode="numpy1"):
    """reference :606-709.  search_value float32 [n], transform_candidate float32 [n,6], search_size float32 [6].
    Returns (success, min_tsdf, mean_transform float32 [7], invalid) -- `invalid` where the reference prints and exits (:662-669)."""
    T = _Types(mode)
    up = T.up
    mean_transform = np.zeros(7, dtype=np.float32)
    origin_tsdf = f32(search_value[0])
    sum_tx = sum_ty = sum_tz = sum_qw = sum_qx = sum_qy = sum_qz = sum_weight = sum_tsdf = T.zero()
    count_search = 0
    for j in range(1, len(search_value)):
        if f32(search_value[j]) < origin_tsdf:
            tx, ty, tz, qx, qy, qz = (f32(v) for v in transform_candidate[j])
            cur_fit = f32(search_value[j])
            weight = f32(origin_tsdf - cur_fit)                       # float32 - float32
            sum_tx = sum_tx + up(f32(tx * weight))                    # float32 product; the running sum is float64 (numpy 1) / float32 (numpy 2)
            sum_ty = sum_ty + up(f32(ty * weight))
            sum_tz = sum_tz + up(f32(tz * weight))
            sum_qx = sum_qx + up(f32(qx * weight))
            sum_qy = sum_qy + up(f32(qy * weight))
            sum_qz = sum_qz + up(f32(qz * weight))
            qx = f32(qx * f32(search_size[3]))
            qy = f32(qy * f32(search_size[4]))
            qz = f32(qz * f32(search_size[5]))
            rad = up(1) - up(f32(qx * qx)) - up(f32(qy * qy)) - up(f32(qz * qz))      # int - float32
            if rad < 0:
                return False, origin_tsdf, mean_transform, True
            qw = T.sqrt(rad)
            sum_qw = sum_qw + qw * up(weight)
            sum_weight = sum_weight + up(weight)
            sum_tsdf = sum_tsdf + up(f32(cur_fit * weight))
            count_search += 1
            if count_search == count_search_max:
                break
    if count_search <= 0:
        return False, origin_tsdf, mean_transform, False
    mean_tsdf = sum_tsdf / sum_weight
    mean_transform[0] = (sum_tx / sum_weight) * up(search_size[0])
    mean_transform[1] = (sum_ty / sum_weight) * up(search_size[1])
    mean_transform[2] = (sum_tz / sum_weight) * up(search_size[2])
    qww = sum_qw / sum_weight
    qxx = (sum_qx / sum_weight) * up(search_size[3])
    qyy = (sum_qy / sum_weight) * up(search_size[4])
    qzz = (sum_qz / sum_weight) * up(search_size[5])
    lens = up(1) / T.sqrt(qww * qww + qxx * qxx + qyy * qyy + qzz * qzz)
    mean_transform[3] = qww * lens
    mean_transform[4] = qxx * lens
    mean_transform[5] = qyy * lens
    mean_transform[6] = qzz * lens
    return True, mean_tsdf, mean_transform, False


def update_PST(search_size, tsdf, mean_transform, min_scale=1e-3, scale=0.09, mode="numpy1"):
    """reference :493-534, in place on search_size (float32 [6]).  tsdf: a success's mean or a failure's origin value (float32);
    numpy 1.21: everything float64 (`float32 + Python float`); numpy 2: a float32 tsdf keeps every product float32, a float64
    one (numpy 1's mean) makes them float64 -- here tsdf has the type cal_transform of the same mode returned."""
    T = _Types(mode)
    up = T.up
    ms, sc = up(min_scale), up(scale)
    s_tx = abs(up(mean_transform[0])) + ms
    s_ty = abs(up(mean_transform[1])) + ms
    s_tz = abs(up(mean_transform[2])) + ms
    s_qx = abs(up(mean_transform[4])) + ms
    s_qy = abs(up(mean_transform[5])) + ms
    s_qz = abs(up(mean_transform[6])) + ms
    trans_norm = T.sqrt(s_tx * s_tx + s_ty * s_ty + s_tz * s_tz + s_qx * s_qx + s_qy * s_qy + s_qz * s_qz)
    t = up(tsdf)
    search_size[3] = sc * t * (s_qx / trans_norm) + ms
    search_size[4] = sc * t * (s_qy / trans_norm) + ms
    search_size[5] = sc * t * (s_qz / trans_norm) + ms
    search_size[0] = sc * t * (s_tx / trans_norm) + ms
    search_size[1] = sc * t * (s_ty / trans_norm) + ms
    search_size[2] = sc * t * (s_tz / trans_norm) + ms


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
                iterative_scale, beta=0.9, mode="numpy1"):
    """one iteration of the loop after `evaluate_tsdf` (reference :757-826).  Returns `invalid` (the reference's exit)."""
    T = _Types(mode)
    success, min_tsdf, mt, invalid = cal_transform(search_value, transform_candidate, st.search_size, count_search, mode)
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
    update_PST(st.search_size, min_tsdf, mt, scale=scaling_coefficient, mode=mode)
    if st.previous_success and success:
        for k in range(6):                                        # Python float * float32 scalar: float64 (numpy 1) / float32 (numpy 2)
            st.search_size[k] = T.up(beta) * T.up(st.search_size[k]) + T.up(1 - beta) * T.up(st.previous_search_size[k])
    elif success:
        if iterative_scale:
            st.previous_success = True
        st.previous_search_size[:] = st.search_size
    if not success:
        st.previous_success = False
    if i == 0:
        st.first_success = bool(success)
    return False
