"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the counter-based uniforms a bundle-adjustment iteration may draw for
itself (``rfx_ba_desc.seed_u`` / ``rfx_uniform_draws``, include/rfx.h).

The reference draws the sampler jitter and the TV-lattice offset with ``torch.rand`` (model/scene_rep.py:437,
mp_slam/slam.py:198-203): any stream of independent uniforms on [0, 1) is the same algorithm.  The library's stream is
Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator behind torch's own
CUDA/HIP draws): element e of stream s under a 64-bit seed is word 0 of the block with counter (e_lo, e_hi, s, 0) and key
(seed_lo, seed_hi), mapped to a float with 24 random bits, ``(word >> 8) * 2**-24``.  Pinned by the three known-answer
vectors of the Random123 distribution (tests/test_oracle_draws.py).  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK, _S32 = np.uint64(0xFFFFFFFF), np.uint64(32)


def philox4x32_10(counter, key):
    """counter: four uint32 arrays (or scalars) of one shape, key: two uint32 scalars -> four uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) for x in counter]
    k = [np.uint64(key[0]), np.uint64(key[1])]
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [(p1 >> _S32) ^ c[1] ^ k[0], p1 & _MASK, (p0 >> _S32) ^ c[3] ^ k[1], p0 & _MASK]
        k = [(k[0] + _W0) & _MASK, (k[1] + _W1) & _MASK]
    return [x.astype(np.uint32) for x in c]


def uniform_draws(seed: int, stream: int, n: int) -> np.ndarray:
    """the float32 [n] buffer rfx_uniform_draws(seed, stream, n) fills"""
    e = np.arange(n, dtype=np.uint64)
    w0 = philox4x32_10([e & _MASK, e >> _S32, np.full(n, stream, np.uint64), np.zeros(n, np.uint64)],
                       [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])[0]
    return ((w0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
