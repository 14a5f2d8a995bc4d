"""oracle/draws_oracle.py against the known-answer vectors of Philox4x32-10 (Random123 kat_vectors), and the properties the
library's uniforms need (range, resolution, independence of the streams)."""
import numpy as np

from oracle import draws_oracle as DO

KAT = [  # counter, key, expected output
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_philox_known_answers():
    for counter, key, want in KAT:
        got = DO.philox4x32_10(counter, key)
        assert tuple(int(x) for x in got) == want


def test_uniform_draws_properties():
    u = DO.uniform_draws(0x1234567887654321, 0, 200000)
    assert u.dtype == np.float32 and float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert np.all(u * np.float32(2 ** 24) == np.floor(u * np.float32(2 ** 24)))          # 24-bit lattice
    assert abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1.0 / 12.0) < 2e-3
    h = np.histogram(u, bins=64, range=(0.0, 1.0))[0]
    assert h.min() > 0.9 * len(u) / 64 and h.max() < 1.1 * len(u) / 64
    v = DO.uniform_draws(0x1234567887654321, 1, 200000)                                # another stream: unrelated numbers
    assert abs(float(np.corrcoef(u, v)[0, 1])) < 1e-2
    assert abs(float(np.corrcoef(u[:-1], u[1:])[0, 1])) < 1e-2
    # element e depends on (seed, stream, e) only: a prefix of a longer draw
    assert np.array_equal(DO.uniform_draws(7, 1, 6), DO.uniform_draws(7, 1, 100)[:6])
