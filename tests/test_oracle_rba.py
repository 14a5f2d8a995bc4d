"""oracle/rba_oracle.py (kornia 0.6.12's published angle-axis formulas, restated) pinned by properties that do not depend
on the product: scipy's Rotation as an independent implementation of the same maps, orthogonality, closed forms, round
trips, the first-order branch, and the reference's camera-0 gauge."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from oracle import rba_oracle as RO


def _aa(n, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn((n, 3), generator=g, dtype=torch.float64) * scale)


def test_angle_axis_to_rotation_matrix_against_scipy_and_closed_forms():
    aa = _aa(500, 0, 1.2)
    R = RO.angle_axis_to_rotation_matrix(aa)
    ref = torch.from_numpy(Rotation.from_rotvec(aa.numpy()).as_matrix())
    # kornia normalises the axis by theta + 1e-6: a relative 1e-6 / theta in the axis is its published behaviour
    assert float((R - ref).abs().max()) < 5e-6
    eye = torch.eye(3, dtype=torch.float64)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5
    Rz = RO.angle_axis_to_rotation_matrix(torch.tensor([[0.0, 0.0, np.pi / 2]], dtype=torch.float64))[0]
    assert float((Rz - torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)).abs().max()) < 2e-6
    # below theta^2 = 1e-6: the first-order matrix I + [aa]x, exactly
    tiny = torch.tensor([[1e-4, -2e-4, 3e-4], [0.0, 0.0, 0.0]], dtype=torch.float64)
    Rt = RO.angle_axis_to_rotation_matrix(tiny)
    assert torch.equal(Rt[1], eye)
    assert torch.equal(Rt[0], torch.tensor([[1.0, -3e-4, -2e-4], [3e-4, 1.0, -1e-4], [2e-4, 1e-4, 1.0]], dtype=torch.float64))


def test_rotation_matrix_to_angle_axis_against_scipy_all_quaternion_branches():
    aa = _aa(400, 1, 0.9)
    # rotations by nearly pi about each axis and about a diagonal: the three trace <= 0 branches
    near_pi = torch.tensor([[3.1, 0.0, 0.0], [0.0, 3.1, 0.0], [0.0, 0.0, 3.1], [1.8, 1.8, 1.7], [-2.2, 2.1, 0.3]], dtype=torch.float64)
    aa = torch.cat([aa, near_pi])
    R = torch.from_numpy(Rotation.from_rotvec(aa.numpy()).as_matrix())
    got = RO.rotation_matrix_to_angle_axis(R)
    ref = torch.from_numpy(Rotation.from_matrix(R.numpy()).as_rotvec())
    assert float((got - ref).abs().max()) < 1e-6          # (kornia's eps = 1e-8 under the square roots)
    inside = aa.norm(dim=-1) < np.pi - 1e-3
    assert float((got - aa)[inside].abs().max()) < 1e-6 and int(inside.sum()) > 380      # round trip where |aa| < pi
    q = RO.rotation_matrix_to_quaternion_wxyz(R)
    assert float((q.norm(dim=-1) - 1.0).abs().max()) < 1e-7
    used = [int(((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]) > 0).sum()), int(((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]) <= 0).sum())]
    assert used[0] > 100 and used[1] >= 5
    ident = RO.rotation_matrix_to_angle_axis(torch.eye(3, dtype=torch.float64)[None])
    assert float(ident.abs().max()) < 1e-7


def test_rba_forward_gauge_shapes_and_float32():
    g = torch.Generator().manual_seed(3)
    K, num = 6, 10
    init_r = torch.randn((num, 3), generator=g) * 0.5
    init_t = torch.randn((num, 3), generator=g)
    dims = [(256, 7), (256,), (256, 256), (256,), (256, 256), (256,), (6, 256), (6,)]
    params = [torch.randn(d, generator=g) * 0.05 for d in dims]
    ids = torch.tensor([0, 1, 2, 5, 5, 9])
    c2w = RO.rba_forward(params, init_r, init_t, ids, num, 1e-2)
    assert c2w.shape == (K, 4, 4) and c2w.dtype == torch.float32
    # camera 0 is the gauge: its pose is its initial pose whatever the MLP says (model/rba.py:95-96)
    assert torch.allclose(c2w[0], RO.make_c2w(init_r[:1], init_t[:1])[0], atol=0, rtol=0)
    assert torch.equal(c2w[3], c2w[4]) and not torch.equal(c2w[1], RO.make_c2w(init_r[1:2], init_t[1:2])[0])
    assert torch.equal(c2w[:, 3, :], torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(K, 1))
    # scale -> 0: every camera at its initial pose
    c0 = RO.rba_forward(params, init_r, init_t, ids, num, 0.0)
    assert torch.allclose(c0, RO.make_c2w(init_r[ids], init_t[ids]), atol=1e-7)
