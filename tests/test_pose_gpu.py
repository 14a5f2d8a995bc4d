"""GPU tests of the fused RBA pose MLP: librfx kernels vs the same map written as ATen ops
(`RBA.forward_torch`, the reference formulation of model/rba.py:79-100)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rba(num_cams=40, seed=0, scale=1e-2):
    from remixfusion_amd.model.rba import RBA, make_c2w
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    aa = torch.randn((num_cams, 3), generator=g) * 0.8
    aa[3] = 0.0                                           # identity rotation: first-order branch
    aa[4] = torch.tensor([1e-4, -2e-4, 3e-4])            # theta^2 < eps
    t = torch.randn((num_cams, 3), generator=g) * 2.0
    init = make_c2w(aa, t)
    m = RBA(num_cams, init_c2w=init, scale=scale, device="cuda").cuda()
    return m


@pytest.mark.parametrize("scale", [1e-2, 1.0])
def test_fused_rba_matches_torch_ops(scale):
    m = _rba(scale=scale)
    ids = torch.tensor([0, 1, 2, 3, 4, 7, 7, 39, 20, 5], device="cuda").unsqueeze(-1)
    ref = m.forward_torch(ids)
    got = m(ids)
    assert got.shape == ref.shape == (10, 4, 4)
    assert float((got - ref).detach().abs().max()) < 2e-6
    assert float((got[0] - m.init_c2w[0]).detach().abs().max()) < 1e-5      # camera 0 is pinned to its initial pose
    g = torch.Generator().manual_seed(9)
    dp = torch.randn((10, 4, 4), generator=g).cuda()
    params = list(m.parameters())
    gr = torch.autograd.grad(ref, params, dp, allow_unused=True)
    gg = torch.autograd.grad(got, params, dp, allow_unused=True)
    for p, a, b in zip(params, gg, gr):
        assert a is not None and a.shape == p.shape
        tol = 2e-4 * float(b.abs().max()) + 1e-9
        assert float((a - b).abs().max()) <= tol, (tuple(p.shape), float((a - b).abs().max()), float(b.abs().max()))
    assert float(gr[2].abs().max()) > 0


def test_fused_rba_in_the_pose_loop():
    """a few Adam steps on a pose-only objective move the poses the same way through both paths."""
    import copy
    m1 = _rba(num_cams=12, seed=3)
    m2 = copy.deepcopy(m1)
    ids = torch.arange(12, device="cuda").unsqueeze(-1)
    target = m1.forward_torch(ids).detach().clone()
    target[:, :3, 3] += 0.05
    for m, fn in ((m1, m1.forward), (m2, m2.forward_torch)):
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        for _ in range(20):
            opt.zero_grad()
            loss = ((fn(ids) - target) ** 2).sum()
            loss.backward()
            opt.step()
    a, b = m1(ids), m2.forward_torch(ids)
    assert float((a - b).detach().abs().max()) < 1e-4
    assert float((a[1:, :3, 3] - target[1:, :3, 3]).detach().abs().mean()) < 0.05 - 1e-3      # moved towards the target
