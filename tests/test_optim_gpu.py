"""rfx_adam_step / remixfusion_amd.optim.Adam against torch.optim.Adam (the reference's optimizer,
mp_slam/slam.py:271-286): same updates, same state layout."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    shapes = [(32, 81), (16, 32), (6,), (1 << 16, 2), (4099,), (7, 256)]
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda", generator=g)) for s in shapes]
    groups = [{"params": ps[:3], "weight_decay": 1e-6, "lr": 0.01},
              {"params": ps[3:5], "eps": 1e-15, "lr": 0.01},
              {"params": ps[5:], "weight_decay": 1e-6, "eps": 1e-15, "lr": 5e-4}]
    return ps, groups


def test_adam_step_matches_torch_adam_and_shares_its_state_layout():
    from remixfusion_amd.optim import Adam
    pa, ga = _groups(3)
    pb, gb = _groups(3)
    ours, ref = Adam(ga, betas=(0.9, 0.99)), torch.optim.Adam(gb, betas=(0.9, 0.99), foreach=False, fused=False)
    g = torch.Generator(device="cuda").manual_seed(9)
    for it in range(6):
        for a, b in zip(pa, pb):
            grad = torch.randn(a.shape, device="cuda", generator=g) * (10.0 ** (it - 3))
            if it == 2:
                grad[grad.abs() < 0.5 * grad.abs().max()] = 0.0            # mostly-zero gradients, as the hash table sees
            a.grad, b.grad = grad.clone(), grad.clone()
        ours.step(); ref.step()
        for a, b in zip(pa, pb):
            assert float((a - b).detach().abs().max()) <= 2e-6 * float(b.detach().abs().max()), (it, tuple(a.shape))
            for key in ("exp_avg", "exp_avg_sq"):
                x, y = ours.state[a][key], ref.state[b][key]
                assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max()) + 1e-30, (it, key)
            assert float(ours.state[a]["step"]) == float(ref.state[b]["step"]) == it + 1
    # a parameter without a gradient is left alone; state_dict round-trips through torch's loader both ways
    before = pa[0].detach().clone()
    pa[0].grad = None
    for a in pa[1:]:
        a.grad = torch.ones_like(a)
    ours.step()
    assert torch.equal(pa[0], before)
    sd = ours.state_dict()
    pc, gc = _groups(3)
    third = torch.optim.Adam(gc, betas=(0.9, 0.99), foreach=False, fused=False)
    third.load_state_dict(sd)
    assert float(third.state[pc[1]]["step"]) == 7.0
    assert torch.equal(third.state[pc[3]]["exp_avg"], ours.state[pa[3]]["exp_avg"])
    fourth = Adam(_groups(3)[1], betas=(0.9, 0.99))
    fourth.load_state_dict(ref.state_dict())
    assert float(fourth.state[fourth.param_groups[0]["params"][0]]["step"]) == 6.0


def test_adam_step_argument_checks():
    from remixfusion_amd import _lib as L
    lib = L.load()
    st = L.stream_ptr(torch.device("cuda"))
    f = torch.zeros(64, device="cuda")
    ok = L.AdamTensor(L.ptr(f), L.ptr(f), L.ptr(f), L.ptr(f), 64, 0.9, 0.99, 0.1, 0.01, 1e-8, 0.0, -0.01, 0.1)
    assert lib.rfx_adam_step((L.AdamTensor * 1)(ok), 1, st) == 0
    assert lib.rfx_adam_step(None, 0, st) == 0
    assert lib.rfx_adam_step(None, 1, st) == -1
    assert lib.rfx_adam_step((L.AdamTensor * 1)(ok), L.ADAM_MAX_TENSORS + 1, st) == -1
    bad = L.AdamTensor(L.ptr(f), None, L.ptr(f), L.ptr(f), 64, 0.9, 0.99, 0.1, 0.01, 1e-8, 0.0, -0.01, 0.1)
    assert lib.rfx_adam_step((L.AdamTensor * 1)(bad), 1, st) == -1
    bad = L.AdamTensor(L.ptr(f), L.ptr(f), L.ptr(f), L.ptr(f), 64, 0.9, 0.99, 0.1, 0.01, 1e-8, 0.0, -0.01, 0.0)     # step 0
    assert lib.rfx_adam_step((L.AdamTensor * 1)(bad), 1, st) == -1
    torch.cuda.synchronize()
