"""remixfusion_amd.optim.Adam on CPU tensors: the wrapper hands over to torch's own step (no HIP path), keeps torch's state
layout, and fires step hooks exactly once although it bypasses torch's wrapped `step`."""
import torch


def test_cpu_parameters_take_torchs_step_and_hooks_fire_once():
    from remixfusion_amd.optim import Adam
    p = torch.nn.Parameter(torch.linspace(-1, 1, 12).reshape(3, 4).clone())
    q = torch.nn.Parameter(p.detach().clone())
    ours = Adam([p], lr=0.05, betas=(0.9, 0.99), weight_decay=1e-6)
    ref = torch.optim.Adam([q], lr=0.05, betas=(0.9, 0.99), weight_decay=1e-6)      # instantiating it wraps torch.optim.Adam.step
    pre, post = [], []
    g = torch.Generator().manual_seed(3)
    for it in range(4):
        grad = torch.randn(p.shape, generator=g)
        p.grad, q.grad = grad.clone(), grad.clone()
        if it == 1:
            h0 = ours.register_step_pre_hook(lambda *a: pre.append(it))
            h1 = ours.register_step_post_hook(lambda *a: post.append(it))
        if it == 3:
            h0.remove(); h1.remove()
        ours.step(); ref.step()
        assert torch.equal(p.detach(), q.detach())
        assert float(ours.state[p]["step"]) == float(ref.state[q]["step"]) == it + 1
    assert pre == [1, 2] and post == [1, 2]
    sd = ours.state_dict()
    third = torch.optim.Adam([torch.nn.Parameter(torch.zeros(3, 4))], lr=0.05, betas=(0.9, 0.99))
    third.load_state_dict(sd)
    assert float(third.state[third.param_groups[0]["params"][0]]["step"]) == 4.0


def test_closure_is_evaluated_with_grad_enabled():
    from remixfusion_amd.optim import Adam
    p = torch.nn.Parameter(torch.ones(5))
    opt = Adam([p], lr=0.1)

    def closure():
        opt.zero_grad()
        loss = (p * p).sum()
        loss.backward()
        return loss

    out = opt.step(closure)
    assert float(out.detach()) == 5.0 and float(p.detach().max()) < 1.0
