"""dcn_adam_step (the `optimizer.step()` of training.py:346) through the C ABI -- kernels compiled for the host
(tests/hostemu) -- against torch.optim.Adam, which IS the reference's optimizer (training.py:133-145).
CPU only; the same checks run on the real gfx950 build in test_gpu_parity.py."""
import copy

import pytest
import torch

from helpers import use_emulation_library


@pytest.fixture(scope="module", autouse=True)
def _lib():
    return use_emulation_library()


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64,), (7,), (1,), (5, 3), (64, 64, 3, 3), (8200,), (4097,)]
    ps = [torch.randn(*s, generator=g) for s in shapes]
    ps.append(torch.randn(16, 8, 3, 3, generator=g).contiguous(memory_format=torch.channels_last))   # conv weight layout of the engine
    ps.append(torch.randn(4099, generator=g)[3:])                                                   # 4-byte aligned only
    return [torch.nn.Parameter(p) for p in ps]


def _set_grads(ps, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    for p in ps:
        p.grad = torch.empty_like(p).copy_(torch.randn(p.shape, generator=g) * scale)


def _max_rel(a, b):
    return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("wd", [0.0, 1.0e-4])
def test_adam_matches_torch_over_steps(wd):
    """Ten updates, learning-rate decay in the middle (adjust_learning_rate, training.py:544-558), gradient magnitudes from
    1e-8 to 1e3: parameters and both moments follow torch.optim.Adam to float32 round-off."""
    from dcn_hip.optim import Adam
    ours, ref = _params(), _params()
    o = Adam(ours, lr=1.0e-3, weight_decay=wd)
    r = torch.optim.Adam(ref, lr=1.0e-3, weight_decay=wd, foreach=False)
    for it in range(10):
        scale = [1.0, 1e-8, 1e3, 1e-3][it % 4]
        _set_grads(ours, 100 + it, scale)
        _set_grads(ref, 100 + it, scale)
        if it == 5:
            for opt in (o, r):
                for gr in opt.param_groups:
                    gr["lr"] *= 0.9
        o.step()
        r.step()
    for a, b in zip(ours, ref):
        assert _max_rel(a, b) < 2e-6
        assert _max_rel(o.state[a]["exp_avg"], r.state[b]["exp_avg"]) < 2e-6
        assert _max_rel(o.state[a]["exp_avg_sq"], r.state[b]["exp_avg_sq"]) < 2e-6
        assert float(o.state[a]["step"]) == float(r.state[b]["step"]) == 10.0


def test_adam_first_step_is_sign_of_gradient():
    """Known answer: after one step from zero moments, p moves by -lr * g / (|g| + eps) whatever the gradient's scale."""
    from dcn_hip.optim import Adam
    p = torch.nn.Parameter(torch.zeros(1000))
    p.grad = torch.linspace(-3, 3, 1000) * 1e-3
    Adam([p], lr=0.5).step()
    want = -0.5 * p.grad / (p.grad.abs() + 1e-8)
    assert float((p.detach() - want).abs().max()) < 1e-6


def test_adam_state_dict_round_trips_with_torch():
    """`.pth.opt` files (training.py:501-521) written by either optimizer load into the other and training continues identically."""
    from dcn_hip.optim import Adam
    ours, ref = _params(1), _params(1)
    o, r = Adam(ours, lr=2e-3, weight_decay=1e-4), torch.optim.Adam(ref, lr=2e-3, weight_decay=1e-4)
    for it in range(3):
        _set_grads(ours, it); _set_grads(ref, it)
        o.step(); r.step()
    # swap: each continues from the other's checkpoint
    o2 = Adam(ours, lr=1.0)
    o2.load_state_dict(copy.deepcopy(r.state_dict()))
    r2 = torch.optim.Adam(ref, lr=1.0)
    r2.load_state_dict(copy.deepcopy(o.state_dict()))
    assert o2.param_groups[0]["lr"] == 2e-3 and r2.param_groups[0]["weight_decay"] == 1e-4
    for it in range(3, 6):
        _set_grads(ours, it); _set_grads(ref, it)
        o2.step(); r2.step()
    for a, b in zip(ours, ref):
        assert _max_rel(a, b) < 3e-6
    assert float(o2.state[ours[0]]["step"]) == 6.0


def test_adam_skips_parameters_without_gradient_and_relays_foreign_layouts():
    from dcn_hip.optim import Adam
    ours, ref = _params(2), _params(2)
    o, r = Adam(ours, lr=1e-2), torch.optim.Adam(ref, lr=1e-2)
    for it in range(2):
        _set_grads(ours, it); _set_grads(ref, it)
        ours[1].grad = None; ref[1].grad = None
        # an NCHW-contiguous gradient for the channels_last weight (what autograd hands out without FlatGradients)
        ours[7].grad = ours[7].grad.contiguous(memory_format=torch.contiguous_format)
        assert ours[7].grad.stride() != ours[7].stride()
        o.step(); r.step()
    assert len(o.state[ours[1]]) == 0
    for a, b in zip(ours, ref):
        assert _max_rel(a, b) < 2e-6


def test_adam_rejects_what_it_does_not_implement():
    from dcn_hip.optim import Adam
    p = torch.nn.Parameter(torch.zeros(4))
    with pytest.raises(NotImplementedError):
        Adam([p], amsgrad=True)
    with pytest.raises(ValueError):
        Adam([p], lr=-1.0)
    q = torch.nn.Parameter(torch.zeros(4, dtype=torch.float64))
    q.grad = torch.zeros(4, dtype=torch.float64)
    with pytest.raises(TypeError):
        Adam([q]).step()
    s = torch.nn.Parameter(torch.zeros(8, 8)[:, ::2])
    s.grad = torch.zeros(8, 4)
    with pytest.raises(ValueError):
        Adam([s]).step()


def test_adam_fast_path_and_its_invalidation():
    """The steady-state path of Adam.step() (round 5: the pointer tables of the previous call, after checking that no
    parameter, gradient buffer or state tensor moved) against torch.optim.Adam: gradients written IN PLACE into the same buffers
    (the fast path), then a gradient tensor replaced, a missing gradient, a changed learning rate and a loaded state dict --
    each must fall back to the general path and stay on torch's trajectory."""
    from dcn_hip.optim import Adam
    ours, ref = _params(3), _params(3)
    o = Adam(ours, lr=2e-3, weight_decay=1e-4)
    r = torch.optim.Adam(ref, lr=2e-3, weight_decay=1e-4, foreach=False)
    _set_grads(ours, 1)
    _set_grads(ref, 1)
    taken = []
    orig = o._fast_step
    o._fast_step = lambda *a: taken.append(orig(*a)) or taken[-1]

    def inplace_grads(seed):
        g = torch.Generator().manual_seed(seed)
        for p, q in zip(ours, ref):
            v = torch.randn(p.shape, generator=g)
            p.grad.copy_(v)
            q.grad.copy_(v)
    for it in range(4):          # steps 2-4 reuse the buffers: fast path
        if it:
            inplace_grads(10 + it)
        o.step(); r.step()
    assert taken == [False, True, True, True]
    ours[2].grad = ours[2].grad.clone()          # a new gradient tensor for one parameter
    o.step(); r.step()
    assert taken[-1] is False
    inplace_grads(30)
    o.step(); r.step()
    assert taken[-1] is True
    for gr in o.param_groups + r.param_groups:   # adjust_learning_rate: read from the group every call
        gr["lr"] *= 0.5
    inplace_grads(31)
    o.step(); r.step()
    assert taken[-1] is True
    o.load_state_dict(copy.deepcopy(o.state_dict()))   # new state tensors
    inplace_grads(32)
    o.step(); r.step()
    assert taken[-1] is False
    g5 = ours[5].grad
    ours[5].grad, ref[5].grad = None, None       # a parameter without a gradient is skipped (its step does not advance)
    o.step(); r.step()
    assert taken[-1] is False
    ours[5].grad, ref[5].grad = g5, g5.clone()
    o.step(); r.step()                           # steps now differ between parameters: general path, two launches
    assert taken[-1] is False
    for a, b in zip(ours, ref):
        assert _max_rel(a, b) < 3e-6
        assert float(o.state[a]["step"]) == float(r.state[b]["step"])
