"""Round-6 fixes of ADVICE r5, on the host-emulated kernels (CPU only):
  * Adam's steady-state path compares pointer / strides / dtype of EVERY gradient on every call and keeps no gradient tensors;
  * a trunk submodule replaced after the first forward call is seen (owner identity, not only slot identity);
  * the row-window weight-gradient kernel's eligibility rejects zero channel counts instead of dividing by zero tiles;
  * the engine's per-arena forward records die with the arena tensor (dcn_plan_forget_saved)."""
import copy
import ctypes
import gc

import pytest
import torch

from helpers import rel_err, use_emulation_library


@pytest.fixture(scope="module", autouse=True)
def _lib():
    return use_emulation_library()


def test_adam_fast_path_follows_a_repointed_gradient_and_keeps_no_gradients():
    from dcn_hip.optim import Adam
    g = torch.Generator().manual_seed(0)
    ours = [torch.nn.Parameter(torch.randn(257, generator=g)), torch.nn.Parameter(torch.randn(16, 8, 3, 3, generator=g))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    o, r = Adam(ours, lr=1e-2), torch.optim.Adam(ref, lr=1e-2, foreach=False)
    bufs = [[torch.randn(p.shape, generator=g) for p in ours] for _ in range(3)]
    for p, q, b in zip(ours, ref, bufs[0]):
        p.grad, q.grad = b.clone(), b.clone()
    o.step(); r.step()
    o.step(); r.step()                        # second call: the steady-state path
    assert o._fast and all(it[1] is None for it in o._fast[0]["items"])     # pointers only: last step's gradients are not kept alive
    # the SAME gradient tensor objects, re-pointed at other storage (`p.grad.data = ...`): the cached address tables are stale
    for p, q, b in zip(ours, ref, bufs[1]):
        keep = p.grad
        p.grad.data = b.clone()
        assert p.grad is keep
        q.grad = b.clone()
    o.step(); r.step()
    for p, q in zip(ours, ref):
        assert rel_err(p, q) < 2e-6
    # a gradient of another dtype on the same object is refused by the general path, not read as float32
    ours[0].grad.data = ours[0].grad.data.double()
    with pytest.raises(TypeError):
        o.step()


def _narrow():
    import pytorch_segmentation_detection.models.resnet_dilated as rd
    torch.manual_seed(0)
    return rd.Resnet18_8s(num_classes=3, base_width=8)


def test_replaced_submodule_is_seen_by_the_engine_tables():
    net = _narrow()
    net.eval()
    x = torch.randn(1, 3, 32, 48)
    with torch.no_grad():
        y0 = net(x).clone()
        trunk = getattr(net, net.attr)
        block = trunk.get_submodule("layer1.0")   # (parameter containers: layer1 -> "0" -> conv1)
        old = block.conv1
        new = copy.deepcopy(old)
        new.weight.mul_(0.5)
        block.conv1 = new                     # the detached `old` still owns ITS tensors: slot identity alone would pass
        y1 = net(x).clone()
        assert not torch.equal(y0, y1)
        new.weight.copy_(old.weight)
        assert torch.equal(net(x), y0)


def test_row_window_wgrad_eligibility_rejects_degenerate_descriptors():
    from dcn_hip import _lib as L
    lib = L.get()
    for cin, cout in ((0, 64), (64, 0), (0, 0)):
        d = L.ConvDesc(2, 16, 32, cin, 16, 32, cout, 3, 3, 1, 1, 1, cout, 0)
        assert lib.dcn_conv_wgrad_hl_kind(ctypes.byref(d)) == 0          # (was: SIGFPE in wgrad_hlr_splits, 256 / 0 tiles)
        assert lib.dcn_conv_wgrad_hl_eligible(ctypes.byref(d)) == 0
        assert lib.dcn_conv_wgrad_workspace_hl(ctypes.byref(d)) == 0
    d = L.ConvDesc(2, 16, 32, 32, 16, 32, 64, 3, 3, 1, 1, 1, 64, 0)     # half a 64-channel tile: the tile kernel's, never the row-window one's
    assert lib.dcn_conv_wgrad_hl_kind(ctypes.byref(d)) == 1
    d = L.ConvDesc(2, 16, 32, 64, 16, 32, 64, 3, 3, 1, 1, 1, 64, 0)
    assert lib.dcn_conv_wgrad_workspace_hl(ctypes.byref(d)) > 0


def test_forward_records_die_with_their_arena():
    from dcn_hip import backbone as bb
    net = _narrow()
    net.train()
    x = torch.randn(1, 3, 32, 48)
    plan = bb.get_plan("Resnet18_8s", 8, 1, 32, 48, 3)
    base = plan.num_forward_records()
    ys = [net(x) for _ in range(3)]           # three arenas alive, none differentiated yet
    assert plan.num_forward_records() == base + 3
    ys[0].sum().backward()                    # backward releases its arena
    gc.collect()
    assert plan.num_forward_records() == base + 2
    del ys
    gc.collect()
    assert plan.num_forward_records() == base
    with torch.no_grad():                     # training-mode forward without a graph: the arena is dropped at once
        net(x)
    gc.collect()
    assert plan.num_forward_records() == base
