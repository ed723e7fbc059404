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


# ---- order-independent loss backward (64-bit fixed-point accumulation) ----------------------------------------------------

def _loss_grads(A, B, lists8, cfg_dict, H, W, exact, D=None):
    """gradient maps of the composed loss for ONE image pair through the product's get_loss, with the exact / float backward"""
    from dcn_hip import loss as K
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    old = K.EXACT_BACKWARD
    K.EXACT_BACKWARD = exact
    try:
        a = A.clone().requires_grad_(True)
        b = B.clone().requires_grad_(True)
        pcl = PixelwiseContrastiveLoss([H, W], cfg_dict)
        out = loss_composer.get_loss(pcl, torch.tensor([0]), a, b, *lists8)
        out[0].backward()
        return a.grad.clone(), b.grad.clone()
    finally:
        K.EXACT_BACKWARD = old


def test_exact_loss_backward_matches_the_reference_goldens_and_is_order_independent():
    import glob
    import os
    from helpers import lists_from_golden, load_golden_loss
    for name in ("within_d3", "within_d16", "within_margins", "within_blind"):
        z, cfg = load_golden_loss(os.path.join(os.path.dirname(__file__), "golden", "loss_ref_%s.npz" % name))
        A, B = torch.tensor(z["A"]), torch.tensor(z["B"])
        H, W = int(z["H"]), int(z["W"])
        lists = lists_from_golden(z)
        ga, gb = _loss_grads(A, B, lists, cfg, H, W, exact=True)
        assert rel_err(ga, z["gradA"]) < 1e-5 and rel_err(gb, z["gradB"]) < 1e-5          # the reference's own gradients
        fa, fb = _loss_grads(A, B, lists, cfg, H, W, exact=False)
        assert rel_err(ga, fa) < 1e-6 and rel_err(gb, fb) < 1e-6
        # the same multiset of pixel pairs in another order: other lanes, other workgroups, another order of the atomics
        g = torch.Generator().manual_seed(5)
        shuffled = []
        for t in range(4):
            a_, b_ = lists[2 * t], lists[2 * t + 1]
            if a_.numel() > 1:
                perm = torch.randperm(a_.numel(), generator=g)
                a_, b_ = a_[perm], b_[perm]
            shuffled += [a_, b_]
        ga2, gb2 = _loss_grads(A, B, tuple(shuffled), cfg, H, W, exact=True)
        assert torch.equal(ga, ga2) and torch.equal(gb, gb2), name


def test_exact_loss_backward_with_every_pair_on_one_pixel_and_non_finite_input():
    from oracle import synth
    H, W, D = 16, 24, 3
    g = torch.Generator().manual_seed(2)
    A = (torch.rand(1, H * W, D, generator=g) * 2 - 1) * 0.3
    B = (torch.rand(1, H * W, D, generator=g) * 2 - 1) * 0.3
    n = 3000
    ma = torch.full((n,), 7, dtype=torch.int64)                 # 3000 contributions to ONE pixel of A
    mb = torch.randint(0, H * W, (n,), generator=g)
    ka, kb = torch.randint(0, H * W, (n,), generator=g), torch.full((n,), 11, dtype=torch.int64)
    empty = torch.tensor([-1])
    lists = (ma, mb, ka, kb, ka.clone(), kb.clone(), empty, empty)
    ga, gb = _loss_grads(A, B, lists, synth.LOSS_CONFIG, H, W, exact=True)
    # float64 reference of the accumulated gradient (loss_numpy-style, via torch double autograd on the oracle)
    from oracle import loss_oracle
    a64, b64 = A.double().requires_grad_(True), B.double().requires_grad_(True)
    pcl = loss_oracle.PixelwiseContrastiveLoss([H, W], synth.LOSS_CONFIG)
    loss_oracle.get_loss(pcl, torch.tensor([0]), a64, b64, *lists)[0].backward()
    assert rel_err(ga, a64.grad) < 2e-6 and rel_err(gb, b64.grad) < 2e-6
    perm = torch.randperm(n, generator=g)
    lists2 = (ma[perm], mb[perm], ka.flip(0), kb.flip(0), ka.clone(), kb.clone(), empty, empty)
    ga2, gb2 = _loss_grads(A, B, lists2, synth.LOSS_CONFIG, H, W, exact=True)
    assert torch.equal(ga, ga2) and torch.equal(gb, gb2)
    # a NaN descriptor that a pair reads poisons that image pair's maps (the float path puts the NaN where it lands)
    A2 = A.clone()
    A2[0, 7, 1] = float("nan")
    ga3, _ = _loss_grads(A2, B, lists, synth.LOSS_CONFIG, H, W, exact=True)
    assert torch.isnan(ga3).any()
    # no pixel pairs at all: zero maps
    none = (empty,) * 8
    from dcn_hip import loss as K
    from dcn_hip.loss import PairLists
    pl = PairLists.from_lists([none], torch.device("cpu"), hw=H * W)
    assert pl.total == 0


def test_hl32_eligibility_from_reduction_length_128(dcn_env):
    """Round 6: the 1 x 1 convolutions fed by 128 - 1023 channels (downsample branches, bottleneck conv3) take the hl32 kernels;
    DCN_HL_MIN_K=1024 restores the thresholds of round 5 (forward / dgrad from K = 512, weight gradient from K = 1024)."""
    from dcn_hip import _lib as L
    lib = L.get()
    down4 = L.ConvDesc(8, 60, 80, 256, 60, 80, 512, 1, 1, 1, 0, 1, 512, 0)      # layer4 downsample 1 x 1 256 -> 512
    down3 = L.ConvDesc(8, 60, 80, 128, 60, 80, 256, 1, 1, 1, 0, 1, 256, 0)      # layer3 downsample 1 x 1 128 -> 256
    conv3 = L.ConvDesc(4, 120, 160, 256, 120, 160, 1024, 1, 1, 1, 0, 1, 1024, 0)  # ResNet50-8s layer3 conv3
    wide = L.ConvDesc(8, 60, 80, 512, 60, 80, 512, 3, 3, 1, 4, 4, 512, 0)
    for d in (down4, down3, conv3):
        assert lib.dcn_conv_hl_eligible(ctypes.byref(d), 0) == 1 and lib.dcn_conv_wgrad_hl_eligible(ctypes.byref(d)) == 1
    dcn_env(DCN_HL_MIN_K=1024)
    for d in (down4, down3, conv3):
        assert lib.dcn_conv_hl_eligible(ctypes.byref(d), 0) == 0 and lib.dcn_conv_wgrad_hl_eligible(ctypes.byref(d)) == 0
    assert lib.dcn_conv_hl_eligible(ctypes.byref(wide), 0) == 1 and lib.dcn_conv_wgrad_hl_eligible(ctypes.byref(wide)) == 1
