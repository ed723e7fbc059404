"""The whole backbone engine (plan -> forward -> backward; every kernel, through the C ABI, host-emulated)
against the oracle on narrow networks.  The oracle itself is float32, so the yard-stick is a float64 copy of
it: the engine must be as close to the float64 truth as the float32 oracle is (x3 + 1e-5)."""
import copy
import ctypes

import pytest
import torch

from helpers import rel_err, use_emulation_library


@pytest.fixture(scope="module", autouse=True)
def _lib():
    return use_emulation_library()


@pytest.fixture(autouse=True, params=["fp32", "f16x3"])
def conv_mode(request):
    """Every test of this file runs in both convolution arithmetics; the tolerances are the same (include/dcn_hip.h)."""
    from dcn_hip import backbone
    backbone.set_conv_mode(request.param)
    yield request.param
    backbone.set_conv_mode(None)


def _pair(arch, D, bw, seed=0):
    from oracle import resnet_dilated_oracle as orc
    from pytorch_segmentation_detection.models import resnet_dilated as prod
    o = orc.build(arch, D, seed=seed, base_width=bw)
    m = getattr(prod, arch)(num_classes=D, base_width=bw)
    m.load_state_dict(o.state_dict(), strict=True)
    return m, o


@pytest.mark.parametrize("arch,bw,shape,D", [
    ("Resnet18_8s", 8, (2, 32, 40), 3),
    ("Resnet34_8s", 8, (2, 32, 40), 3),
    ("Resnet34_8s", 16, (1, 40, 24), 16),
    ("Resnet50_8s", 8, (2, 32, 40), 5),
])
def test_forward_backward_vs_oracle(arch, bw, shape, D):
    N, H, W = shape
    m, o = _pair(arch, D, bw)
    o64 = copy.deepcopy(o).double()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, 3, H, W, generator=g)
    gy = torch.randn(N, D, H, W, generator=g)
    m.train(); o.train(); o64.train()
    y = m(x)
    assert y.shape == (N, D, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    yo, y64 = o(x), o64(x.double())
    tol = 3 * rel_err(yo, y64) + 1e-5
    assert rel_err(y, y64) < tol, (rel_err(y, y64), tol)
    (y * gy).sum().backward(); (yo * gy).sum().backward(); (y64 * gy.double()).sum().backward()
    for (k, p), (_, po), (_, p6) in zip(m.named_parameters(), o.named_parameters(), o64.named_parameters()):
        tol = 3 * rel_err(po.grad, p6.grad) + 2e-5
        assert rel_err(p.grad, p6.grad) < tol, (k, rel_err(p.grad, p6.grad), tol)
    # BN running statistics / counters follow nn.BatchNorm2d
    for (k, b), (_, bo) in zip(m.named_buffers(), o.named_buffers()):
        assert rel_err(b.float(), bo.float()) < 1e-4 or float((b.float() - bo.float()).abs().max()) < 1e-5, k


@pytest.mark.parametrize("gscale", [1e-30, 1e-9, 1e6, 1e25])
def test_gradient_scale_invariance(gscale, conv_mode):
    """Tiny / huge incoming gradients (loss scaling, very small learning signals): the split-fp16 convolutions pre-scale
    every gradient tensor by a power of two chosen from its abs-max, so the result is that of the fp32 kernels."""
    m, o = _pair("Resnet18_8s", 3, 8)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 32, 40, generator=g)
    gy = torch.randn(2, 3, 32, 40, generator=g) * gscale
    m.train(); o.train()
    (m(x) * gy).sum().backward(); (o(x) * gy).sum().backward()
    for (k, p), po in zip(m.named_parameters(), o.parameters()):
        assert rel_err(p.grad, po.grad) < 2e-4, (k, conv_mode)


@pytest.mark.parametrize("scale", [1.0, 3.0e5, 1.0e-6])
def test_activation_bounds_and_operand_prescale(scale):
    """Split-fp16 mode: every convolution input carries an abs-max scalar -- the image's exact abs-max, and for batch-norm
    outputs an upper BOUND computed from the statistics by bn_finalize (no atomics in the big kernels).  The bounds must hold
    for the true activations (checked on the oracle's), the status word stays clear, and activations far outside fp16's range
    (bn1 scaled by 3e5 / 1e-6) still give fp32-level results."""
    from dcn_hip import backbone
    backbone.set_conv_mode("f16x3")
    try:
        m, o = _pair("Resnet18_8s", 3, 8)
        with torch.no_grad():
            o.resnet18_8s.bn1.weight.mul_(scale)
            o.resnet18_8s.bn1.bias.fill_(0.1 * scale)
        m.load_state_dict(o.state_dict())
        o64 = copy.deepcopy(o).double()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 3, 32, 40, generator=g) * 3
        m.train(); o.train(); o64.train()
        acts = {}
        net = o.resnet18_8s
        hooks = [net.maxpool.register_forward_hook(lambda mod, i, out: acts.__setitem__("pool", out.detach())),
                 net.layer1[0].register_forward_hook(lambda mod, i, out: acts.__setitem__("l1.0", out.detach())),
                 net.layer2[0].register_forward_hook(lambda mod, i, out: acts.__setitem__("l2.0", out.detach()))]
        y, yo, y64 = m(x), o(x), o64(x.double())
        for h in hooks:
            h.remove()
        amax, status = m.last_forward_status()
        assert int(status) == 0 and bool((amax > 0).all()) and bool(torch.isfinite(amax).all())
        assert float(amax[0]) == float(x.abs().max())                       # slot 0: the image, exact
        assert float(amax[1]) >= float(acts["pool"].abs().max())             # slot 1: stem activation / max-pool output
        # block outputs come right after the block's mid activation: slots (2: l1.0 mid, 3: l1.0 out, ...); the downsample
        # block layer2.0 has one more slot (its shortcut branch)
        assert float(amax[3]) >= float(acts["l1.0"].abs().max())
        assert float(amax[3]) <= 8 * float(acts["l1.0"].abs().max()) + 1e-30  # ... and not absurdly loose
        assert rel_err(y, y64) < 3 * rel_err(yo, y64) + 1e-5, (scale, rel_err(y, y64), rel_err(yo, y64))
    finally:
        backbone.set_conv_mode(None)


def test_forward_backward_with_stream_k_forced(dcn_env):
    """Same network with every gather-GEMM launch forced through the stream-K split + fix-up path (engine workspace)."""
    dcn_env(DCN_GEMM_SK=5)
    m, o = _pair("Resnet18_8s", 3, 8)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 32, 40, generator=g)
    gy = torch.randn(2, 3, 32, 40, generator=g)
    m.train(); o.train()
    y, yo = m(x), o(x)
    assert rel_err(y, yo) < 2e-5
    (y * gy).sum().backward(); (yo * gy).sum().backward()
    assert max(rel_err(p.grad, po.grad) for p, po in zip(m.parameters(), o.parameters())) < 1e-4


@pytest.mark.parametrize("arch,groups", [("Resnet18_8s", 1), ("Resnet50_8s", 1), ("Resnet18_8s", 2)])
def test_bn_backward_reduction_fused_into_dgrad(arch, groups, conv_mode, dcn_env):
    """Split-fp16 mode: the dgrad that produces a batch norm's upstream gradient masks it with the ReLU bits and leaves the
    per-tile sums of the BN backward reduction behind (GemmConv::bnb_*), for every batch norm but the stem's (fed by the
    max-pool backward) and the downsample branches' (fed by the residual gradient).  Same gradients as the separate
    reduce pass (DCN_BN_BWD_FUSED=0) up to summation order; stream-K tiles (inline completion and fix-up kernel) included."""
    from dcn_hip import backbone
    grads = {}
    for fused, sk, fix in ((0, None, None), (1, None, None), (1, 3, "inline"), (1, 3, "kernel")):
        env = {"DCN_BN_BWD_FUSED": fused}
        if sk:
            env.update(DCN_GEMM_SK=sk, DCN_GEMM_SK_FIXUP=fix)
        dcn_env(**env)
        m, _ = _pair(arch, 3, 8)
        g = torch.Generator().manual_seed(5)
        H, W = (64, 64) if groups == 2 else (32, 40)
        xa = torch.randn(2, 3, H, W, generator=g)
        xb = torch.randn(2, 3, H, W, generator=g)
        gy = torch.randn(2, 3, H, W, generator=g)
        m.train()
        if groups == 2:
            ya, yb = m.forward_pair(xa, xb)
            ((ya * gy).sum() + (yb * gy).sum()).backward()
        else:
            (m(xa) * gy).sum().backward()
        plan = m._last_plan
        assert plan.groups == groups
        n_bn = len(plan.bn_names)
        n_down = sum(1 for k in plan.bn_names if "downsample" in k)
        want = (n_bn - n_down - 1) if (fused and conv_mode == "f16x3") else 0
        assert plan.fused_bn_backward() == want, (plan.fused_bn_backward(), want)
        grads[(fused, sk, fix)] = [p.grad.clone() for p in m.parameters()]
    ref = grads[(0, None, None)]
    # (summation order differs: per-tile sums instead of 128-row chunks, K split by stream-K -- and these narrow, tiny-batch
    # networks amplify round-off through their batch norms, cf. the float64 yard-stick of test_forward_backward_vs_oracle)
    for key, gl in grads.items():
        assert max(rel_err(a, b) for a, b in zip(gl, ref)) < (1e-2 if key[1] else 1e-3), key
    assert all(torch.equal(a, b) for a, b in zip(grads[(1, 3, "inline")], grads[(1, 3, "kernel")]))


@pytest.mark.parametrize("arch", ["Resnet18_8s", "Resnet50_8s"])
def test_residual_gradient_added_in_bn_backward(arch, dcn_env):
    """The second summand of a block's input gradient (identity gradient, or the downsample branch's dgrad partner) is
    handed to the previous block's batch-norm backward (dy + dy2 in its streaming passes) instead of being added in the
    dgrad epilogue: the same two floats are added either way -- bit-identical gradients (DCN_DEFER_RESIDUAL_ADD=0 / 1)."""
    grads = []
    for defer in (0, 1):
        dcn_env(DCN_DEFER_RESIDUAL_ADD=defer)
        m, _ = _pair(arch, 3, 8)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(2, 3, 32, 40, generator=g)
        gy = torch.randn(2, 3, 32, 40, generator=g)
        m.train()
        (m(x) * gy).sum().backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))


def test_normalized_descriptors_forward_backward():
    """DenseCorrespondenceNetwork(normalize=True): res / ||res||_2 over D (network.py:256-259), fused into the upsample
    kernel; backward through it."""
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    m, o = _pair("Resnet18_8s", 4, 8)
    dcn = DenseCorrespondenceNetwork(m, 4, image_width=40, image_height=32, normalize=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 32, 40, generator=g)
    gy = torch.randn(1, 4, 32, 40, generator=g)
    dcn.train(); o.train()
    y = dcn.forward(x)
    ro = o(x)
    yo = ro / torch.norm(ro, 2, 1, keepdim=True)
    assert rel_err(y, yo) < 2e-5
    assert float((y.norm(2, 1) - 1).abs().max()) < 1e-5
    (y * gy).sum().backward(); (yo * gy).sum().backward()
    assert max(rel_err(p.grad, po.grad) for p, po in zip(m.parameters(), o.parameters())) < 2e-4


def test_real_width_resnet34_small_image():
    """The real Resnet34_8s (base width 64, 21.3 M parameters): exercises full 128x128 tiles, multiple N tiles,
    K = 4608 reductions and the split-K wgrad path."""
    m, o = _pair("Resnet34_8s", 3, 64)
    assert sum(p.numel() for p in m.parameters()) == 21286211
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 3, 32, 32, generator=g)
    gy = torch.randn(1, 3, 32, 32, generator=g)
    m.train(); o.train()
    y, yo = m(x), o(x)
    assert rel_err(y, yo) < 1e-4
    (y * gy).sum().backward(); (yo * gy).sum().backward()
    worst = max(rel_err(p.grad, po.grad) for p, po in zip(m.parameters(), o.parameters()))
    assert worst < 2e-3, worst   # 16 output pixels per BN: ill-conditioned, the float32 oracle is no better


@pytest.mark.parametrize("arch,bw,D", [("Resnet18_8s", 8, 3), ("Resnet34_8s", 8, 4), ("Resnet50_8s", 8, 5)])
def test_eval_mode_uses_running_statistics(arch, bw, D):
    """Inference: running statistics; in the split-fp16 mode every conv + BN (+ residual) + ReLU is one fused pass with the
    BN scale folded into the weight images (identity and downsample residuals, basic and bottleneck blocks)."""
    m, o = _pair(arch, D, bw)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 32, 32, generator=g)
    m.train(); o.train()
    with torch.no_grad():
        for _ in range(2):
            m(x * 1.5 + 0.2); o(x * 1.5 + 0.2)       # running statistics away from their initial (0, 1)
    m.eval(); o.eval()
    with torch.no_grad():
        y, yo = m(x), o(x)
        assert rel_err(y, yo) < 2e-5
        yn = m(x, normalize=True)
        assert rel_err(yn, yo / yo.norm(2, 1, keepdim=True)) < 1e-4
        # running statistics and counters untouched by inference
        assert int(getattr(m, m.attr).bn1.num_batches_tracked) == 2
        for (k, b), bo in zip(m.named_buffers(), o.buffers()):
            assert rel_err(b.float(), bo.float()) < 1e-4 or float((b.float() - bo.float()).abs().max()) < 1e-5, k
        # eval mode through the grouped entry point gives the same maps
        ya, yb = m.forward_pair(torch.cat([x, x]), torch.cat([x * 0.5, x * 0.5]))
        assert rel_err(ya[:2], yo) < 2e-5 and rel_err(yb[:2], o(x * 0.5)) < 2e-5


def test_container_nodes_refuse_to_run_and_cpu_guard():
    from dcn_hip import _lib
    m, _ = _pair("Resnet18_8s", 3, 8)
    with pytest.raises(RuntimeError):
        m.resnet18_8s.layer1(torch.zeros(1))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 32, 32))
    assert _lib.is_hostemu()   # (the shipped library rejects CPU tensors: see test_abi.py)


@pytest.mark.parametrize("arch,bw,shape", [("Resnet18_8s", 8, (1, 64, 64)), ("Resnet34_8s", 8, (2, 64, 64))])
def test_grouped_pair_forward_equals_two_forward_calls(arch, bw, shape):
    """forward_pair(a, b) == (forward(a), forward(b)): per-batch BN statistics, running statistics updated once per batch
    in call order, parameter gradients = sum of both calls' gradients.  Forward against the oracle called twice; gradients
    against the engine's own two-call result (same arithmetic, so no ReLU-kink lottery: a single pre-activation within
    round-off of zero moves a whole channel's gradient by per cent, whichever fp32-accurate implementation computes it)."""
    N, H, W = shape
    D = 3
    m, o = _pair(arch, D, bw)
    m2 = copy.deepcopy(m)
    g = torch.Generator().manual_seed(3)
    xa = torch.randn(N, 3, H, W, generator=g)
    xb = torch.randn(N, 3, H, W, generator=g) * 1.7 + 0.3        # different statistics per batch
    ga = torch.randn(N, D, H, W, generator=g)
    gb = torch.randn(N, D, H, W, generator=g)
    m.train(); m2.train(); o.train()
    ya, yb = m.forward_pair(xa, xb)
    from dcn_hip import backbone as _bb
    assert any(k[-1] == 2 and k[2] == 2 * N for k in _bb._PLANS), "forward_pair fell back to two calls"
    za, zb = m2(xa), m2(xb)
    oa, ob = o(xa), o(xb)
    assert ya.shape == oa.shape and rel_err(ya, oa) < 2e-5 and rel_err(yb, ob) < 2e-5
    assert rel_err(ya, za) < 1e-6 and rel_err(yb, zb) < 1e-6
    ((ya * ga).sum() + (yb * gb).sum()).backward()
    ((za * ga).sum() + (zb * gb).sum()).backward()
    for (k, p), p2 in zip(m.named_parameters(), m2.parameters()):
        assert rel_err(p.grad, p2.grad) < 1e-5, (k, rel_err(p.grad, p2.grad))
    for (k, b), b2, bo in zip(m.named_buffers(), m2.buffers(), o.buffers()):
        assert rel_err(b.float(), b2.float()) < 1e-6, k
        assert rel_err(b.float(), bo.float()) < 1e-4 or float((b.float() - bo.float()).abs().max()) < 1e-5, k
    # a batch whose rows are not tile-aligned silently takes the two-call route
    xs = torch.randn(1, 3, 40, 24, generator=g)
    y1, y2 = m.forward_pair(xs, xs * 0.5)
    assert rel_err(y1, o(xs)) < 2e-5 and rel_err(y2, o(xs * 0.5)) < 2e-5


@pytest.mark.parametrize("arch,bw,shape,groups,producers,hlr", [
    ("Resnet18_8s", 32, (1, 32, 40), 1, 1, 1), ("Resnet18_8s", 32, (1, 32, 40), 1, 0, 1), ("Resnet50_8s", 32, (1, 32, 32), 1, 1, 1),
    ("Resnet18_8s", 32, (2, 128, 128), 2, 1, 1),
    # DCN_WGRAD_HLR=2: the 3 x 3 / dilation-1 layers with 64 (base width 32: layer 2) or 128 (base width 64) input channels
    # take the row-window weight-gradient kernel -- saved hl32 images of THEIR inputs, the max-pool output's among them
    ("Resnet18_8s", 32, (1, 32, 40), 1, 0, 2), ("Resnet18_8s", 64, (1, 32, 40), 1, 1, 2)])
def test_wide_layers_through_the_hl32_path(arch, bw, shape, groups, producers, hlr, dcn_env, conv_mode):
    """DCN_GEMM_HL=2: every convolution the pre-split (hl32) LDS-DMA kernel supports takes it (forward and dgrad, engine
    workspace, weight images per call), DCN_WGRAD_HL=2: every weight gradient the hl32 wgrad kernel supports (saved hl32 images
    of the activations, hl32 images of the gradients from the batch-norm backward passes).  Forward: against the engine's own fp32-operand kernels and the float64 oracle.
    Backward: on the SAME forward pass (saved arena written with DCN_GEMM_HL=0, so that both backward passes see identical
    ReLU masks -- a pre-activation within round-off of zero otherwise moves a whole channel's gradient by per cent in
    whichever arithmetic it flips) the hl32 dgrads must reproduce the fp32-operand ones.  With two statistics groups
    (forward_pair) only layers whose groups are whole 256-row tiles qualify.  producers: the hl32 activation / gradient
    images are written by the batch-norm apply passes that produce the tensors (1) or by stand-alone split passes (0)."""
    if conv_mode != "f16x3":
        pytest.skip("the hl32 path belongs to the split-fp16 arithmetic")
    N, H, W = shape
    D = 3
    m, o = _pair(arch, D, bw)
    m2, m3 = copy.deepcopy(m), copy.deepcopy(m)
    o64 = copy.deepcopy(o).double()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, 3, H, W, generator=g)
    gy = torch.randn(N, D, H, W, generator=g)
    for net in (m, m2, m3, o, o64):
        net.train()

    def fwd(net):
        if groups == 2:
            return torch.cat(net.forward_pair(x[:N // 2], x[N // 2:]))
        return net(x)
    from dcn_hip import backbone as _bb
    dcn_env(DCN_GEMM_HL=2, DCN_WGRAD_HL=2, DCN_HL_PRODUCERS=producers, DCN_WGRAD_HLR=hlr)
    _bb._PLANS.clear()                     # (plans reserve the saved hl32 images when they are built: after the switches are set)
    y = fwd(m)
    if hlr == 2:   # the plan's convolutions that take the row-window kernel: at least the stride-1 3 x 3 ones of one layer
        from dcn_hip import _lib as L
        ch = 64 if bw == 32 else 128
        for hh_, ww_, c_ in ((H // 8, W // 8, ch),) + (((H // 4, W // 4, 64),) if bw == 64 else ()):
            d = L.ConvDesc(N, hh_, ww_, c_, hh_, ww_, c_, 3, 3, 1, 1, 1, c_, 0)
            assert L.get().dcn_conv_wgrad_hl_kind(ctypes.byref(d)) == 2
    dcn_env(DCN_GEMM_HL=0, DCN_WGRAD_HL=2, DCN_HL_PRODUCERS=producers)
    y2, y3 = fwd(m2), fwd(m3)
    if groups == 1:
        yo, y64 = o(x), o64(x.double())
        cond = rel_err(yo, y64)            # (how far float32 round-off moves this network's output)
        assert rel_err(y, y64) < 3 * cond + 1e-5
        assert rel_err(y, y2) < 3 * cond + 2e-5   # same products, other summation order, through the whole network
    else:
        assert rel_err(y, y2) < 1e-4
    for (k, b), b2 in zip(m.named_buffers(), m2.buffers()):
        assert rel_err(b.float(), b2.float()) < 1e-4 or float((b.float() - b2.float()).abs().max()) < 1e-5, k
    dcn_env(DCN_GEMM_HL=0, DCN_WGRAD_HL=0)
    (y2 * gy).sum().backward()             # fp32-operand dgrads and weight gradients
    dcn_env(DCN_GEMM_HL=2, DCN_WGRAD_HL=2, DCN_HL_PRODUCERS=producers)
    (y3 * gy).sum().backward()             # hl32 dgrads and weight gradients on an identical saved arena
    for (k, p2), p3 in zip(m2.named_parameters(), m3.parameters()):
        assert rel_err(p3.grad, p2.grad) < 3e-5, (k, rel_err(p3.grad, p2.grad))
    _bb._PLANS.clear()


@pytest.mark.parametrize("arch,bw,shape,groups", [("Resnet18_8s", 32, (1, 16, 256), 1),     # maps 2 x 32 at 1/8: layers 1-4 qualify
                                                  ("Resnet50_8s", 16, (1, 16, 256), 1),     # two such tensors per block (layers 2-4)
                                                  ("Resnet18_8s", 32, (2, 64, 128), 2)])    # layer 1 (16 x 32 maps), two groups
def test_activations_that_are_never_stored(arch, bw, shape, groups, dcn_env, conv_mode):
    """Inside a block, the activation between two convolutions has two readers -- the next convolution and that convolution's
    weight gradient (the batch norm's own backward takes the ReLU mask).  When both run on the hl32 kernels the apply pass
    writes the hl32 image only (DCN_HL_ONLY_MID, default on).  The stem's activation is not stored either: its batch norm +
    ReLU is applied inside the max-pool pass, which also writes the sign mask (DCN_STEM_POOL_FUSED, default on; both
    arithmetics).  Arenas poisoned with NaN bytes: forward output, running statistics and every gradient must equal, bit for
    bit, the run that writes those tensors."""
    if conv_mode == "fp32" and (arch != "Resnet18_8s" or groups != 1):
        pytest.skip("fp32 arithmetic: only the stem fusion applies; one network covers it")
    from dcn_hip import backbone as _bb
    N, H, W = shape
    D = 3
    m, _ = _pair(arch, D, bw)
    m2 = copy.deepcopy(m)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, 3, H, W, generator=g)
    gy = torch.randn(N, D, H, W, generator=g)

    def run(net):
        net.train()
        y = torch.cat(net.forward_pair(x[:N // 2], x[N // 2:])) if groups == 2 else net(x)
        (y * gy).sum().backward()
        return y.detach()
    _bb.POISON_ARENAS = True
    try:
        dcn_env(DCN_GEMM_HL=2, DCN_WGRAD_HL=2, DCN_HL_ONLY_MID=1, DCN_STEM_POOL_FUSED=1)
        _bb._PLANS.clear()
        ya = run(m)
        dcn_env(DCN_GEMM_HL=2, DCN_WGRAD_HL=2, DCN_HL_ONLY_MID=0, DCN_STEM_POOL_FUSED=0)
        yb = run(m2)
    finally:
        _bb.POISON_ARENAS = False
        _bb._PLANS.clear()
    assert bool(torch.isfinite(ya).all()) and torch.equal(ya, yb)
    for (k, p1), p2 in zip(m.named_parameters(), m2.parameters()):
        assert bool(torch.isfinite(p1.grad).all()) and torch.equal(p1.grad, p2.grad), k
    for (k, b1), b2 in zip(m.named_buffers(), m2.buffers()):
        assert torch.equal(b1, b2), k


def _torchvision_like_state_dict(arch, bw, seed=3):
    """A state dict with the keys and shapes of stock ``torchvision.models.resnetNN`` (no torchvision in this image: the
    layout is rebuilt from the published architecture -- stem, four stages of BasicBlock / Bottleneck, 1000-way fc)."""
    g = torch.Generator().manual_seed(seed)
    layers, bott = {"Resnet18_8s": ([2, 2, 2, 2], False), "Resnet34_8s": ([3, 4, 6, 3], False),
                    "Resnet50_8s": ([3, 4, 6, 3], True)}[arch]
    sd = {}

    def conv(name, o, c, k):
        sd[name + ".weight"] = torch.randn(o, c, k, k, generator=g) * 0.05

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_var"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".num_batches_tracked"] = torch.tensor(123)
    conv("conv1", bw, 3, 7); bn("bn1", bw)
    inpl, exp = bw, (4 if bott else 1)
    for li, nb in enumerate(layers):
        planes = bw << li
        for bi in range(nb):
            p = "layer%d.%d" % (li + 1, bi)
            if bott:
                conv(p + ".conv1", planes, inpl, 1); bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
                conv(p + ".conv3", planes * 4, planes, 1); bn(p + ".bn3", planes * 4)
            else:
                conv(p + ".conv1", planes, inpl, 3); bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
            if bi == 0 and (li > 0 or inpl != planes * exp):
                conv(p + ".downsample.0", planes * exp, inpl, 1); bn(p + ".downsample.1", planes * exp)
            inpl = planes * exp
    sd["fc.weight"] = torch.randn(1000, inpl, generator=g) * 0.01
    sd["fc.bias"] = torch.zeros(1000)
    return sd


@pytest.mark.parametrize("arch", ["Resnet34_8s", "Resnet50_8s"])
def test_load_imagenet_trunk_from_a_torchvision_state_dict(arch, tmp_path, conv_mode):
    """The original backbone starts from ImageNet weights (pretrained=True behind network.py:373-375,
    doc/model_zoo.md:6-18): stock torchvision keys map one to one onto the trunk, the 1000-way classifier is dropped and the
    scoring layer re-initialised."""
    if conv_mode != "fp32":
        pytest.skip("host-side loader: one arithmetic is enough")
    from pytorch_segmentation_detection.models import resnet_dilated as prod
    bw, D = 8, 3
    m = getattr(prod, arch)(num_classes=D, base_width=bw)
    sd = _torchvision_like_state_dict(arch, bw)
    path = str(tmp_path / "resnet.pth")
    torch.save(sd, path)
    fc_before = m.state_dict()[m.attr + ".fc.weight"].clone()
    loaded = m.load_imagenet_trunk(path)
    own = m.state_dict()
    assert len(loaded) == len([k for k in sd if not k.startswith("fc.")])
    for k, v in sd.items():
        if k.startswith("fc."):
            continue
        assert torch.equal(own[m.attr + "." + k].float(), v.float()), k
    fcw = own[m.attr + ".fc.weight"]
    assert tuple(fcw.shape) == (D, fc_before.shape[1], 1, 1) and not torch.equal(fcw, fc_before) and float(fcw.std()) < 0.05
    assert float(own[m.attr + ".fc.bias"].abs().max()) == 0.0
    # the network runs from the loaded trunk (train mode: batch statistics; the running statistics moved from the loaded ones)
    m.train()
    y = m(torch.randn(1, 3, 32, 40))
    assert bool(torch.isfinite(y).all())
    assert int(own[m.attr + ".bn1.num_batches_tracked"]) == 124
    # strictness: a missing tensor, an extra tensor, a wrong shape
    bad = dict(sd); bad.pop("layer2.0.conv1.weight")
    with pytest.raises(KeyError):
        m.load_imagenet_trunk(bad)
    assert len(m.load_imagenet_trunk(bad, strict=False)) == len(loaded) - 1
    bad = dict(sd); bad["layer9.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        m.load_imagenet_trunk(bad)
    bad = dict(sd); bad["conv1.weight"] = torch.zeros(bw, 3, 3, 3)
    with pytest.raises(ValueError):
        m.load_imagenet_trunk(bad)
    # DataParallel-style "module." prefixes are accepted
    m.load_imagenet_trunk({"module." + k: v for k, v in sd.items()})


def test_status_word_flags_nan_input_and_out_of_range_weights(conv_mode):
    """The status word behind the activation abs-max slots: bit 0 for a non-finite input -- inf AND NaN (the abs-max producers
    use fmaxf, which drops a NaN: it is reported as an infinite bound instead) --, bit 1 for a convolution weight outside the
    range of its fp16 image (|w| >= 1023 at the fixed weight scale 64)."""
    if conv_mode != "f16x3":
        pytest.skip("operand ranges belong to the split-fp16 arithmetic")
    m, _ = _pair("Resnet18_8s", 3, 8)
    m.train()
    x = torch.randn(1, 3, 32, 40)
    m(x)
    assert int(m.last_forward_status()[1]) == 0
    xn = x.clone(); xn[0, 1, 5, 7] = float("nan")
    m(xn)
    assert int(m.last_forward_status()[1]) & 1
    xi = x.clone(); xi[0, 2, 0, 0] = float("inf")
    m(xi)
    assert int(m.last_forward_status()[1]) & 1
    m(x)
    assert int(m.last_forward_status()[1]) == 0                              # cleared by the next call
    with torch.no_grad():
        m.resnet18_8s.get_parameter("layer2.0.conv1.weight")[3, 2, 1, 1] = 2000.0           # 64 * 2000 > 65504
    m(x)
    assert int(m.last_forward_status()[1]) & 2
    with torch.no_grad():
        m.resnet18_8s.get_parameter("layer2.0.conv1.weight")[3, 2, 1, 1] = 900.0            # inside the range again
    # the training path does not go on silently: the status word of a call is read (without a synchronisation) at the start
    # of a later call, which raises
    with pytest.raises(FloatingPointError, match="outside the range of its fp16 image"):
        m(x)
    m(x)
    assert not (int(m.last_forward_status()[1]) & 2)
    # a NaN batch-norm parameter: the statistics path reports it through the bound
    with torch.no_grad():
        m.resnet18_8s.get_parameter("layer1.0.bn1.bias")[0] = float("nan")
    m(x)
    assert int(m.last_forward_status()[1]) & 1


def test_backward_refuses_an_arena_whose_forward_decisions_no_longer_hold(dcn_env, conv_mode):
    """The forward pass records, per saved arena, what it decided under the tuning switches of the moment (which activations
    exist as the hl32 image only, which saved images were written, the arithmetic).  A backward pass under switches that would
    make it read a tensor that forward call never wrote returns DCN_E_INVALID instead of silently wrong gradients."""
    if conv_mode != "f16x3":
        pytest.skip("the hl32 decisions belong to the split-fp16 arithmetic")
    from dcn_hip import backbone as _bb
    m, _ = _pair("Resnet18_8s", 3, 32)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 3, 16, 128, generator=g)
    m.train()
    try:
        dcn_env(DCN_GEMM_HL=2, DCN_WGRAD_HL=2, DCN_HL_ONLY_MID=1)
        _bb._PLANS.clear()
        y = m(x)                                   # mid activations of the qualifying blocks: hl32 image only
        dcn_env(DCN_GEMM_HL=2, DCN_WGRAD_HL=0)     # their weight gradients would now read the fp32 tensors
        with pytest.raises(RuntimeError, match="DCN_E_INVALID"):
            y.sum().backward(retain_graph=True)
        dcn_env(DCN_GEMM_HL=2, DCN_WGRAD_HL=2, DCN_HL_ONLY_MID=1)
        _bb.set_conv_mode("fp32")                  # another arithmetic than the one that filled the arena
        with pytest.raises(RuntimeError, match="DCN_E_INVALID"):
            y.sum().backward(retain_graph=True)
        _bb.set_conv_mode("f16x3")
        m.zero_grad()
        y.sum().backward()                         # the arena is still differentiable under the switches it was made with
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    finally:
        _bb._PLANS.clear()


def test_many_forward_calls_outstanding_before_one_backward(conv_mode):
    """Ten separate training-mode forward calls of one plan, losses summed, ONE backward(): every arena is differentiated
    (the engine used to keep the decisions of the newest 8 forward calls only and refused the older arenas)."""
    m, o = _pair("Resnet18_8s", 3, 8)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(1, 3, 32, 40, generator=g) for _ in range(10)]
    gy = torch.randn(1, 3, 32, 40, generator=g)
    m.train(); o.train()
    loss = sum((m(x) * gy).sum() for x in xs)
    loss_o = sum((o(x) * gy).sum() for x in xs)
    loss.backward(); loss_o.backward()
    for (k, p), (_, po) in zip(m.named_parameters(), o.named_parameters()):
        assert rel_err(p.grad, po.grad) < 2e-4, (k, rel_err(p.grad, po.grad))


def test_profile_reports_every_engine_launch_by_category(conv_mode):
    """dcn_plan_profile_end_all: between begin and end every launch of the engine is bracketed and attributed to a category
    with its algorithmic work (FLOPs for the matrix-core categories, HBM bytes for the streaming passes)."""
    from dcn_hip import backbone as _bb
    arch, bw, (N, H, W), D = "Resnet18_8s", 8, (2, 32, 40), 3
    m, _ = _pair(arch, D, bw)
    m.train()
    x = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(1))
    plan = _bb.get_plan(arch, bw, N, H, W, D)
    plan.profile_begin()
    m(x).sum().backward()
    prof = plan.profile_end()
    assert set(prof) == set(_bb.Plan.PROFILE_CATEGORIES)
    n_bn, n_conv = len(plan.bn_names), len(plan.bn_names) + 1            # every convolution but the scoring layer has a batch norm
    assert prof["bn_finalize"][1] == 2 * n_bn                             # forward + backward finalize per batch norm
    assert prof["bn_bwd_reduce"][1] == n_bn and prof["bn_bwd_apply"][1] == n_bn
    # one apply pass per batch norm, except the stem's (applied inside the max pool) and the three downsample branches'
    # (folded into their block's last pass)
    assert prof["bn_apply"][1] == n_bn - 1 - 3
    assert prof["conv_gemm"][1] == n_conv + (n_conv - 1)                  # forward of every convolution + dgrad of all but the stem
    assert prof["conv_wgrad"][1] == n_conv
    assert abs(prof["conv_gemm"][2] + prof["conv_wgrad"][2] - (3 * plan.forward_flops - 2.0 * N * (H // 2) * (W // 2) * bw * 147)) \
        < 1e-6 * plan.forward_flops                                      # 3 x forward - the stem's missing dgrad
    for k in ("bn_apply", "bn_bwd_reduce", "bn_bwd_apply", "resample"):
        assert prof[k][2] > 0, k
    assert prof["resample"][1] == 1 + 2 + 1 + 2   # input layout, max pool forward / backward, upsample forward, upsample backward (2 passes)
    assert prof["other"][1] >= 6
    # nothing is recorded outside a begin / end window
    m(x).sum().backward()
    plan.profile_begin()
    assert all(v[1] == 0 for v in plan.profile_end().values())


def test_product_state_dict_layout_equals_the_committed_fixture(conv_mode):
    """Keys, order and shapes of the product module's checkpoint == tests/golden/resnet34_8s_d3_state_dict_layout.txt (the
    fixture tests/test_oracle.py holds the oracle and the second statement of the architecture against)."""
    if conv_mode != "f16x3":
        pytest.skip("layout does not depend on the arithmetic")
    import os
    from pytorch_segmentation_detection.models import resnet_dilated as prod
    want = []
    for line in open(os.path.join(os.path.dirname(__file__), "golden", "resnet34_8s_d3_state_dict_layout.txt")):
        if not line.startswith("#") and line.strip():
            k, shp = line.split()
            want.append((k, () if shp == "scalar" else tuple(int(v) for v in shp.split("x"))))
    got = [(k, tuple(v.shape)) for k, v in prod.Resnet34_8s(num_classes=3).state_dict().items()]
    assert got == want


def test_bn_passes_walked_back_to_front_are_bit_identical(dcn_env, conv_mode):
    """DCN_BN_REVERSE: the apply passes of the batch norm (forward and backward) walk their tensors back to front -- a pure
    re-ordering of independent elements: outputs, running statistics and gradients equal bit for bit (two groups included)."""
    m, _ = _pair("Resnet18_8s", 3, 8)
    m2 = copy.deepcopy(m)
    g = torch.Generator().manual_seed(2)
    xa, xb = torch.randn(1, 3, 64, 64, generator=g), torch.randn(1, 3, 64, 64, generator=g)
    gy = torch.randn(1, 3, 64, 64, generator=g)
    outs = []
    for net, rev in ((m, 0), (m2, 3)):
        dcn_env(DCN_BN_REVERSE=rev, DCN_BN_NT=rev)   # (and DCN_BN_NT: non-temporal loads of the last-use tensors)
        net.train()
        ya, yb = net.forward_pair(xa, xb)
        y1 = net(xa)
        ((ya * gy).sum() + (yb * gy).sum() + (y1 * gy).sum()).backward()
        outs.append((ya.detach(), yb.detach(), y1.detach()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    for (k, p), p2 in zip(m.named_parameters(), m2.parameters()):
        assert torch.equal(p.grad, p2.grad), k
    for (k, b), b2 in zip(m.named_buffers(), m2.buffers()):
        assert torch.equal(b, b2), k


@pytest.mark.parametrize("cap", [128])
def test_bn_backward_reduction_with_wide_workgroups(cap, dcn_env, conv_mode):
    """DCN_BN_REDUCE_WIDE: workgroups of the batch-norm backward reduction that cover 128 ... 512 channels of a row instead of
    64 -- another partition of the same sums: gradients equal to round-off of the default partition."""
    if conv_mode != "f16x3":
        pytest.skip("the pass does not depend on the convolution arithmetic")
    m, _ = _pair("Resnet18_8s", 3, 64)     # 64 / 128 / 256 / 512 channels: workgroups of 64 (stem: one group), 32, 64 and 128 quads
    m2 = copy.deepcopy(m)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 3, 32, 32, generator=g)
    gy = torch.randn(1, 3, 32, 32, generator=g)
    for net, c in ((m, 16), (m2, cap)):
        dcn_env(DCN_BN_REDUCE_WIDE=c)
        net.train()
        (net(x) * gy).sum().backward()
    for (k, p), p2 in zip(m.named_parameters(), m2.parameters()):
        assert rel_err(p2.grad, p.grad) < 2e-5, (k, rel_err(p2.grad, p.grad))


def test_weight_images_made_on_the_side_stream_are_the_same_images(dcn_env, conv_mode):
    """DCN_WSPLIT_OVERLAP: the forward call makes all weight images -- its own and the channel-transposed ones of its backward
    pass, the latter into the saved arena -- on the side stream during the stem; the backward pass then makes none.  Same bits
    as the images made on the caller's stream in front of each pass (NaN-poisoned arenas: nothing is read before it is written)."""
    if conv_mode != "f16x3":
        pytest.skip("weight images belong to the split-fp16 arithmetic")
    from dcn_hip import backbone as _bb
    m, _ = _pair("Resnet18_8s", 3, 8)
    m2 = copy.deepcopy(m)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, 32, 40, generator=g)
    gy = torch.randn(2, 3, 32, 40, generator=g)
    _bb.POISON_ARENAS = True
    try:
        outs = []
        for net, on in ((m, 1), (m2, 0)):
            dcn_env(DCN_WSPLIT_OVERLAP=on)
            net.train()
            y = net(x)
            (y * gy).sum().backward()
            outs.append(y.detach())
    finally:
        _bb.POISON_ARENAS = False
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
    for (k, p), p2 in zip(m.named_parameters(), m2.parameters()):
        assert bool(torch.isfinite(p.grad).all()) and torch.equal(p.grad, p2.grad), k
