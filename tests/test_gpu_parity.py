"""Parity tests proper: the gfx950 library on a real MI355X, through the C ABI, against
 (1) the golden vectors made from the reference's own loss source   (tests/golden/loss_ref_*.npz)
 (2) the oracle (CPU, fp32) on the same seeded inputs at sizes it finishes in seconds
 (3) the committed config-1 oracle fixture                           (tests/golden/config1_oracle.npz)
 (4) size-independent properties at BASELINE's full sizes (determinism, a<->b symmetry, zero-sum gradients).
Tolerance: 1e-4 relative on descriptor maps and loss (BASELINE.json north_star); integer outputs exact."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import lists_from_golden, load_golden_loss, rel_err, use_gfx950_library

pytestmark = pytest.mark.gpu
GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
GOLDENS = sorted(glob.glob(os.path.join(GOLDEN_DIR, "loss_ref_*.npz")))
TOL = 1e-4


@pytest.fixture(scope="module")
def L():
    lib = use_gfx950_library()
    assert torch.cuda.is_available()
    info = lib.library_info()
    assert info["path"].endswith("libdcn_hip.so") and "gfx950" in info["version"] and not info["hostemu"]
    return lib


def _tuple(Ld, dev):
    keys = ("matches_a", "matches_b", "masked_non_matches_a", "masked_non_matches_b", "background_non_matches_a",
            "background_non_matches_b", "blind_non_matches_a", "blind_non_matches_b")
    return tuple(Ld[k].to(dev) for k in keys)


# ------------------------------------------------------------------------------------------------ loss (K9)
@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[9:-4] for p in GOLDENS])
def test_loss_matches_reference_goldens(L, path):
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    z, cfg = load_golden_loss(path)
    A = torch.tensor(z["A"], device="cuda", requires_grad=True)
    B = torch.tensor(z["B"], device="cuda", requires_grad=True)
    pcl = PixelwiseContrastiveLoss([int(z["H"]), int(z["W"])], cfg)
    out = loss_composer.get_loss(pcl, torch.tensor([int(z["match_type"])]), A, B, *lists_from_golden(z, "cuda"))
    got = np.array([float(o.sum().item()) for o in out])
    np.testing.assert_allclose(got, z["out"], rtol=5e-6, atol=1e-9)
    out[0].backward()
    assert rel_err(A.grad.cpu(), z["gradA"]) < 1e-5 and rel_err(B.grad.cpu(), z["gradB"]) < 1e-5


def test_loss_building_blocks_on_gpu(L):
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss as PCL
    z, cfg = load_golden_loss(os.path.join(GOLDEN_DIR, "loss_ref_within_d16.npz"))
    A = torch.tensor(z["A"], device="cuda", requires_grad=True)
    B = torch.tensor(z["B"], device="cuda")
    ma, mb, ka, kb = [torch.tensor(z[k], device="cuda") for k in ("matches_a", "matches_b", "masked_a", "masked_b")]
    ml, _, _ = PCL.match_loss(A, B, ma, mb)
    np.testing.assert_allclose(ml.item(), z["f6_match_loss"], rtol=5e-6)
    vec, hn, _, _ = PCL.non_match_descriptor_loss(A, B, ka, kb, M=cfg["M_masked"])
    np.testing.assert_allclose(vec.detach().cpu().numpy(), z["f7_vec"], rtol=1e-5, atol=1e-8)
    assert hn == int(z["f7_hard"])


def test_loss_config3_scale_vs_oracle_and_properties(L):
    """BASELINE config 3 list sizes (10 000 match + 50 000 + 50 000 non-match per pair, D = 16, HW = 307 200),
    2 pairs: oracle comparison + symmetry + zero-sum gradient + run-to-run determinism of the forward."""
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import loss_oracle, synth
    H, W, D, B = 480, 640, 16, 2
    g = torch.Generator().manual_seed(3)
    A = ((torch.rand(B, H * W, D, generator=g) * 2 - 1) * 0.12)
    Bd = ((torch.rand(B, H * W, D, generator=g) * 2 - 1) * 0.12)
    lists = synth.make_index_lists(B, H * W, 10000, 50000, 50000, g)
    Ac, Bc = A.cuda().requires_grad_(True), Bd.cuda().requires_grad_(True)
    pcl = PixelwiseContrastiveLoss([H, W], synth.LOSS_CONFIG)
    tup = [_tuple(Ld, "cuda") for Ld in lists]
    loss, terms, hard = loss_composer.get_loss_batched(pcl, 0, Ac, Bc, tup)
    loss.backward()
    loss_again = loss_composer.get_loss_batched(pcl, 0, Ac.detach(), Bc.detach(), tup)[0]
    assert loss_again.item() == loss.item(), "forward reduction must be run-to-run deterministic"
    A2, B2 = A.clone().requires_grad_(True), Bd.clone().requires_grad_(True)
    opcl = loss_oracle.PixelwiseContrastiveLoss([H, W], synth.LOSS_CONFIG)
    tot = 0
    for b in range(B):
        out = loss_oracle.get_loss(opcl, torch.tensor([0]), A2[b:b + 1], B2[b:b + 1], *_tuple(lists[b], "cpu"))
        tot = tot + out[0]
        np.testing.assert_allclose(terms[b].cpu().numpy(), [float(o.sum()) for o in out], rtol=TOL)
        hk = int(hard[b, 1])
        vec, hn, _, _ = loss_oracle.PixelwiseContrastiveLoss.non_match_descriptor_loss(
            A2[b:b + 1], B2[b:b + 1], lists[b]["masked_non_matches_a"], lists[b]["masked_non_matches_b"], M=0.5)
        assert abs(hk - hn) <= 2, (hk, hn)   # tie band: pairs with |M - d| ~ 1e-7 may flip
        assert 0 < hk < 50000
    (tot / B).backward()
    assert abs(loss.item() - (tot / B).item()) <= TOL * abs((tot / B).item())
    assert rel_err(Ac.grad.cpu(), A2.grad) < TOL and rel_err(Bc.grad.cpu(), B2.grad) < TOL
    # property: every pair contributes +g to A and -g to B
    assert float((Ac.grad.sum(1) + Bc.grad.sum(1)).abs().max()) < 1e-5 * float(Ac.grad.abs().sum(1).max())
    # property: swapping the roles of a and b leaves the loss unchanged
    swapped = [tuple(t[i ^ 1] for i in range(8)) for t in tup]
    loss_sw = loss_composer.get_loss_batched(pcl, 0, Bc.detach(), Ac.detach(), swapped)[0]
    assert abs(loss_sw.item() - loss.item()) <= 1e-6 * abs(loss.item())


# ------------------------------------------------------------------------------------------------ conv kernels
GPU_CONV_CASES = [
    (2, 9, 7, 4, 12, 7, 2, 3, 1),
    (1, 12, 10, 16, 24, 3, 2, 1, 1),
    (1, 10, 12, 20, 136, 3, 1, 2, 2),
    (2, 30, 40, 64, 64, 3, 1, 1, 1),       # layer1-like
    (1, 60, 80, 256, 256, 3, 1, 2, 2),     # layer3 shape
    (1, 60, 80, 512, 512, 3, 1, 4, 4),     # layer4 shape: 59 % of the network's FLOPs
    (1, 60, 80, 256, 512, 1, 1, 0, 1),     # layer4 downsample
    (1, 120, 160, 64, 128, 1, 2, 0, 1),    # layer2 downsample, stride 2
]


@pytest.mark.parametrize("case", GPU_CONV_CASES, ids=[str(c) for c in GPU_CONV_CASES])
def test_conv_kernels_vs_torch_cpu(L, case):
    lib = L.get()
    n, hin, win, cin, cout, k, stride, pad, dil = case
    hout = (hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wout = (win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    d = L.ConvDesc(n, hin, win, cin, hout, wout, cout, k, k, stride, pad, dil, cout)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, cin, hin, win, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)).requires_grad_(True)
    dout = torch.randn(n, hout, wout, cout, generator=g)
    ref = F.conv2d(x, w, None, stride, pad, dil)
    ref.backward(dout.permute(0, 3, 1, 2))
    refn = ref.detach().permute(0, 2, 3, 1)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    wg = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.full((n, hout, wout, cout), float("nan"), device="cuda")
    mt = lib.dcn_conv_num_mtiles(ctypes.byref(d))
    part = torch.full((mt, 3, cout), float("nan"), device="cuda")
    st = L.stream_ptr()
    ws_f = torch.empty(max(lib.dcn_conv_gemm_workspace(ctypes.byref(d), 0), 4) // 4, device="cuda")
    ws_d = torch.empty(max(lib.dcn_conv_gemm_workspace(ctypes.byref(d), 1), 4) // 4, device="cuda")
    assert lib.dcn_conv_forward(ctypes.byref(d), L.ptr(xg), L.ptr(wg), None, L.ptr(out), L.ptr(part), L.ptr(ws_f), st) == 0
    assert rel_err(out.cpu(), refn) < 1e-5
    assert rel_err(part.sum(0)[0].cpu(), refn.sum((0, 1, 2))) < 2e-5
    assert rel_err(part.sum(0)[1].cpu(), (refn ** 2).sum((0, 1, 2))) < 2e-5
    wt = torch.empty(cin, k * k, cout, device="cuda")
    assert lib.dcn_transpose_weight(L.ptr(wg), L.ptr(wt), cout, k * k, cin, cout, st) == 0
    add = torch.randn(n, hin, win, cin, generator=g)
    din = torch.full((n, hin, win, cin), float("nan"), device="cuda")
    dg = dout.cuda()
    addg = add.cuda()
    assert lib.dcn_conv_dgrad(ctypes.byref(d), L.ptr(dg), L.ptr(wt), L.ptr(addg), L.ptr(din), L.ptr(ws_d), st) == 0
    assert rel_err(din.cpu(), x.grad.permute(0, 2, 3, 1) + add) < 1e-5
    dw = torch.full((cout, k, k, cin), float("nan"), device="cuda")
    slab = torch.empty(max(lib.dcn_conv_wgrad_workspace(ctypes.byref(d)), 4) // 4, device="cuda")
    assert lib.dcn_conv_wgrad(ctypes.byref(d), L.ptr(xg), L.ptr(dg), L.ptr(dw), L.ptr(slab), st) == 0
    assert rel_err(dw.cpu(), w.grad.permute(0, 2, 3, 1)) < 1e-5
    # deterministic: same bits on a second run (fixed-order split reduction, no float atomics)
    dw2 = torch.empty_like(dw)
    assert lib.dcn_conv_wgrad(ctypes.byref(d), L.ptr(xg), L.ptr(dg), L.ptr(dw2), L.ptr(slab), st) == 0
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("case", GPU_CONV_CASES, ids=[str(c) for c in GPU_CONV_CASES])
def test_conv_f16x3_kernels_vs_torch_cpu(L, case):
    """The split-fp16 kernels (fp16 MFMA pipe, hi/lo operands) against the same torch CPU fp32 convolution and the same
    1e-5 bound as the fp32 MFMA kernels; gradients of magnitude 1e-6 exercise the abs-max driven pre-scale."""
    lib = L.get()
    n, hin, win, cin, cout, k, stride, pad, dil = case
    hout = (hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wout = (win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    d = L.ConvDesc(n, hin, win, cin, hout, wout, cout, k, k, stride, pad, dil, cout)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, cin, hin, win, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)).requires_grad_(True)
    dout = torch.randn(n, hout, wout, cout, generator=g) * 1e-6
    ref = F.conv2d(x, w, None, stride, pad, dil)
    ref.backward(dout.permute(0, 3, 1, 2))
    refn = ref.detach().permute(0, 2, 3, 1)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    wg = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    st = L.stream_ptr()
    K, Kt = k * k * cin, k * k * cout
    wh = torch.empty(cout, lib.dcn_f16_kpad(K), dtype=torch.float16, device="cuda"); wl = torch.empty_like(wh)
    assert lib.dcn_split_rows_f16(L.ptr(wg), L.ptr(wh), L.ptr(wl), cout, K, 64.0, st) == 0
    out = torch.full((n, hout, wout, cout), float("nan"), device="cuda")
    part = torch.full((lib.dcn_conv_num_mtiles_f16(ctypes.byref(d)), 3, cout), float("nan"), device="cuda")
    ws_f = torch.empty(max(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 0), 4) // 4, device="cuda")
    ws_d = torch.empty(max(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 1), 4) // 4, device="cuda")
    assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(xg), None, L.ptr(wh), L.ptr(wl), 64.0, None, L.ptr(out),
                                    L.ptr(part), L.ptr(ws_f), st) == 0
    assert rel_err(out.cpu(), refn) < 1e-5
    for big in (2.0e5, 1.0e-7):   # activations outside fp16's range: the operand pre-scale from the abs-max scalar
        xb = xg * big
        xmax = xb.abs().max().reshape(1)
        outb = torch.full((n, hout, wout, cout), float("nan"), device="cuda")
        assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(xb), L.ptr(xmax), L.ptr(wh), L.ptr(wl), 64.0, None,
                                        L.ptr(outb), None, L.ptr(ws_f), st) == 0
        assert rel_err(outb.cpu(), refn * big) < 1e-5, big
    assert rel_err(part.sum(0)[0].cpu(), refn.sum((0, 1, 2))) < 2e-5
    assert rel_err(part.sum(0)[1].cpu(), (refn ** 2).sum((0, 1, 2))) < 2e-5
    wt = torch.empty(cin, k * k, cout, device="cuda")
    assert lib.dcn_transpose_weight(L.ptr(wg), L.ptr(wt), cout, k * k, cin, cout, st) == 0
    wth = torch.empty(cin, lib.dcn_f16_kpad(Kt), dtype=torch.float16, device="cuda"); wtl = torch.empty_like(wth)
    assert lib.dcn_split_rows_f16(L.ptr(wt), L.ptr(wth), L.ptr(wtl), cin, Kt, 64.0, st) == 0
    dg = dout.cuda()
    amax = dg.abs().max().reshape(1)
    add = torch.randn(n, hin, win, cin, generator=g) * 1e-6
    addg = add.cuda()
    din = torch.full((n, hin, win, cin), float("nan"), device="cuda")
    assert lib.dcn_conv_dgrad_f16(ctypes.byref(d), L.ptr(dg), L.ptr(wth), L.ptr(wtl), 64.0, L.ptr(amax), L.ptr(addg),
                                  L.ptr(din), L.ptr(ws_d), st) == 0
    assert rel_err(din.cpu(), x.grad.permute(0, 2, 3, 1) + add) < 1e-5
    dw = torch.full((cout, k, k, cin), float("nan"), device="cuda")
    slab = torch.empty(max(lib.dcn_conv_wgrad_workspace_f16(ctypes.byref(d)), 4) // 4, device="cuda")
    M = n * hout * wout
    xs = torch.empty(xg.numel(), dtype=torch.float32, device="cuda")                  # split tensors: fp32-sized byte buffers
    assert lib.dcn_split_act_f16(L.ptr(xg), L.ptr(xs), xg.numel(), st) == 0
    dq = torch.empty(lib.dcn_grad_blocked_bytes(M, cout) // 4, dtype=torch.float32, device="cuda")
    assert lib.dcn_split_grad_blocked_f16(L.ptr(dg), M, cout, L.ptr(amax), L.ptr(dq), st) == 0
    wg_args = (ctypes.byref(d), L.ptr(xs), 0, None, L.ptr(dq), L.ptr(amax))
    assert lib.dcn_conv_wgrad_f16(*wg_args, L.ptr(dw), L.ptr(slab), st) == 0
    assert rel_err(dw.cpu(), w.grad.permute(0, 2, 3, 1)) < 1e-5
    dw2 = torch.empty_like(dw)
    assert lib.dcn_conv_wgrad_f16(*wg_args, L.ptr(dw2), L.ptr(slab), st) == 0
    dw3 = torch.empty_like(dw)   # activation operand = the fp32 tensor, split on the fly: same bits
    assert lib.dcn_conv_wgrad_f16(ctypes.byref(d), L.ptr(xg), 1, None, L.ptr(dq), L.ptr(amax), L.ptr(dw3), L.ptr(slab), st) == 0
    assert torch.equal(dw, dw3)
    xb = xg * 2.0e5
    xmax = xb.abs().max().reshape(1)
    dw4 = torch.empty_like(dw)
    assert lib.dcn_conv_wgrad_f16(ctypes.byref(d), L.ptr(xb), 1, L.ptr(xmax), L.ptr(dq), L.ptr(amax), L.ptr(dw4), L.ptr(slab), st) == 0
    assert rel_err(dw4.cpu(), w.grad.permute(0, 2, 3, 1) * 2.0e5) < 1e-5
    assert torch.equal(dw, dw2)


@pytest.fixture(params=["f16x3", "fp32"])
def conv_mode(request):
    """Backbone tests run in both convolution arithmetics against the SAME tolerances (include/dcn_hip.h)."""
    from dcn_hip import backbone
    backbone.set_conv_mode(request.param)
    yield request.param
    backbone.set_conv_mode(None)



# ------------------------------------------------------------------------------------------------ backbone + step
def _dcn_and_oracle(arch, D, H, W):
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    from oracle import resnet_dilated_oracle
    cfg = {"descriptor_dimension": D, "image_width": W, "image_height": H,
           "backbone": {"model_class": "Resnet", "resnet_name": arch}}
    dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False)
    o = resnet_dilated_oracle.build(arch, D, seed=0)
    dcn.fcn.load_state_dict(o.state_dict())
    assert next(dcn.parameters()).is_cuda and dcn.training
    return dcn, o


def _gpu_step(dcn, img_a, img_b, lists, B):
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import synth
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    ya = dcn.forward(img_a.cuda())
    yb = dcn.forward(img_b.cuda())
    pa, pb = dcn.process_network_output(ya, B), dcn.process_network_output(yb, B)
    assert pa.is_contiguous()
    loss, terms, hard = loss_composer.get_loss_batched(pcl, 0, pa, pb, [_tuple(Ld, "cuda") for Ld in lists])
    return loss, terms, hard, ya, yb


def test_config1_full_size_vs_live_oracle(L, conv_mode):
    """Same configuration against the oracle run live on the host cores (takes a few seconds)."""
    from oracle import step as ostep, synth
    c = synth.CONFIGS[1]
    dcn, o = _dcn_and_oracle(c["backbone"], c["D"], c["H"], c["W"])
    img_a, img_b, lists = synth.make_batch(c["B"], c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=1)
    o.train()
    loss_o, terms_o, da_o, db_o = ostep.forward_loss(o, img_a, img_b, lists, synth.LOSS_CONFIG)
    loss_o.backward()
    loss, terms, hard, ya, yb = _gpu_step(dcn, img_a, img_b, lists, c["B"])
    loss.backward()
    assert rel_err(ya.detach().cpu(), da_o) < TOL and rel_err(yb.detach().cpu(), db_o) < TOL
    assert abs(loss.item() - loss_o.item()) <= TOL * abs(loss_o.item())
    # Gradients are NOT judged float32-against-float32 tensor by tensor: both sides carry their own ill-conditioning noise
    # (the float32 oracle is up to 3.6e-2 of max|g| away from its float64 self at this size).  tests/test_gpu_configs.py
    # judges them against the FLOAT64 oracle.  Here only: the two float32 results are within the sum of two such errors of
    # each other, in the r.m.s. over the tensors (yard-sticks from the committed fixture of the same configuration).
    z = np.load(os.path.join(GOLDEN_DIR, "config1_oracle.npz"))
    yard = {str(k): float(e) / float(n) for k, e, n in zip(z["grad_names"], z["grad_err32_l2"], z["grad_norms64"])}
    rel, ref = [], []
    for (k, p), (_, po) in zip(dcn.fcn.named_parameters(), o.named_parameters()):
        if k.endswith("fc.bias"):
            continue
        rel.append(float((p.grad.cpu() - po.grad).norm() / po.grad.norm()))
        ref.append(yard[k])
    rms = lambda v: float(np.sqrt(np.mean(np.square(v))))
    assert rms(rel) <= 2.5 * rms(ref), (rms(rel), rms(ref))
    assert max(rel) <= 3.0 * max(ref), (max(rel), max(ref))


def test_batched_step_small_images_vs_oracle(L, conv_mode):
    """B = 3 pairs (BN statistics over 3 images per call), D = 16, 96x128 images, with Adam."""
    from oracle import step as ostep, synth
    H, W, D, B = 96, 128, 16, 3
    dcn, o = _dcn_and_oracle("Resnet34_8s", D, H, W)
    img_a, img_b, lists = synth.make_batch(B, H, W, 300, 200, 200, seed=4)
    opt_o = torch.optim.Adam(o.parameters(), lr=1e-4, weight_decay=1e-4)
    from dcn_hip.optim import Adam
    opt = Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4)
    o.train()
    loss_o, _, da_o, _ = ostep.train_step(o, opt_o, img_a, img_b, lists, synth.LOSS_CONFIG)
    opt.zero_grad()
    loss, terms, hard, ya, yb = _gpu_step(dcn, img_a, img_b, lists, B)
    loss.backward()
    opt.step()
    assert rel_err(ya.detach().cpu(), da_o) < TOL
    assert abs(loss.item() - loss_o.item()) <= TOL * abs(loss_o.item())
    # Adam's first update is lr * sign(g) for every element: where |g| is below the float32 noise of the gradient the
    # two implementations may step in opposite directions (2 * lr apart) -- so parameters agree to 2.5 * lr, not 1e-4
    for (k, p), (_, po) in zip(dcn.fcn.named_parameters(), o.named_parameters()):
        assert float((p.detach().cpu() - po).abs().max()) < 2.5e-4, k
    # second iteration from IDENTICAL parameters (running statistics included): forward + loss must agree again
    dcn.fcn.load_state_dict(o.state_dict())
    loss_o2, _, da_o2, _ = ostep.forward_loss(o, img_a, img_b, lists, synth.LOSS_CONFIG)
    loss2, _, _, ya2, _ = _gpu_step(dcn, img_a, img_b, lists, B)
    assert rel_err(ya2.detach().cpu(), da_o2.detach()) < TOL
    assert abs(loss2.item() - loss_o2.item()) <= TOL * abs(loss_o2.item())


def test_resnet50_8s_forward_backward_vs_oracle(L, conv_mode):
    """Bottleneck family (BASELINE config 5's backbone) at a small size, D = 32."""
    import copy
    H, W, D = 128, 160, 32    # 2 x 16 x 20 = 640 samples per batch-norm channel in layer3/4
    dcn, o = _dcn_and_oracle("Resnet50_8s", D, H, W)
    o64 = copy.deepcopy(o).double()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, H, W, generator=g)
    gy = torch.randn(2, D, H, W, generator=g)
    o.train(); o64.train()
    y = dcn.forward(x.cuda())
    yo, y64 = o(x), o64(x.double())
    assert rel_err(y.detach().cpu(), y64) < 3 * rel_err(yo, y64) + 1e-5
    (y * gy.cuda()).sum().backward(); (yo * gy).sum().backward(); (y64 * gy.double()).sum().backward()
    import parity_common as pc
    pc.assert_as_accurate_as_float32(pc.grad_error_stats(dcn.fcn.named_parameters(), o.parameters(), o64.parameters(), ()),
                                     factor=2.0, floor=5e-4)   # (small case: 2 images, few hundred pixels per BN channel)


def test_config2_full_size_properties(L, conv_mode):
    """BASELINE config 2 (B = 4 pairs, 640x480): too slow for the CPU oracle in a unit test, so properties:
    bitwise run-to-run determinism of the forward, finite outputs, eval-mode idempotence, gradient flat buffer."""
    from dcn_hip.distributed import FlatGradients
    from oracle import synth
    c = synth.CONFIGS[2]
    dcn, _ = _dcn_and_oracle(c["backbone"], c["D"], c["H"], c["W"])
    grads = FlatGradients(dcn)
    img_a, img_b, lists = synth.make_batch(c["B"], c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=1)
    dcn.eval()
    with torch.no_grad():
        y1 = dcn.forward(img_a.cuda())
        y2 = dcn.forward(img_a.cuda())
    assert torch.equal(y1, y2) and torch.isfinite(y1).all()
    dcn.train()
    loss, terms, hard, ya, yb = _gpu_step(dcn, img_a, img_b, lists, c["B"])
    loss.backward()
    assert torch.isfinite(grads.flat).all() and float(grads.flat.abs().max()) > 0
    assert ya.shape == (4, 3, 480, 640) and ya.is_contiguous(memory_format=torch.channels_last)
    assert (hard[:, 1] <= c["Pk"]).all() and (hard[:, 0] == c["Pm"]).all()
    assert abs(float(terms[:, 0].mean()) - loss.item()) <= 1e-6 * abs(loss.item())


def test_best_match_search_full_size(L):
    """100 queries against a 640x480 D=3 descriptor image (evaluation.py:932-950's workload) vs the numpy formula."""
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork as DCN
    H, W, D, Q = 480, 640, 3, 100
    g = torch.Generator().manual_seed(8)
    res_a = torch.randn(H, W, D, generator=g)
    res_b = torch.randn(H, W, D, generator=g)
    pix = torch.stack([torch.randint(0, W, (Q,), generator=g), torch.randint(0, H, (Q,), generator=g)], 1)
    res_b[100, 200] = res_b[400, 17] = res_a[int(pix[0, 1]), int(pix[0, 0])]   # exact tie -> first occurrence wins
    uv, dist, nd = DCN.find_best_matches(pix.cuda(), res_a.cuda(), res_b.cuda(), return_norm_diffs=True)
    uv, dist = uv.cpu(), dist.cpu()
    for i in range(0, Q, 7):
        ref_uv, ref_diff, ref_nd = DCN.find_best_match((int(pix[i, 0]), int(pix[i, 1])), res_a.numpy(), res_b.numpy())
        assert (int(uv[i, 0]), int(uv[i, 1])) == (int(ref_uv[0]), int(ref_uv[1])), i
        np.testing.assert_allclose(dist[i].item(), ref_diff, rtol=1e-5, atol=1e-6)
        if i == 0:
            np.testing.assert_allclose(nd[0].cpu().numpy(), ref_nd, rtol=1e-5, atol=1e-6)
    assert (int(uv[0, 0]), int(uv[0, 1])) == (200, 100)


def test_normalized_descriptor_training_step(L, conv_mode):
    """normalize=True (network.py:256-259) forward + backward on the GPU vs the oracle."""
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    from oracle import resnet_dilated_oracle
    H, W, D = 96, 128, 3
    cfg = {"descriptor_dimension": D, "image_width": W, "image_height": H, "normalize": True}
    dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False)
    o = resnet_dilated_oracle.build("Resnet34_8s", D, seed=0)
    dcn.fcn.load_state_dict(o.state_dict())
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, H, W, generator=g)
    gy = torch.randn(2, D, H, W, generator=g)
    o.train()
    y = dcn.forward(x.cuda())
    ro = o(x)
    yo = ro / torch.norm(ro, 2, 1, keepdim=True)
    # unit vectors are ill-conditioned where ||v|| is small: an error e in v moves v/||v|| by ~e/||v||, so the 1e-4 bound
    # on the raw map (relative to max|v|) becomes 1e-4 * max|v| / ||v|| per pixel
    nv = torch.norm(ro, 2, 1, keepdim=True).detach()
    scaled = ((y.detach().cpu() - yo.detach()).abs() * nv).max() / ro.detach().abs().max()
    assert float(scaled) < 2 * TOL, float(scaled)
    assert float((y.detach().norm(2, 1) - 1).abs().max()) < 1e-5
    import copy
    o64 = copy.deepcopy(o).double()
    for p6 in o64.parameters():
        p6.grad = None
    o64.train()
    r64 = o64(x.double())
    y64 = r64 / torch.norm(r64, 2, 1, keepdim=True)
    (y * gy.cuda()).sum().backward(); (yo * gy).sum().backward(); (y64 * gy.double()).sum().backward()
    # against the float64 oracle, with the float32 oracle's own error as the yard-stick (unit vectors of near-zero raw
    # descriptors make this ill-conditioned: the float32 oracle itself is per cent off on some tensors)
    import parity_common as pc
    pc.assert_as_accurate_as_float32(pc.grad_error_stats(dcn.fcn.named_parameters(), o.parameters(), o64.parameters(), ()),
                                     factor=2.0, floor=1e-3)


def test_forward_pair_equals_two_forward_calls_full_size(L, conv_mode):
    """config 1 at full size: forward_pair(img_a, img_b) (one grouped launch sequence over both image batches) against two
    forward calls of the same network -- descriptors, loss, every parameter gradient and the BN running statistics."""
    import copy
    from oracle import synth
    c = synth.CONFIGS[1]
    dcn, _ = _dcn_and_oracle(c["backbone"], c["D"], c["H"], c["W"])
    dcn2 = copy.deepcopy(dcn)
    img_a, img_b, lists = synth.make_batch(2, c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=4)
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    tup = [_tuple(Ld, "cuda") for Ld in lists]
    ya, yb = dcn.forward_pair(img_a.cuda(), img_b.cuda())
    za, zb = dcn2.forward(img_a.cuda()), dcn2.forward(img_b.cuda())
    # (not bit-identical: 2N images pick other tile shapes / stream-K splits than N, i.e. another fp32 summation order;
    #  through 36 conv+BN layers that is ~1e-5 on the descriptors -- the same size as either result's distance from float64)
    assert rel_err(ya.detach().cpu(), za.detach().cpu()) < 5e-5 and rel_err(yb.detach().cpu(), zb.detach().cpu()) < 5e-5
    l1 = loss_composer.get_loss_batched(pcl, 0, dcn.process_network_output(ya, 2), dcn.process_network_output(yb, 2), tup)[0]
    l2 = loss_composer.get_loss_batched(pcl, 0, dcn2.process_network_output(za, 2), dcn2.process_network_output(zb, 2), tup)[0]
    assert abs(l1.item() - l2.item()) <= TOL * abs(l2.item())
    l1.backward(); l2.backward()
    for (k, p), p2 in zip(dcn.fcn.named_parameters(), dcn2.fcn.parameters()):
        if k.endswith("fc.bias"):
            continue
        l2n = float((p.grad - p2.grad).norm() / p2.grad.norm())   # ill-conditioned (ReLU kinks): same bound as the live-oracle test
        assert l2n < 2e-2, (k, l2n)
    for (k, b), b2 in zip(dcn.fcn.named_buffers(), dcn2.fcn.buffers()):
        assert rel_err(b.float().cpu(), b2.float().cpu()) < 1e-5, k


def test_triplet_loss_kernel_vs_reference_golden_and_oracle(L):
    """pcl.py:104-129 on the GPU: the reference's own golden value (D = 16 case), then a 640x480-sized random case
    (100 k triplets, 10 non-matches per match) against the oracle, values and gradients."""
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss as PCL
    from oracle import loss_oracle
    path = [p for p in GOLDENS if "triplet" in np.load(p).files][0]
    z = np.load(path)
    t = lambda k: torch.tensor(z[k])
    A = t("A").cuda().requires_grad_(True)
    B = t("B").cuda().requires_grad_(True)
    trip = PCL.get_triplet_loss(A, B, t("matches_a").cuda(), t("matches_b").cuda(), t("triplet_non_matches_a").cuda(),
                                t("masked_b").cuda(), 0.1)
    np.testing.assert_allclose(trip.item(), z["triplet"], rtol=1e-6)
    g = torch.Generator().manual_seed(5)
    HW, D, Pm, mult = 640 * 480, 3, 10000, 10
    Ac = (torch.rand(1, HW, D, generator=g) - 0.5)
    Bc = (torch.rand(1, HW, D, generator=g) - 0.5)
    ma = torch.randint(0, HW, (Pm,), generator=g)
    mb = torch.randint(0, HW, (Pm,), generator=g)
    na = ma.repeat_interleave(mult)
    nb = torch.randint(0, HW, (Pm * mult,), generator=g)
    Ag, Bg = Ac.cuda().requires_grad_(True), Bc.cuda().requires_grad_(True)
    lg = PCL.get_triplet_loss(Ag, Bg, ma.cuda(), mb.cuda(), na.cuda(), nb.cuda(), 0.1)
    Ao, Bo = Ac.clone().requires_grad_(True), Bc.clone().requires_grad_(True)
    lo = loss_oracle.PixelwiseContrastiveLoss.get_triplet_loss(Ao, Bo, ma, mb, na, nb, 0.1)
    assert abs(lg.item() - lo.item()) <= TOL * abs(lo.item())
    lg.backward(); lo.backward()
    assert rel_err(Ag.grad.cpu(), Ao.grad) < TOL and rel_err(Bg.grad.cpu(), Bo.grad) < TOL


# ------------------------------------------------------------------------------------------------ pair generation (8f-2)
CORR_GOLDENS = sorted(glob.glob(os.path.join(GOLDEN_DIR, "corr_ref_*.npz")))


@pytest.mark.parametrize("path", CORR_GOLDENS, ids=[os.path.basename(p)[:-4] for p in CORR_GOLDENS])
def test_pair_generation_vs_reference_goldens(L, path):
    """Device pair generation against the outputs of the reference's own correspondence_finder source: the surviving
    candidates (order and values) and the non-match samples are exact, sub-pixel projections within 3e-4 px."""
    from dcn_hip import pairgen
    from oracle import correspondence_oracle as co
    z = np.load(path)
    dep = lambda a: torch.from_numpy(a.astype(np.uint16).view(np.int16)).cuda()
    ua, va, ub, vb = pairgen.find_correspondences(dep(z["depth_a"]), dep(z["depth_b"]), co.get_default_K_matrix(), z["pose_a"],
                                                  z["pose_b"], torch.tensor(z["cand_u"]).cuda(), torch.tensor(z["cand_v"]).cuda())
    assert np.array_equal(ua.cpu().numpy(), z["uv_a_u"]) and np.array_equal(va.cpu().numpy(), z["uv_a_v"])
    np.testing.assert_allclose(ub.cpu().numpy(), z["uv_b_u"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(vb.cpu().numpy(), z["uv_b_v"], rtol=0, atol=3e-4)
    H, W = z["depth_a"].shape
    n = z["non_u"].size
    rand = torch.tensor(z["rand"]).cuda()
    if z["mask"].size:
        lst, cnt = pairgen.mask_nonzero(torch.tensor(z["mask"]).cuda())
        u, v = pairgen.sample_pixels(rand, n, W, H, lst, cnt)
    else:
        u, v = pairgen.sample_pixels(rand, n, W, H)
    assert np.array_equal(u.cpu().view(z["non_u"].shape).numpy(), z["non_u"])
    assert np.array_equal(v.cpu().view(z["non_v"].shape).numpy(), z["non_v"])


def test_pair_generation_reference_api_full_size(L):
    """The mirrored correspondence_finder API with its own random candidates at the training configuration's sizes
    (10 000 attempts, 150 non-matches per match): every returned match must satisfy the oracle's geometric filter, the
    masked candidates lie on the mask, non-matches lie on (off) the mask."""
    from dense_correspondence.correspondence_tools import correspondence_finder as cf
    from oracle import correspondence_oracle as co
    z = np.load(CORR_GOLDENS[1])
    H, W = z["depth_a"].shape
    mask = z["mask"]
    torch.manual_seed(0)
    uv_a, uv_b = cf.batch_find_pixel_correspondences(z["depth_a"], z["pose_a"], z["depth_b"], z["pose_b"], num_attempts=10000,
                                                     img_a_mask=mask)
    assert uv_a is not None and 1000 < uv_a[0].numel() <= 10000
    ua, va = uv_a[0].cpu(), uv_a[1].cpu()
    assert bool((torch.tensor(mask)[va, ua] != 0).all())
    oa, ob = co.find_correspondences_for_candidates(z["depth_a"], z["pose_a"], z["depth_b"], z["pose_b"], ua, va)
    assert oa[0].numel() == ua.numel()          # the oracle keeps every one of them
    np.testing.assert_allclose(uv_b[0].cpu().numpy(), ob[0].numpy(), rtol=0, atol=3e-4)
    nm = cf.create_non_correspondences(uv_b, (H, W), num_non_matches_per_match=150, img_b_mask=torch.tensor(mask))
    assert nm[0].shape == (ua.numel(), 150)
    assert bool((torch.tensor(mask)[nm[1].cpu().long(), nm[0].cpu().long()] != 0).all())
    bg = cf.create_non_correspondences(uv_b, (H, W), num_non_matches_per_match=150, img_b_mask=1 - torch.tensor(mask))
    assert bool((torch.tensor(mask)[bg[1].cpu().long(), bg[0].cpu().long()] == 0).all())
    un = cf.create_non_correspondences(uv_b, (H, W), num_non_matches_per_match=7)
    assert float(un[0].max()) <= W - 1 and float(un[1].max()) <= H - 1 and float(un[0].min()) >= 0


def test_step_does_not_depend_on_workspace_contents(L, conv_mode):
    """Every workspace / saved-arena byte that is read has been written in the same step: poison the allocator's free
    blocks with NaN between two identical steps -- loss and every gradient must come out bit-identical."""
    from dcn_hip import backbone as bb
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import synth
    c = synth.CONFIGS[1]
    dcn, _ = _dcn_and_oracle(c["backbone"], c["D"], c["H"], c["W"])
    img_a, img_b, lists = synth.make_batch(1, c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=2)
    img_a, img_b = img_a.cuda(), img_b.cuda()
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    tup = [_tuple(Ld, "cuda") for Ld in lists]
    params = list(dcn.parameters())

    def step():
        for p in params:
            p.grad = None
        ya, yb = dcn.forward_pair(img_a, img_b)
        loss = loss_composer.get_loss_batched(pcl, 0, dcn.process_network_output(ya, 1), dcn.process_network_output(yb, 1), tup)[0]
        loss.backward()
        return loss.item(), [p.grad.clone() for p in params if not p.shape == torch.Size([c["D"]])]   # (fc.bias: atomics order)

    l0, g0 = step()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    plan = bb.get_plan(c["backbone"], 64, 2, c["H"], c["W"], c["D"], 2)
    junk = [torch.full((n // 4 + 1024,), float("nan"), device="cuda")
            for n in (plan.saved_bytes, plan.workspace_bytes, plan.workspace_bytes, 64 << 20, 16 << 20, 4 << 20, 1 << 20)]
    torch.cuda.synchronize()
    del junk
    l1, g1 = step()
    assert l0 == l1
    assert all(torch.isfinite(b).all() for b in g1)
    # the loss backward scatters with fp32 atomics (order-dependent in the last bits); everything downstream is deterministic
    assert max(rel_err(a.cpu(), b.cpu()) for a, b in zip(g0, g1)) < 1e-5


def test_eval_mode_fused_inference_full_size(L, conv_mode):
    """Inference at 640x480 (running statistics; fused conv + folded BN + residual + ReLU passes in the split-fp16 mode)
    against the oracle in eval mode, after two training-mode passes have moved the running statistics."""
    dcn, o = _dcn_and_oracle("Resnet34_8s", 3, 480, 640)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 3, 480, 640, generator=g)
    o.train()
    with torch.no_grad():
        for _ in range(2):
            dcn.forward(x.cuda()); o(x)
    dcn.eval(); o.eval()
    with torch.no_grad():
        y = dcn.forward(x.cuda()).cpu()
        yo = o(x)
    assert rel_err(y, yo) < TOL
    res = dcn.forward_single_image_tensor(x[0])                    # [H, W, D], network.py:265-299
    assert rel_err(res.cpu(), yo[0].permute(1, 2, 0)) < TOL


def test_match_statistics_vs_reference_golden_and_full_size(L):
    """evaluation.py:1046-1100 on the GPU: the reference's golden (12 matches), then 100 matches at 640x480 against the
    oracle for a few of them."""
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork as DCN
    from oracle import evaluation_oracle as eo
    z = np.load(os.path.join(GOLDEN_DIR, "eval_ref.npz"))
    uv = torch.tensor(z["uv"]).cuda()
    s = DCN.compute_match_statistics(uv, uv, torch.tensor(z["res_a"]).cuda(), torch.tensor(z["res_b"]).cuda(),
                                     torch.tensor(z["mask_b"]).cuda())
    assert np.array_equal(s["uv_b_pred"].cpu().numpy(), z["uv_b_pred"].astype(np.int64))
    assert np.array_equal(s["uv_b_pred_masked"].cpu().numpy(), z["uv_b_pred_masked"].astype(np.int64))
    for name in ("", "_masked"):
        c = s["num_pixels_closer_than_ground_truth" + name].cpu().numpy().astype(np.int64)
        assert np.abs(c - z["num_pixels_closer_than_ground_truth" + name]).max() <= 1
    g = torch.Generator().manual_seed(3)
    H, W, D, Q = 480, 640, 3, 100
    res_a = torch.randn(H, W, D, generator=g)
    res_b = res_a + 0.5 * torch.randn(H, W, D, generator=g)
    mask = torch.zeros(H, W)
    mask[100:400, 150:500] = 1
    uv = torch.stack([torch.randint(150, 500, (Q,), generator=g), torch.randint(100, 400, (Q,), generator=g)], 1)
    s = DCN.compute_match_statistics(uv.cuda(), uv.cuda(), res_a.cuda(), res_b.cuda(), mask.cuda())
    for q in (0, 17, 99):
        o = eo.match_statistics((int(uv[q, 0]), int(uv[q, 1])), (int(uv[q, 0]), int(uv[q, 1])), res_a.numpy(), res_b.numpy(), mask.numpy())
        assert tuple(s["uv_b_pred"][q].tolist()) == tuple(int(x) for x in o["uv_b_pred"])
        assert tuple(s["uv_b_pred_masked"][q].tolist()) == tuple(int(x) for x in o["uv_b_pred_masked"])
        assert abs(int(s["num_pixels_closer_than_ground_truth"][q]) - o["num_pixels_closer_than_ground_truth"]) <= 1
        assert abs(int(s["num_pixels_closer_than_ground_truth_masked"][q]) - o["num_pixels_closer_than_ground_truth_masked"]) <= 1
        assert abs(float(s["norm_diff_descriptor_ground_truth"][q]) - float(o["norm_diff_descriptor_ground_truth"])) < 1e-5


# ------------------------------------------------------------------------------------------------ optimizer step (F16)
def test_adam_step_vs_torch_adam_on_cpu_and_full_model(L):
    """dcn_adam_step against torch.optim.Adam -- the reference's optimizer (training.py:133-145) -- run on the CPU:
    (1) mixed shapes / alignments / layouts over 10 steps, (2) every parameter of Resnet34_8s (21.3 M floats) for 3 steps
    with the engine's channels_last weights; a checksum over all parameters as the size-independent property."""
    from dcn_hip.optim import Adam
    g = torch.Generator().manual_seed(0)
    shapes = [(64,), (7,), (1,), (5, 3), (64, 64, 3, 3), (8200,), (4097,)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    base.append(torch.randn(16, 8, 3, 3, generator=g).contiguous(memory_format=torch.channels_last))
    ref = [torch.nn.Parameter(p.clone(memory_format=torch.preserve_format)) for p in base]
    ours = [torch.nn.Parameter(p.cuda()) for p in base] + [torch.nn.Parameter(torch.randn(4099, generator=g).cuda()[3:])]
    ref.append(torch.nn.Parameter(ours[-1].detach().cpu().clone()))
    assert ours[7].stride() == ref[7].stride() and ours[-1].data_ptr() % 16 != 0
    o, r = Adam(ours, lr=1e-3, weight_decay=1e-4), torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-4, foreach=False)
    for it in range(10):
        for a, b in zip(ours, ref):
            b.grad = torch.empty_like(b).copy_(torch.randn(b.shape, generator=g) * [1.0, 1e-8, 1e3, 1e-3][it % 4])
            a.grad = b.grad.cuda()
        o.step(); r.step()
    for a, b in zip(ours, ref):
        assert rel_err(a.detach().cpu(), b.detach()) < 2e-6
        assert rel_err(o.state[a]["exp_avg_sq"].cpu(), r.state[b]["exp_avg_sq"]) < 2e-6

    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    cfg = {"descriptor_dimension": 3, "image_width": 64, "image_height": 64}
    torch.manual_seed(0)
    dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False).cuda()
    cpu = [torch.nn.Parameter(p.detach().cpu().clone(memory_format=torch.preserve_format)) for p in dcn.parameters()]
    o, r = Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4), torch.optim.Adam(cpu, lr=1e-4, weight_decay=1e-4)
    for it in range(3):
        for a, b in zip(dcn.parameters(), cpu):
            b.grad = torch.empty_like(b).copy_(torch.randn(b.shape, generator=g) * 1e-2)
            a.grad = b.grad.cuda()
            assert a.grad.stride() == a.stride()
        o.step(); r.step()
    tot = sum(float(p.detach().double().sum()) for p in dcn.parameters())
    tot_ref = sum(float(p.detach().double().sum()) for p in cpu)
    assert abs(tot - tot_ref) <= 1e-6 * abs(tot_ref) + 1e-4
    worst = max(rel_err(a.detach().cpu(), b.detach()) for a, b in zip(dcn.parameters(), cpu))
    assert worst < 2e-6, worst


def test_backward_side_stream_equals_serial_schedule_bitwise(L, dcn_env):
    """The weight-gradient GEMMs run on the plan's side stream next to the dgrad / BN-backward chain (two alternating
    gradient images ordered by events).  With a FIXED output gradient (no loss atomics) the backward pass is deterministic,
    so every parameter gradient must be bit-identical to the serial schedule (DCN_BACKWARD_OVERLAP=0, a fresh plan) --
    three times in a row at the full config-2 size, where the two streams really do overlap."""
    from dcn_hip import backbone as bb
    from oracle import synth
    c = synth.CONFIGS[2]
    B = 4
    bb.set_conv_mode("f16x3")
    dcn, _ = _dcn_and_oracle(c["backbone"], c["D"], c["H"], c["W"])
    img_a, img_b, _ = synth.make_batch(B, c["H"], c["W"], 10, 10, 10, seed=5)
    img_a, img_b = img_a.cuda(), img_b.cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    ga = torch.randn(B, c["D"], c["H"], c["W"], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    gb = torch.randn(B, c["D"], c["H"], c["W"], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    params = list(dcn.parameters())

    def grads():
        for p in params:
            p.grad = None
        dcn.train()
        ya, yb = dcn.forward_pair(img_a, img_b)
        torch.autograd.backward([ya, yb], [ga, gb])
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params]

    runs = {}
    for mode in ("1", "0"):
        dcn_env(DCN_BACKWARD_OVERLAP=mode)
        bb._PLANS.clear()                       # the schedule is fixed at a plan's first backward pass
        runs[mode] = [grads() for _ in range(3)]
    bb._PLANS.clear()
    bb.set_conv_mode(None)
    ref = runs["0"][0]
    assert all(torch.isfinite(t).all() for t in ref)
    for mode in ("0", "1"):
        for r in runs[mode]:
            assert all(torch.equal(a, b) for a, b in zip(r, ref)), mode
