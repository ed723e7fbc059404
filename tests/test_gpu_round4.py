"""-m gpu tests of round 4: a multi-step training trajectory against the oracle (both arithmetics, both call patterns),
forward_pair / backward on two base pointers, the launch profile by category on hardware."""
import pytest
import torch

from helpers import use_gfx950_library
import parity_common as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = use_gfx950_library()
    assert torch.cuda.is_available()
    return lib


@pytest.mark.parametrize("mode,separate", [("f16x3", True), ("f16x3", False), ("fp32", True)])
def test_training_trajectory_tracks_the_oracle(L, mode, separate):
    """12 iterations of the reference's loop (zero_grad, forward(img_a), forward(img_b), loss, backward, Adam step, learning-rate
    decay: training.py:325-346, :544-558) from identical weights, the real Resnet34_8s at 64 x 128, B = 2, default
    initialisation, by the product on the MI355X, the float32 oracle and the float64 oracle.  The first step's loss within the
    north star's 1e-4; every later step as close to the float64 trajectory as the float32 oracle is (x3 + 1e-4) -- a fixed
    1e-3 cannot hold for ANY float32 implementation: the float32 oracle itself is 1e-4 off after one Adam step and 1e-2 within
    ten (parity_common.run_trajectory says why).  The eval-mode descriptor map after the last step by the same yard-stick."""
    from dcn_hip import backbone as bb
    bb.set_conv_mode(mode)
    try:
        dcn, o = pc.build_dcn("Resnet34_8s", 3, 64, 128)
        r = pc.run_trajectory(dcn, o, 2, 64, 128, 12, torch.device("cuda"), separate_forwards=separate)
    finally:
        bb.set_conv_mode(None)
    print(mode, "separate" if separate else "pair", "loss %.4f -> %.4f" % (r["loss_o64"][0], r["loss_o64"][-1]),
          "dev product", ["%.1e" % v for v in r["dev_p"]], "dev float32 oracle", ["%.1e" % v for v in r["dev_o32"]],
          "desc %.1e / %.1e" % (r["desc_p"], r["desc_o32"]))
    assert r["loss_o64"][-1] < 0.7 * r["loss_o64"][0], "the synthetic batch must actually train"
    assert abs(r["loss_p"][-1] - r["loss_o64"][-1]) < 0.1 * r["loss_o64"][-1]
    pc.assert_trajectory_as_close_as_float32(r)


def test_forward_pair_takes_two_base_pointers_and_separate_gradients(L):
    """forward_pair(a, b) on two tensors that are NOT adjacent in memory (no concatenated copy is made) == two forward calls;
    the two outputs' gradients reach the engine as two pointers, one of them missing (None -> zero map) included."""
    import copy
    dcn, _ = pc.build_dcn("Resnet34_8s", 3, 64, 128)
    dcn2 = copy.deepcopy(dcn)
    g = torch.Generator().manual_seed(3)
    pool = torch.randn(5, 2, 3, 64, 128, generator=g).cuda()
    xa, xb = pool[0], pool[3] * 1.5 + 0.2
    ya, yb = dcn.fcn.forward_pair(xa, xb)
    za, zb = dcn2.fcn(xa), dcn2.fcn(xb)
    # (same products, other tile shapes and summation orders through 36 layers: the bound of
    #  test_gpu_parity.py::test_forward_pair_equals_two_forward_calls_full_size)
    e_a = float((ya - za).abs().max() / za.abs().max()), float((yb - zb).abs().max() / zb.abs().max())
    assert max(e_a) < 5e-5, e_a
    ga = torch.randn(ya.shape, generator=g).cuda()
    (ya * ga).sum().backward()          # yb's gradient is missing
    (za * ga).sum().backward()
    for (k, p), p2 in zip(dcn.named_parameters(), dcn2.parameters()):
        d = float((p.grad - p2.grad).norm() / p2.grad.norm().clamp_min(1e-20))
        # fc.weight has no ReLU between it and the loss: the two call patterns must agree to round-off there (round 5: was 2e-2
        # like the rest).  Everywhere else ONE ReLU element whose pre-activation lies within the two patterns' forward
        # difference (other tile shapes, other summation order) of zero flips its mask: at 8 x 16 pixels per channel that is
        # ~4e-3 of the layer's gradient per flip, up to 1e-2 accumulated below layer 4 -- measured with the float64 oracle itself
        # (weights perturbed by 1e-6: __graft_entry__.smoke has the numbers), not an arithmetic error
        assert d < (1e-4 if k.endswith("fc.weight") else 2e-2), (k, d)


def test_profile_categories_on_hardware(L):
    """Every engine launch of a training step lands in a category with a positive duration; the streaming passes move their
    algorithmic bytes at a plausible HBM rate (0.3 - 8 TB/s at this size)."""
    from dcn_hip import backbone as bb
    dcn, _ = pc.build_dcn("Resnet34_8s", 3, 480, 640)
    x = torch.randn(2, 3, 480, 640, device="cuda")
    plan = bb.get_plan("Resnet34_8s", 64, 2, 480, 640, 3)
    for _ in range(2):
        dcn.fcn(x).sum().backward()
    plan.profile_begin()
    dcn.fcn(x).sum().backward()
    prof = plan.profile_end()
    assert prof["conv_gemm"][1] == 37 + 36 and prof["conv_wgrad"][1] == 37 and prof["bn_finalize"][1] == 72
    for k, (ms, n, work) in prof.items():
        if k == "conv_gemm_hl":
            continue   # (a sub-count of conv_gemm)
        assert n > 0 and ms > 0, k
    for k in ("bn_apply", "bn_bwd_reduce", "bn_bwd_apply"):
        ms, n, b = prof[k]
        rate = b / (ms * 1e-3) / 1e12
        print(k, "%.2f TB/s over %d launches" % (rate, n))
        assert 0.3 < rate < 8.0, (k, rate)


def test_out_of_range_weight_stops_the_training_path(L):
    """Split-fp16 arithmetic: a convolution weight outside the range of its fp16 image (|w| >= 1023 at the fixed weight scale)
    only raises a status bit inside the call (no synchronisation on the hot path) -- the status word is copied to pinned host
    memory behind the call and a LATER forward call raises, so a training loop cannot go on silently."""
    from dcn_hip import backbone
    backbone.set_conv_mode("f16x3")
    try:
        dcn, _ = pc.build_dcn("Resnet34_8s", 3, 64, 96)
        x = torch.randn(1, 3, 64, 96).cuda()
        dcn.train()
        dcn.forward(x)
        with torch.no_grad():
            dcn.fcn.resnet34_8s.get_parameter("layer2.0.conv1.weight")[3, 2, 1, 1] = 2000.0
        dcn.forward(x)                       # sets the bit; returns normally
        torch.cuda.synchronize()
        with pytest.raises(FloatingPointError, match="outside the range of its fp16 image"):
            for _ in range(3):               # (one status word is watched at a time: the raise comes one or two calls later)
                dcn.forward(x)
                torch.cuda.synchronize()
    finally:
        backbone.set_conv_mode(None)


def test_weight_images_made_on_the_side_stream_full_size(L, dcn_env):
    """Config-2 shapes, forward_pair, NaN-poisoned arenas: with the weight images of forward AND backward made on the side stream
    during the stem (DCN_WSPLIT_OVERLAP, default) descriptors and all gradients equal, bit for bit, those of the run that makes
    them on the caller's stream in front of each pass -- three times in a row (a missing cross-stream dependency would show up
    as a layer that read its weight image before it was written)."""
    import copy
    from dcn_hip import backbone
    from pytorch_segmentation_detection.models import resnet_dilated as prod
    backbone.set_conv_mode("f16x3")
    try:
        torch.manual_seed(3)
        m = prod.Resnet34_8s(num_classes=3).cuda()
        m2 = copy.deepcopy(m)
        g = torch.Generator().manual_seed(5)
        xa, xb = torch.randn(4, 3, 480, 640, generator=g).cuda(), torch.randn(4, 3, 480, 640, generator=g).cuda()
        gy = torch.randn(4, 3, 480, 640, generator=g).cuda()
        backbone.POISON_ARENAS = True
        res = []
        for net, on in ((m, 1), (m2, 0)):
            dcn_env(DCN_WSPLIT_OVERLAP=on)
            net.train()
            for rep in range(3):
                net.zero_grad()
                ya, yb = net.forward_pair(xa, xb)
                ((ya * gy).sum() + (yb * gy).sum()).backward()
            torch.cuda.synchronize()
            res.append((ya.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    finally:
        backbone.POISON_ARENAS = False
        backbone.set_conv_mode(None)
        backbone._PLANS.clear()
        torch.cuda.empty_cache()
    assert bool(torch.isfinite(res[0][0]).all()) and torch.equal(res[0][0], res[1][0])
    for (k, _), g1, g2 in zip(m.named_parameters(), res[0][1], res[1][1]):
        assert bool(torch.isfinite(g1).all()) and torch.equal(g1, g2), k
