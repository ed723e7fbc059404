"""-m gpu tests of the second half of round 2: stream-K tiles completed inside the GEMM launch (no fix-up kernel), and the
batch-norm backward reduction in the epilogue of the dgrad that produces its upstream gradient."""
import os

import pytest
import torch

from helpers import rel_err, use_gfx950_library
import kernel_checks
import parity_common as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = use_gfx950_library()
    assert torch.cuda.is_available()
    return lib


@pytest.mark.parametrize("shape", [
    # n, h, w, cin, cout, k, dil, DCN_GEMM_SK
    (8, 60, 80, 256, 256, 3, 2, "1"),     # layer 3 of config 2: 300 tiles of 256 x 128 on 256 CUs -> 256 data-parallel + 44 stream-K
    (8, 60, 80, 512, 512, 3, 4, "1"),     # layer 4: 600 tiles, 88 stream-K'd
    (4, 60, 80, 128, 256, 1, 1, "100"),   # 1x1: 4 K stages per tile, forced over 100 workgroups
])
def test_stream_k_inline_completion_at_layer_shapes(L, shape, dcn_env):
    """The last contributor of a stream-K tile sums the parked partials and runs the epilogue inside the GEMM launch:
    bit-identical to the separate fix-up kernel, on a garbage-filled workspace, 8 launches in a row (race screen: the
    arrival words, the device-wide release / acquire around them and the self-reset are what is being exercised)."""
    n, h, w, cin, cout, k, dil, sk = shape
    kernel_checks.check_stream_k_inline(L, "cuda", dcn_env, n, h, w, cin, cout, k, dil, sk=sk, repeats=8)


def test_stream_k_inline_forced_small_splits(L, dcn_env):
    """Forced stream-K over few workgroups: every tile has many contributors and every workgroup two segments."""
    kernel_checks.check_stream_k_inline(L, "cuda", dcn_env, 2, 60, 80, 128, 128, 3, 1, sk="37", repeats=8)
    kernel_checks.check_stream_k_inline(L, "cuda", dcn_env, 2, 30, 40, 64, 64, 3, 1, sk="11", tile_m="64", repeats=8)


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, dil, groups, add, relu, sk
    (2, 60, 80, 128, 128, 3, 1, 1, True, True, "0"),
    (8, 60, 80, 256, 256, 3, 2, 2, True, True, "1"),      # grouped (forward_pair), stream-K'd tiles carry the reduction too
    (2, 60, 80, 64, 256, 1, 1, 1, False, True, "0"),      # bottleneck conv3 -> bn2
    (2, 120, 160, 64, 64, 3, 1, 1, True, True, "0"),      # layer 1: 64-wide tiles
    (1, 60, 80, 512, 4, 1, 1, 1, False, True, "0"),       # scoring layer (D = 3 padded to 4) -> last block's batch norm
])
def test_dgrad_with_fused_bn_backward_reduction(L, case, dcn_env):
    n, h, w, cin, cout, k, dil, groups, add, relu, sk = case
    dcn_env(DCN_GEMM_SK=sk)
    r = kernel_checks.check_dgrad_with_bn_backward(L, "cuda", n, h, w, cin, cout, k, dil, groups, add, relu)
    print("dgrad + BN backward reduction", case, r)


@pytest.mark.parametrize("arch,H,W,D", [("Resnet34_8s", 256, 320, 3), ("Resnet50_8s", 128, 160, 8)])
def test_network_step_fused_vs_separate_bn_reduction(L, arch, H, W, D, dcn_env):
    """A training step of the real network with the fused reduction (default) and with the separate reduce pass
    (DCN_BN_BWD_FUSED=0): every batch norm but the stem's and the downsample branches' is fused; gradients agree to
    summation-order round-off, measured against the float64 oracle both are as accurate as the float32 oracle."""
    import copy
    from dcn_hip import backbone
    backbone.set_conv_mode("f16x3")
    try:
        g = torch.Generator().manual_seed(3)
        xa = torch.randn(2, 3, H, W, generator=g)
        xb = torch.randn(2, 3, H, W, generator=g)
        gy = torch.randn(2, D, H, W, generator=g)
        grads = {}
        for fused in (0, 1):
            dcn_env(DCN_BN_BWD_FUSED=fused)
            dcn, o = pc.build_dcn(arch, D, H, W)
            ya, yb = dcn.forward_pair(xa.cuda(), xb.cuda())
            ((ya * gy.cuda()).sum() + (yb * gy.cuda()).sum()).backward()
            torch.cuda.synchronize()
            plan = dcn.fcn._last_plan
            assert plan.groups == 2   # (forward_pair as ONE grouped launch sequence: statistics per group in the fused sums)
            n_bn = len(plan.bn_names)
            n_down = sum(1 for k in plan.bn_names if "downsample" in k)
            assert plan.fused_bn_backward() == (n_bn - n_down - 1 if fused else 0)
            grads[fused] = {k: p.grad.detach().cpu() for k, p in dcn.fcn.named_parameters()}
        o64 = copy.deepcopy(o).double()
        o.train(); o64.train()
        (o(xa) * gy).sum().backward(); (o(xb) * gy).sum().backward()
        (o64(xa.double()) * gy.double()).sum().backward(); (o64(xb.double()) * gy.double()).sum().backward()
        for fused in (0, 1):
            class P:   # (grad_error_stats wants objects with .grad)
                def __init__(s, g): s.grad = g
            st = pc.grad_error_stats([(k, P(v)) for k, v in grads[fused].items()], o.parameters(), o64.parameters())
            print("%s fused=%d: gradient error vs float64 r.m.s. %.2e (float32 oracle %.2e), worst %.2e (%.2e)" %
                  (arch, fused, st["rms_gpu"], st["rms_o32"], st["max_gpu"], st["max_o32"]))
            pc.assert_as_accurate_as_float32(st, factor=2.0, floor=5e-4)
    finally:
        backbone.set_conv_mode(None)


@pytest.mark.parametrize("shape", [(8, 480, 640), (2, 96, 128), (1, 33, 47)], ids=str)
def test_stem_through_uniform_tap_path(L, shape):
    """The 7x7 / 2 stem as a uniform-tap convolution over filter rows (dcn_conv_stem_forward_f16) == the generic gather path ==
    F.conv2d on the CPU, at the full 640x480 image too."""
    kernel_checks.check_stem_uniform_tap(L, "cuda", *shape)
