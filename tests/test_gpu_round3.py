"""-m gpu tests of round 3: the pre-split (hl32) LDS-DMA gather-GEMM of the wide layers (conv_hl_kernels.hip) -- forward and
dgrad against F.conv2d / autograd in float64 and against the fp32-operand split-fp16 kernel, at small ragged shapes and at
the layer shapes of the BASELINE configs, plain launches and stream-K, repeated on one workspace (race screen)."""
import pytest
import torch

from helpers import use_gfx950_library
import kernel_checks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = use_gfx950_library()
    assert torch.cuda.is_available()
    return lib


HL_SMALL = [
    # n, h, w, cin, cout, k, dil, DCN_GEMM_SK, x scale   (the CPU suite's cases: tests/test_emu_kernels.py)
    (1, 12, 20, 32, 40, 3, 2, None, 1.0),
    (2, 16, 24, 64, 288, 3, 1, None, 1.0),
    (2, 16, 24, 64, 288, 3, 1, "5", 1e4),
    (1, 20, 20, 128, 256, 1, 1, "5", 1.0),
    (1, 9, 30, 96, 256, 3, 4, "3", 1e-6),
    (1, 10, 30, 32, 64, 1, 1, None, 1.0),       # one K stage
    (1, 10, 30, 64, 64, 1, 1, None, 1.0),       # two K stages
]
HL_LAYERS = [
    (8, 60, 80, 256, 256, 3, 2, None, 1.0),     # layer 3 of config 2: 150 tiles on 256 CUs, less than a round: data-parallel
    (8, 60, 80, 256, 256, 3, 2, "170", 1.0),    # ... and forced over 170 stream-K workgroups (200 tiles of 192 rows: 170 + 30)
    (8, 60, 80, 512, 512, 3, 4, "1", 1.0),      # layer 4: 300 tiles -> 256 data-parallel + 44 stream-K
    (8, 60, 80, 128, 256, 3, 1, None, 1.0),     # layer3.0.conv1 forward (dgrad: 128 destination channels, one ragged N tile)
    (2, 60, 80, 256, 512, 3, 4, "0", 1.0),      # config 1 size, no stream-K: 38 tiles, ragged last M tile (9600 rows)
    (3, 120, 160, 256, 1024, 1, 1, None, 1.0),  # ResNet50-8s 1x1 expansion at 1280 x 960 (config 5): 225 x 4 tiles, 8 K stages
]


@pytest.mark.parametrize("rows", ["256", "192", "320"])   # (tile height: three software pipelines, conv_hl_kernels.hip)
@pytest.mark.parametrize("case", HL_SMALL + HL_LAYERS, ids=[str(c) for c in HL_SMALL + HL_LAYERS])
def test_conv_hl32_lds_dma_gather_gemm(L, case, rows, dcn_env):
    n, h, w, cin, cout, k, dil, sk, sx = case
    if rows == "320" and sk not in (None, "0"):
        pytest.skip("320-row tiles run data-parallel launches only (hl_shape)")
    for rep in range(2):   # (again on the workspace the previous launch left behind; another seed, other data)
        res = kernel_checks.check_conv_hl(L, "cuda", n, h, w, cin, cout, k, dil, set_env=dcn_env, sk=sk, scale_x=sx,
                                          seed=len(str(case)) + rep, rows=rows)
    print(case, rows, res)


def test_conv_hl32_repeated_launches_are_bit_identical(L, dcn_env):
    """Race screen of the LDS-DMA schedule: 20 launches of the layer-4 convolution on the same operands must agree bit for
    bit (a fragment read that is not covered by counted wait + barrier shows up as a rare wrong tile)."""
    import ctypes
    lib = L.get()
    n, h, w, cin, cout, k, dil = 8, 60, 80, 512, 512, 3, 4
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt_ = (torch.randn(cout, k, k, cin, generator=g) * 0.1).cuda()
    M, K = n * h * w, k * k * cin
    ax = x.abs().max().reshape(1)
    xi = kernel_checks.hl32_image(L, lib, x.reshape(M, cin), ax, "cuda")
    w_hl = torch.empty(cout * K, device="cuda")
    P, I = ctypes.c_void_p, ctypes.c_int
    arr = lambda ty, v: (ty * 1)(v)
    assert lib.dcn_split_weights_hl32(1, arr(P, wt_.data_ptr()), arr(P, w_hl.data_ptr()), arr(I, cout), arr(I, k * k), arr(I, cin),
                                      arr(I, cout), 0, 64.0, None) == 0
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, dil, dil, cout, 0)
    for sk, rows in (("1", "256"), ("0", "256"), ("200", "256"), ("1", "192"), ("0", "192"), ("200", "192")):
        dcn_env(DCN_GEMM_SK=sk, DCN_GEMM_HL_ROWS=rows)
        ws = kernel_checks.garbage(max(lib.dcn_conv_gemm_workspace_hl(ctypes.byref(d), 0), 8), "cuda", 1)
        outs = []
        for _ in range(20):
            out = torch.full((n, h, w, cout), float("nan"), device="cuda")
            assert lib.dcn_conv_forward_hl(ctypes.byref(d), L.ptr(xi), L.ptr(ax), L.ptr(w_hl), 64.0, None, L.ptr(out), None,
                                           L.ptr(ws), L.stream_ptr()) == 0
            outs.append(out)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(outs[0]).all())
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), "sk=%s rows=%s" % (sk, rows)


WGRAD_HL = [
    # n, h, w, cin, cout, k, dil, forced splits   (small: the CPU suite's cases; then the layer shapes of configs 2 and 5)
    (1, 3, 40, 32, 64, 3, 1, None),
    (2, 5, 36, 64, 288, 3, 2, "3"),
    (1, 8, 32, 256, 256, 1, 1, "2"),
    (1, 4, 48, 96, 32, 3, 4, None),
    (8, 60, 80, 256, 256, 3, 2, None),      # layer 3 of config 2: 9 tiles x 28 pixel splits
    (8, 60, 80, 512, 512, 3, 4, None),      # layer 4: 36 tiles x 7 splits
    (8, 60, 80, 128, 256, 3, 1, None),      # layer3.0.conv1: K = 1152 = 4.5 tiles (ragged, taps change inside a tile)
    (2, 120, 160, 512, 512, 3, 4, None),    # ResNet50-8s layer-4 3x3 at 1280 x 960
]


@pytest.mark.parametrize("case", WGRAD_HL, ids=[str(c) for c in WGRAD_HL])
def test_wgrad_hl32_transposing_lds_reads(L, case, dcn_env):
    n, h, w, cin, cout, k, dil, splits = case
    for rep in range(2):
        res = kernel_checks.check_wgrad_hl(L, "cuda", n, h, w, cin, cout, k, dil, set_env=dcn_env, splits=splits,
                                           seed=len(str(case)) + rep)
    print(case, res)


@pytest.mark.parametrize("env", [
    dict(DCN_GEMM_HL=0, DCN_WGRAD_HL=0),                          # round-2 kernels everywhere
    pytest.param(dict(DCN_GEMM_HL=1, DCN_WGRAD_HL=1, DCN_HL_PRODUCERS=0), marks=pytest.mark.slow),   # hl32 kernels, operand images by stand-alone split passes
    dict(DCN_GEMM_HL=2, DCN_WGRAD_HL=2),                          # every supported convolution (narrow layers included)
    pytest.param(dict(DCN_GEMM_HL=2, DCN_GEMM_HL_ROWS=192), marks=pytest.mark.slow),   # ... all of them on 192-row tiles
    pytest.param(dict(DCN_GEMM_HL_ROWS=-320), marks=pytest.mark.slow),                 # the round-3 tile choice (never 320 rows)
], ids=["hl-off", "split-passes", "forced-everywhere", "forced-192", "no-320"])
def test_headline_step_vs_fixture_under_hl32_switches(L, env, dcn_env):
    """The headline workload (config 2, forward_pair) against its float32 / float64 oracle fixture with the hl32 path switched
    off, fed by stand-alone split passes, and forced onto every supported layer (also with 192-row tiles only): same tolerances as the default
    (tests/test_gpu_configs.py)."""
    import parity_common as pc
    from dcn_hip import backbone
    from test_gpu_configs import _check
    backbone.set_conv_mode("f16x3")
    try:
        dcn_env(**env)
        backbone._PLANS.clear()        # (plans reserve the saved hl32 images when they are built)
        r = pc.run_config_against_fixture(2, pair_call=True)
        _check(r, 2)
    finally:
        backbone._PLANS.clear()
        backbone.set_conv_mode(None)
        torch.cuda.empty_cache()


def test_activations_that_are_never_stored_full_size(L, dcn_env):
    """Config-2 shapes (forward_pair of 4 + 4 images, 640 x 480): inside the blocks of layers 3-4 the batch-norm apply pass
    writes the hl32 image only (both readers take it), and the stem's batch norm + ReLU is applied inside the max-pool pass
    (its activation is never stored).  Saved arena and workspace poisoned with NaN bytes before every call: descriptors,
    running statistics and all gradients bit-identical to the run that writes those tensors."""
    import copy
    from dcn_hip import backbone
    from pytorch_segmentation_detection.models import resnet_dilated as prod
    backbone.set_conv_mode("f16x3")
    torch.manual_seed(3)
    m = prod.Resnet34_8s(num_classes=3).cuda().train()
    m2 = copy.deepcopy(m)
    g = torch.Generator().manual_seed(4)
    xa = torch.randn(4, 3, 480, 640, generator=g).cuda()
    xb = torch.randn(4, 3, 480, 640, generator=g).cuda()
    gy = torch.randn(8, 3, 480, 640, generator=g).cuda()

    def run(net):
        y = torch.cat(net.forward_pair(xa, xb))
        (y * gy).sum().backward()
        torch.cuda.synchronize()
        return y.detach()
    backbone.POISON_ARENAS = True
    try:
        dcn_env(DCN_HL_ONLY_MID=1, DCN_STEM_POOL_FUSED=1)
        backbone._PLANS.clear()
        ya = run(m)
        dcn_env(DCN_HL_ONLY_MID=0, DCN_STEM_POOL_FUSED=0)
        yb = run(m2)
    finally:
        backbone.POISON_ARENAS = False
        backbone._PLANS.clear()
        backbone.set_conv_mode(None)
        torch.cuda.empty_cache()
    assert bool(torch.isfinite(ya).all()) and torch.equal(ya, yb)
    for (k, p1), p2 in zip(m.named_parameters(), m2.parameters()):
        assert bool(torch.isfinite(p1.grad).all()) and torch.equal(p1.grad, p2.grad), k
    for (k, b1), b2 in zip(m.named_buffers(), m2.buffers()):
        assert torch.equal(b1, b2), k
