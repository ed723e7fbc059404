"""-m gpu tests of round 5: the small-tile hl32 gather-GEMM (conv_hlx_kernels.hip: 160 x 256 and 160 x 128 tiles of
16 x 16 x 32 MFMAs, K groups inside the workgroup, K split over workgroups) -- the launches of the reference's own batch size
(training.yaml:14 `batch_size: 1`) and of its two separate forward calls (training.py:329-333) -- against F.conv2d / autograd
in float64 and the fp32-operand kernel, at the CPU suite's ragged shapes and at the layer shapes of BASELINE configs 1 and 2;
repeated launches bit for bit (race screen of the LDS-DMA loop and of the in-launch completion of the K splits)."""
import ctypes

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


HLX_SMALL = [
    # n, h, w, cin, cout, k, dil, "kg,splits" forced, x scale   (the CPU suite's cases: tests/test_emu_kernels.py)
    (1, 12, 20, 32, 40, 3, 2, "1,1", 1.0),
    (2, 16, 24, 64, 288, 3, 1, "1,1", 1.0),
    (2, 16, 24, 64, 288, 3, 1, "1,3", 1e4),
    (2, 16, 24, 64, 288, 3, 1, "2,1", 1.0),
    (1, 20, 20, 128, 256, 1, 1, "2,2", 1.0),
    (1, 9, 30, 64, 256, 3, 4, "2,3", 1e-6),
    (1, 10, 30, 32, 64, 1, 1, "1,1", 1.0),
    (1, 10, 30, 64, 64, 1, 1, "2,1", 1.0),
]
HLX_LAYERS = [
    (8, 60, 80, 256, 256, 3, 2, "1,1", 1.0),     # layer 3 of config 2: 240 tiles of 160 x 256 -- one round
    (4, 60, 80, 256, 256, 3, 2, "2,1", 1.0),     # ... of one forward call of the two-call pattern: 240 tiles of 160 x 128
    (2, 60, 80, 512, 512, 3, 4, "2,1", 1.0),     # layer 4 of config 1 (one pair): 240 tiles of 160 x 128
    (2, 60, 80, 256, 256, 3, 2, "2,2", 1.0),     # layer 3 of config 1: 120 tiles x 2 K splits
    (1, 60, 80, 512, 512, 3, 4, "2,2", 1.0),     # layer 4, ONE image (two-call pattern at B = 1): 120 tiles x 2
    (1, 60, 80, 256, 256, 3, 2, "2,4", 1.0),     # layer 3, one image: 60 tiles x 4
    (1, 60, 80, 256, 512, 3, 4, "1,3", 1.0),     # layer4.0 on the wide tile, 3 splits
    (8, 60, 80, 128, 128, 3, 1, "2,1", 1.0),     # layer 2 (128 channels): 240 tiles of 160 x 128
]


@pytest.mark.parametrize("case", HLX_SMALL + HLX_LAYERS, ids=[str(c) for c in HLX_SMALL + HLX_LAYERS])
def test_conv_hlx_small_tiles(L, case, dcn_env):
    n, h, w, cin, cout, k, dil, hlx, sx = case
    for rep in range(2):   # (again on the workspace the previous launch left behind; another seed, other data)
        res = kernel_checks.check_conv_hl(L, "cuda", n, h, w, cin, cout, k, dil, set_env=dcn_env, scale_x=sx,
                                          seed=len(str(case)) + rep, hlx=hlx)
    print(case, res)


def test_conv_hlx_repeated_launches_are_bit_identical(L, dcn_env):
    """20 launches of the layer-4 convolution at two images and of the layer-3 one at one image on the same operands, every
    (K groups, K splits) shape: bit for bit the same tensor (a fragment read not covered by wait + barrier, or a partial summed
    before it is visible, shows up as a rare wrong tile)."""
    lib = L.get()
    for (n, cin, cout, dil), shapes in (((2, 512, 512, 4), ("1,1", "2,1", "2,2", "1,3")), ((1, 256, 256, 2), ("2,4", "1,4", "2,1"))):
        h, w, k = 60, 80, 3
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
        first = None
        for hlx in shapes:
            dcn_env(DCN_GEMM_HLX=hlx)
            info = (ctypes.c_int * 6)()
            assert lib.dcn_conv_hl_shape_info(ctypes.byref(d), 0, info) == 0 and info[0] == 160
            assert "%d,%d" % (info[2], info[3]) == hlx
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
                assert torch.equal(o, outs[0]), "hlx=%s" % hlx
            if first is None:
                first = outs[0]
            else:   # other tile shapes / split counts: the same products in another summation order
                assert float((outs[0] - first).abs().max() / first.abs().max()) < 5e-6


def test_tile_choice_of_the_small_batch_launches(L, dcn_env):
    """What hl_shape picks by default for the launches this round is about -- one round of 240 workgroups where the big tiles
    leave most of the chip idle -- and that the headline shapes keep their big tiles."""
    lib = L.get()
    dcn_env(DCN_GEMM_HL=1)   # (reload: the defaults)
    for n, cin, cout, dil, want in ((4, 256, 256, 2, (160, 128, 2, 1)),     # one call of the two-call pattern at B = 4, layer 3
                                    (2, 512, 512, 4, (160, 128, 2, 1)),     # config 1 as a pair, layer 4
                                    (1, 512, 512, 4, (160, 128, 2, 2)),     # config 1, two calls, layer 4: 120 tiles x 2 K splits
                                    (8, 512, 512, 4, (320, 256, 1, 1))):    # the headline: unchanged
        d = L.ConvDesc(n, 60, 80, cin, 60, 80, cout, 3, 3, 1, dil, dil, cout, 0)
        info = (ctypes.c_int * 6)()
        assert lib.dcn_conv_hl_shape_info(ctypes.byref(d), 0, info) == 0
        assert tuple(info[:4]) == want and info[3] * info[4] * info[5] == 240, (n, cin, list(info))
        assert lib.dcn_conv_hl_eligible(ctypes.byref(d), 0) == 1
    d = L.ConvDesc(8, 60, 80, 256, 60, 80, 256, 3, 3, 1, 2, 2, 256, 0)       # layer 3 at 8 images: 192-row tiles (200) or 160 x 256 (240)
    info = (ctypes.c_int * 6)()
    assert lib.dcn_conv_hl_shape_info(ctypes.byref(d), 0, info) == 0 and info[0] in (160, 192) and info[4] * info[5] >= 200


WGRAD_HLR = [
    # n, h, w, cin, cout, k, dil, forced splits, DCN_WGRAD_HLR
    (1, 3, 40, 64, 64, 3, 1, None, 2),        # the CPU suite's small cases
    (2, 4, 33, 128, 128, 3, 1, "3", 2),
    (1, 2, 20, 64, 192, 3, 1, "1", 2),
    (8, 120, 160, 64, 64, 3, 1, None, 1),     # layer 1 of config 2: one 64-channel tile, 256 stage ranges
    (8, 60, 80, 128, 128, 3, 1, None, 1),     # layer 2: rows of 80 pixels = 2.5 stages (the third half empty), two tiles x 128 ranges
    (2, 120, 160, 64, 64, 3, 1, None, 2),     # config 1 (B = 1): not by default (the step does not gain), forced
    (4, 120, 160, 128, 128, 3, 1, None, 1),   # ResNet50-8s layer-2 3 x 3 at 1280 x 960 (config 5)
    (1, 5, 64, 128, 64, 3, 1, "5", 2),        # odd height: the last row pair's second row is empty
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", WGRAD_HLR, ids=[str(c) for c in WGRAD_HLR])
@pytest.mark.parametrize("pairs", [1, 0])
def test_wgrad_hl32_row_window_kernel(L, case, pairs, dcn_env):
    """conv_wgrad_hlrp_kernel (row pairs, the default) / conv_wgrad_hlr_kernel (wgrad_hl_kernels.hip) on the hardware: against float64
    autograd, the fp32-operand kernel, and itself (bit-reproducible) -- kernel_checks.check_wgrad_hl; the layer shapes take it by
    default."""
    n, h, w, cin, cout, k, dil, splits, hlr = case
    if pairs == 0 and n * h * w > 200000:
        pytest.skip("single-row form: the small cases and one layer shape")
    dcn_env(DCN_WGRAD_HLR=hlr, DCN_WGRAD_HLR_PAIRS=pairs)
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, dil * (k - 1) // 2, dil, cout, 0)
    assert L.get().dcn_conv_wgrad_hl_kind(ctypes.byref(d)) == 2
    assert L.get().dcn_conv_wgrad_hl_eligible(ctypes.byref(d)) == 1
    for rep in range(2):
        res = kernel_checks.check_wgrad_hl(L, "cuda", n, h, w, cin, cout, k, dil, set_env=dcn_env, splits=splits,
                                           seed=len(str(case)) + rep)
    print(case, res)
