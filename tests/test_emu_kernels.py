"""Individual gfx950 kernels through the C ABI (kernels compiled for the host, tests/hostemu) against plain
PyTorch fp32 on CPU: fp32-MFMA implicit-GEMM convolution (forward / dgrad / wgrad incl. the fused BN partial
sums), bilinear upsample and its backward.  The MFMA fragment maps, LDS images and index arithmetic executed
here are the ones that run on the GPU."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, use_emulation_library


@pytest.fixture(scope="module")
def L():
    return use_emulation_library()


CONV_CASES = [
    # n, hin, win, cin, cout, k, stride, pad, dil
    (1, 8, 10, 8, 16, 3, 1, 1, 1),
    (2, 9, 7, 4, 12, 7, 2, 3, 1),       # stem-like: cin padded to 4, K = 196 (ragged K tile)
    (1, 12, 10, 16, 24, 3, 2, 1, 1),    # layer2.0.conv1-like stride 2 (transposed-gather dgrad)
    (1, 12, 10, 16, 24, 1, 2, 0, 1),    # 1x1 / 2 downsample
    (1, 10, 12, 20, 136, 3, 1, 2, 2),   # dilation 2, two N tiles, ragged channels
    (2, 9, 9, 32, 8, 3, 1, 4, 4),       # dilation 4 with halo larger than the image border
    (3, 7, 7, 8, 8, 3, 1, 1, 1),        # M = 147 rows: two M tiles, second one ragged
]


@pytest.mark.parametrize("tile_m,sk", [("32", "0"), ("64", "0"), ("128", "0"), ("32", "3"), ("64", "5"), ("128", "2"),
                                       ("64", "7")])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_forward_dgrad_wgrad(L, case, tile_m, sk, dcn_env):
    # every workgroup-tile height of the gather-GEMM kernel; SK 0: one workgroup per tile; N: stream-K over N workgroups
    dcn_env(DCN_GEMM_TILE_M=tile_m, DCN_GEMM_SK=sk)
    lib = L.get()
    n, hin, win, cin, cout, k, stride, pad, dil = case
    hout = (hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wout = (win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    d = L.ConvDesc(n, hin, win, cin, hout, wout, cout, k, k, stride, pad, dil, cout)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, cin, hin, win, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).requires_grad_(True)
    x_nhwc = x.detach().permute(0, 2, 3, 1).contiguous()
    w_k = w.detach().permute(0, 2, 3, 1).contiguous()
    out = torch.full((n, hout, wout, cout), float("nan"))
    mt = lib.dcn_conv_num_mtiles(ctypes.byref(d))
    assert mt in [(n * hout * wout + b - 1) // b for b in (32, 64, 128)]
    part = torch.full((mt, 3, cout), float("nan"))
    ws_f = torch.empty(max(lib.dcn_conv_gemm_workspace(ctypes.byref(d), 0), 4) // 4)
    ws_d = torch.empty(max(lib.dcn_conv_gemm_workspace(ctypes.byref(d), 1), 4) // 4)
    if sk != "0" and k > 1:
        assert ws_f.numel() > 1 and ws_d.numel() > 1   # stream-K really is exercised (K spans >= 2 K tiles)
    assert lib.dcn_conv_forward(ctypes.byref(d), L.ptr(x_nhwc), L.ptr(w_k), None, L.ptr(out), L.ptr(part), L.ptr(ws_f), None) == 0
    ref = F.conv2d(x, w, None, stride, pad, dil)
    refn = ref.detach().permute(0, 2, 3, 1)
    assert rel_err(out, refn) < 2e-6
    assert rel_err(part.sum(0)[0], refn.sum((0, 1, 2))) < 5e-6
    assert rel_err(part.sum(0)[1], (refn ** 2).sum((0, 1, 2))) < 5e-6
    assert rel_err(part.max(0).values[2], refn.abs().amax((0, 1, 2))) < 2e-6   # third row: max |x| per channel (BN output bound)
    dout = torch.randn(n, hout, wout, cout, generator=g)
    ref.backward(dout.permute(0, 3, 1, 2))
    wt = torch.empty(cin, k * k, cout)
    assert lib.dcn_transpose_weight(L.ptr(w_k), L.ptr(wt), cout, k * k, cin, cout, None) == 0
    assert torch.equal(wt, w_k.reshape(cout, k * k, cin).permute(2, 1, 0))
    add = torch.randn(n, hin, win, cin, generator=g)
    din = torch.full((n, hin, win, cin), float("nan"))
    assert lib.dcn_conv_dgrad(ctypes.byref(d), L.ptr(dout), L.ptr(wt), L.ptr(add), L.ptr(din), L.ptr(ws_d), None) == 0
    assert rel_err(din, x.grad.permute(0, 2, 3, 1) + add) < 3e-6
    dw = torch.full((cout, k, k, cin), float("nan"))
    slab = torch.empty(max(lib.dcn_conv_wgrad_workspace(ctypes.byref(d)), 4) // 4)
    assert lib.dcn_conv_wgrad(ctypes.byref(d), L.ptr(x_nhwc), L.ptr(dout), L.ptr(dw), L.ptr(slab), None) == 0
    assert rel_err(dw, w.grad.permute(0, 2, 3, 1)) < 3e-6


def test_conv_scoring_layer_shape(L):
    """fc: 1x1 conv to D=3 channels with bias, output rows padded to 4 floats (what the upsample kernel reads)."""
    lib = L.get()
    n, h, w_, cin, D, ld = 1, 13, 17, 64, 3, 4
    d = L.ConvDesc(n, h, w_, cin, h, w_, D, 1, 1, 1, 0, 1, ld)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w_, generator=g)
    w = torch.randn(D, cin, 1, 1, generator=g) * 0.1
    b = torch.randn(D, generator=g)
    out = torch.zeros(n, h, w_, ld)
    assert lib.dcn_conv_forward(ctypes.byref(d), L.ptr(x.permute(0, 2, 3, 1).contiguous()),
                                L.ptr(w.permute(0, 2, 3, 1).contiguous()), L.ptr(b), L.ptr(out), None, None, None) == 0
    assert rel_err(out[..., :D], F.conv2d(x, w, b).permute(0, 2, 3, 1)) < 2e-6
    assert float(out[..., D:].abs().max()) == 0.0


def test_conv_rejects_bad_arguments(L):
    lib = L.get()
    d = L.ConvDesc(1, 8, 8, 6, 8, 8, 8, 3, 3, 1, 1, 1, 8)   # cin not a multiple of 4
    t = torch.zeros(4096)
    assert lib.dcn_conv_forward(ctypes.byref(d), L.ptr(t), L.ptr(t), None, L.ptr(t), None, None, None) == -1
    d = L.ConvDesc(1, 8, 8, 8, 8, 8, 8, 3, 3, 1, 1, 1, 8)
    assert lib.dcn_conv_forward(ctypes.byref(d), None, L.ptr(t), None, L.ptr(t), None, None, None) == -1


@pytest.mark.parametrize("shape", [(2, 4, 5, 3, 32, 40), (1, 3, 3, 16, 24, 24), (1, 6, 8, 5, 41, 59)])
def test_upsample_forward_backward(L, shape):
    lib = L.get()
    n, hl, wl, D, H, W = shape
    ld = (D + 3) // 4 * 4
    g = torch.Generator().manual_seed(2)
    low = torch.randn(n, D, hl, wl, generator=g, requires_grad=True)
    lowp = torch.zeros(n, hl, wl, ld)
    lowp[..., :D] = low.detach().permute(0, 2, 3, 1)
    out = torch.empty(n, H, W, D)
    assert lib.dcn_upsample_forward(L.ptr(lowp), n, hl, wl, ld, D, H, W, 0, L.ptr(out), None) == 0
    ref = F.interpolate(low, size=(H, W), mode="bilinear", align_corners=True)
    assert rel_err(out, ref.detach().permute(0, 2, 3, 1)) < 2e-6
    gout = torch.randn(n, H, W, D, generator=g)
    ref.backward(gout.permute(0, 3, 1, 2))
    glow = torch.full((n, hl, wl, ld), float("nan"))
    tmp = torch.empty(lib.dcn_upsample_backward_tmp_bytes(n, hl, W, D) // 4)
    assert lib.dcn_upsample_backward(L.ptr(gout), n, hl, wl, ld, D, H, W, L.ptr(glow), L.ptr(tmp), None) == 0
    assert rel_err(glow[..., :D], low.grad.permute(0, 2, 3, 1)) < 3e-6
    assert float(glow[..., D:].abs().sum()) == 0.0
    # K10: fused per-pixel L2 normalisation (network.py:256-259)
    outn = torch.empty(n, H, W, D)
    assert lib.dcn_upsample_forward(L.ptr(lowp), n, hl, wl, ld, D, H, W, 1, L.ptr(outn), None) == 0
    r = ref.detach()
    assert rel_err(outn, (r / r.norm(2, 1, keepdim=True)).permute(0, 2, 3, 1)) < 3e-6


@pytest.mark.parametrize("D", [3, 16, 5])
def test_best_match_search_vs_numpy(L, D):
    """network.py:517-523: norm_diffs = sqrt(sum(square(res_b - d), axis=2)); argmin (first occurrence)."""
    import numpy as np
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork as DCN
    H, W, Q = 23, 31, 37       # 713 pixels: three workgroups, the last one ragged; two query tiles
    g = torch.Generator().manual_seed(D)
    res_a = torch.randn(H, W, D, generator=g)
    res_b = torch.randn(H, W, D, generator=g)
    res_b[5, 7] = res_b[20, 3] = res_a[2, 4]          # an exact tie: np.argmin returns the first (row-major) pixel
    pix = torch.stack([torch.randint(0, W, (Q,), generator=g), torch.randint(0, H, (Q,), generator=g)], 1)
    pix[0] = torch.tensor([4, 2])
    uv, dist, nd = DCN.find_best_matches(pix, res_a, res_b, return_norm_diffs=True)
    for i in range(Q):
        ref_uv, ref_diff, ref_nd = DCN.find_best_match((int(pix[i, 0]), int(pix[i, 1])), res_a.numpy(), res_b.numpy())
        assert (int(uv[i, 0]), int(uv[i, 1])) == (int(ref_uv[0]), int(ref_uv[1])), i
        np.testing.assert_allclose(dist[i].item(), ref_diff, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(nd[i].numpy(), ref_nd, rtol=1e-5, atol=1e-6)
    assert (int(uv[0, 0]), int(uv[0, 1])) == (7, 5) and dist[0].item() == 0.0
    # masked search (evaluation.py masks out the background): only candidates with mask != 0
    mask = torch.zeros(H, W, dtype=torch.uint8)
    mask[10:, :] = 1
    uvm, distm, _ = DCN.find_best_matches(pix, res_a, res_b, mask_b=mask)
    ndm = nd.clone()
    ndm[:, :10, :] = float("inf")
    flat = ndm.view(Q, -1).argmin(1)
    assert torch.equal(uvm[:, 0], flat % W) and torch.equal(uvm[:, 1], flat // W)
    empty = torch.zeros(H, W, dtype=torch.uint8)
    uve, diste, _ = DCN.find_best_matches(pix[:2], res_a, res_b, mask_b=empty)
    assert torch.isinf(diste).all()


F16_CASES = [
    (1, 8, 10, 8, 16, 3, 1, 1, 1),
    (2, 9, 7, 4, 12, 7, 2, 3, 1),        # stem-like, K = 196 -> padded to 200 halves per weight row
    (1, 12, 10, 16, 24, 3, 2, 1, 1),     # stride 2 (transposed-gather dgrad)
    (1, 10, 12, 20, 136, 3, 1, 2, 2),    # two N tiles, ragged
    (3, 7, 7, 8, 8, 3, 1, 1, 1),
    (1, 6, 5, 64, 72, 3, 1, 4, 4),       # K = 576: 18 stages
    (2, 24, 20, 8, 4, 1, 1, 0, 1),       # 1x1, narrow Cout, 960 pixels: several wgrad splits
    # channel counts that are multiples of 32: the uniform-tap buffer-load path (forward: cin, dgrad: cout)
    (1, 9, 11, 32, 64, 3, 1, 2, 2),      # dilated, ragged M (99 pixels)
    (2, 10, 9, 64, 96, 3, 2, 1, 1),      # stride 2 (transposed gather with the divisibility test), Cout not a tile multiple
    (1, 12, 10, 64, 128, 1, 2, 0, 1),    # 1x1 stride-2 downsample
    (1, 5, 6, 96, 32, 3, 1, 4, 4),       # dilation 4 on a 5x6 map: most taps fall outside the image
    # output rows of >= 32 pixels, width % 4 == 0: wgrad's carried (image, y, x) position instead of divisions
    (1, 3, 36, 8, 8, 3, 1, 1, 1),
    (1, 6, 72, 4, 12, 3, 2, 1, 1),       # stride 2
    (2, 2, 40, 8, 16, 1, 1, 0, 1),       # row and image wrap inside a split
    # 256 output channels and more: wgrad's 256-channel / 8-wavefront tile
    (1, 4, 36, 8, 256, 3, 1, 1, 1),      # carried-position path, 2 x 72 K columns -> one ragged K tile
    (2, 5, 6, 32, 512, 1, 1, 0, 1),      # two channel tiles, division path, 60 pixels
]


# tile_m 256 = the 8-wavefront 256 x 128 tile (only taken when the destination has more than 64 channels)
@pytest.mark.parametrize("tile_m,sk", [("64", "0"), ("128", "0"), ("64", "3"), ("128", "2"), ("256", "0"), ("256", "3")])
@pytest.mark.parametrize("case", F16_CASES, ids=[str(c) for c in F16_CASES])
def test_conv_f16x3_forward_dgrad(L, case, tile_m, sk, dcn_env):
    """Split-fp16 gather-GEMM (fp16 MFMA, hi/lo operands): must reproduce the fp32 convolution to ~1e-6."""
    dcn_env(DCN_GEMM_TILE_M=tile_m, DCN_GEMM_SK=sk)
    lib = L.get()
    n, hin, win, cin, cout, k, stride, pad, dil = case
    hout = (hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wout = (win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    d = L.ConvDesc(n, hin, win, cin, hout, wout, cout, k, k, stride, pad, dil, cout)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, cin, hin, win, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).requires_grad_(True)
    x_nhwc = x.detach().permute(0, 2, 3, 1).contiguous()
    w_k = w.detach().permute(0, 2, 3, 1).contiguous()
    K = k * k * cin
    kp = lib.dcn_f16_kpad(K)
    wh = torch.empty(cout, kp, dtype=torch.float16)
    wl = torch.empty(cout, kp, dtype=torch.float16)
    assert lib.dcn_split_rows_f16(L.ptr(w_k), L.ptr(wh), L.ptr(wl), cout, K, 64.0, None) == 0
    rec = (wh.float() + wl.float())[:, :K] / 64.0
    assert rel_err(rec, w_k.reshape(cout, K)) < 1e-6 and float(wh[:, K:].abs().max() if kp > K else 0) == 0
    out = torch.full((n, hout, wout, cout), float("nan"))
    mt = lib.dcn_conv_num_mtiles_f16(ctypes.byref(d))
    part = torch.full((mt, 3, cout), float("nan"))
    ws_f = torch.empty(max(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 0), 4) // 4)
    ws_d = torch.empty(max(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 1), 4) // 4)
    assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(x_nhwc), None, L.ptr(wh), L.ptr(wl), 64.0, None, L.ptr(out),
                                    L.ptr(part), L.ptr(ws_f), None) == 0
    ref = F.conv2d(x, w, None, stride, pad, dil)
    refn = ref.detach().permute(0, 2, 3, 1)
    assert rel_err(out, refn) < 3e-6
    # activations far outside fp16's range (2e5 overflows, 1e-7 is below its subnormals): the abs-max driven power-of-two
    # pre-scale of the operand keeps the result at fp32 accuracy
    for big in (2.0e5, 1.0e-7):
        xb = (x_nhwc * big).contiguous()
        xmax = xb.abs().max().reshape(1).clone()
        outb = torch.full((n, hout, wout, cout), float("nan"))
        assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(xb), L.ptr(xmax), L.ptr(wh), L.ptr(wl), 64.0, None,
                                        L.ptr(outb), None, L.ptr(ws_f), None) == 0
        assert rel_err(outb, refn * big) < 3e-6, big
    assert rel_err(part.sum(0)[0], refn.sum((0, 1, 2))) < 1e-5
    assert rel_err(part.sum(0)[1], (refn ** 2).sum((0, 1, 2))) < 1e-5
    # dgrad with a TINY gradient tensor (1e-7 scale): the abs-max driven power-of-two pre-scale keeps it inside fp16
    dout = torch.randn(n, hout, wout, cout, generator=g) * 1e-7
    ref.backward(dout.permute(0, 3, 1, 2))
    wt = torch.empty(cin, k * k, cout)
    assert lib.dcn_transpose_weight(L.ptr(w_k), L.ptr(wt), cout, k * k, cin, cout, None) == 0
    Kt = k * k * cout
    kpt = lib.dcn_f16_kpad(Kt)
    wth = torch.empty(cin, kpt, dtype=torch.float16)
    wtl = torch.empty(cin, kpt, dtype=torch.float16)
    assert lib.dcn_split_rows_f16(L.ptr(wt), L.ptr(wth), L.ptr(wtl), cin, Kt, 64.0, None) == 0
    amax = dout.abs().max().reshape(1).clone()
    add = torch.randn(n, hin, win, cin, generator=g) * 1e-7
    din = torch.full((n, hin, win, cin), float("nan"))
    assert lib.dcn_conv_dgrad_f16(ctypes.byref(d), L.ptr(dout), L.ptr(wth), L.ptr(wtl), 64.0, L.ptr(amax), L.ptr(add),
                                  L.ptr(din), L.ptr(ws_d), None) == 0
    assert rel_err(din, x.grad.permute(0, 2, 3, 1) + add) < 5e-6
    if sk != "0":
        # stream-K tiles are completed inside the launch by their last contributor (arrival words behind the parked
        # partials, tagged with a launch id: no clearing needed -- the workspace is filled with garbage here);
        # DCN_GEMM_SK_FIXUP=kernel keeps the separate fix-up kernel: the two must agree bit for bit
        ws_f.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1)
        ws_d.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1)
        dcn_env(DCN_GEMM_TILE_M=tile_m, DCN_GEMM_SK=sk, DCN_GEMM_SK_FIXUP="kernel")
        out_k = torch.full_like(out, float("nan"))
        part_k = torch.full_like(part, float("nan"))
        din_k = torch.full_like(din, float("nan"))
        assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(x_nhwc), None, L.ptr(wh), L.ptr(wl), 64.0, None, L.ptr(out_k),
                                        L.ptr(part_k), L.ptr(ws_f), None) == 0
        assert lib.dcn_conv_dgrad_f16(ctypes.byref(d), L.ptr(dout), L.ptr(wth), L.ptr(wtl), 64.0, L.ptr(amax), L.ptr(add),
                                      L.ptr(din_k), L.ptr(ws_d), None) == 0
        assert torch.equal(out_k, out) and torch.equal(part_k, part) and torch.equal(din_k, din)
        dcn_env(DCN_GEMM_TILE_M=tile_m, DCN_GEMM_SK=sk, DCN_GEMM_SK_FIXUP="inline")
        for _ in range(2):   # (and again on the workspace the previous launches left behind)
            out_i = torch.full_like(out, float("nan"))
            assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(x_nhwc), None, L.ptr(wh), L.ptr(wl), 64.0, None,
                                            L.ptr(out_i), None, L.ptr(ws_f), None) == 0
            assert torch.equal(out_i, out)
    # wgrad: pixels are the reduction index, dy again 1e-7-scaled; fixed split order -> deterministic
    dw = torch.full((cout, k, k, cin), float("nan"))
    slabs = torch.empty(max(lib.dcn_conv_wgrad_workspace_f16(ctypes.byref(d)), 4) // 4)
    M = n * hout * wout
    xs = torch.full((x_nhwc.numel() // 4, 8), float("nan"), dtype=torch.float16)     # [pixel*c/4][hi x4 | lo x4]
    assert lib.dcn_split_act_f16(L.ptr(x_nhwc), L.ptr(xs), x_nhwc.numel(), None) == 0
    assert rel_err((xs[:, :4].float() + xs[:, 4:].float()).reshape(x_nhwc.shape), x_nhwc) < 1e-6
    qb = lib.dcn_grad_blocked_bytes(M, cout)
    mq = (M + 3) // 4
    assert qb == mq * 4 * cout * 4
    dq = torch.full((mq, 4, cout // 4, 2, 4), float("nan"), dtype=torch.float16)     # [m/4][sub][c/4][2 ch][4 px]
    assert lib.dcn_split_grad_blocked_f16(L.ptr(dout), M, cout, L.ptr(amax), L.ptr(dq), None) == 0
    rec = dq[:, :2].float() + dq[:, 2:].float()                                       # [mq][2 pairs][c/4][2][4 px]
    rec = rec.permute(0, 4, 2, 1, 3).reshape(mq * 4, cout)                            # -> [pixel][channel]
    scale = float(rec[:M].abs().max() / dout.abs().max())
    assert abs(math.log2(scale) - round(math.log2(scale))) < 1e-3 and 1024 < scale * float(amax) <= 4096
    assert rel_err(rec[:M] / scale, dout.reshape(M, cout)) < 1e-6 and float(rec[M:].abs().sum()) == 0
    assert lib.dcn_conv_wgrad_f16(ctypes.byref(d), L.ptr(xs), 0, None, L.ptr(dq), L.ptr(amax), L.ptr(dw), L.ptr(slabs), None) == 0
    # same result when the activation operand is the fp32 tensor itself, split on the fly
    dw_direct = torch.full((cout, k, k, cin), float("nan"))
    assert lib.dcn_conv_wgrad_f16(ctypes.byref(d), L.ptr(x_nhwc), 1, None, L.ptr(dq), L.ptr(amax), L.ptr(dw_direct),
                                  L.ptr(slabs), None) == 0
    assert torch.equal(dw_direct, dw)
    xb = (x_nhwc * 2.0e5).contiguous()     # out-of-range activations with their abs-max: pre-scaled operand
    xmax = xb.abs().max().reshape(1).clone()
    dw_big = torch.full((cout, k, k, cin), float("nan"))
    assert lib.dcn_conv_wgrad_f16(ctypes.byref(d), L.ptr(xb), 1, L.ptr(xmax), L.ptr(dq), L.ptr(amax), L.ptr(dw_big),
                                  L.ptr(slabs), None) == 0
    assert rel_err(dw_big, w.grad.permute(0, 2, 3, 1) * 2.0e5) < 5e-6
    assert rel_err(dw, w.grad.permute(0, 2, 3, 1)) < 5e-6


def test_batched_weight_split_equals_per_tensor_kernels(L):
    """dcn_split_weights_f16 (all weight tensors in one launch; more than one table's worth) against the per-tensor
    dcn_split_rows_f16 / dcn_transpose_weight path, bit for bit, forward and transposed images."""
    lib = L.get()
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 49, 4, 64), (64, 9, 64, 64), (128, 9, 64, 128), (3, 1, 32, 4), (40, 1, 24, 40), (8, 9, 8, 8)] * 11   # 66 > 56
    ws = [torch.randn(co, tp, ci, generator=g) * 0.1 for co, tp, ci, _ in shapes]
    n = len(shapes)
    VP, I = ctypes.c_void_p * n, ctypes.c_int * n
    for transposed in (0, 1):
        his, los, refs = [], [], []
        for (co, tp, ci, ldn), w in zip(shapes, ws):
            rows, K = (ci, tp * ldn) if transposed else (co, tp * ci)
            kp = lib.dcn_f16_kpad(K)
            his.append(torch.full((rows, kp), float("nan"), dtype=torch.float16))
            los.append(torch.full((rows, kp), float("nan"), dtype=torch.float16))
            rh, rl = torch.empty(rows, kp, dtype=torch.float16), torch.empty(rows, kp, dtype=torch.float16)
            src = w
            if transposed:
                src = torch.empty(ci, tp, ldn)
                assert lib.dcn_transpose_weight(L.ptr(w), L.ptr(src), co, tp, ci, ldn, None) == 0
            assert lib.dcn_split_rows_f16(L.ptr(src), L.ptr(rh), L.ptr(rl), rows, K, 64.0, None) == 0
            refs.append((rh, rl))
        rc = lib.dcn_split_weights_f16(n, VP(*[w.data_ptr() for w in ws]), VP(*[t.data_ptr() for t in his]),
                                       VP(*[t.data_ptr() for t in los]), I(*[s[0] for s in shapes]), I(*[s[1] for s in shapes]),
                                       I(*[s[2] for s in shapes]), I(*[s[3] for s in shapes]), transposed, 64.0, None)
        assert rc == 0
        for (rh, rl), h, l_ in zip(refs, his, los):
            assert torch.equal(h, rh) and torch.equal(l_, rl)


def test_match_statistics_vs_reference_golden(L):
    """evaluation.py:1046-1100 for 12 matches in one pass, against the outputs of the reference's own source lines.  Counts:
    the reference compares two float32 norms computed by different summation orders, so a pixel whose distance ties with the
    ground truth (the ground-truth pixel itself, here) may or may not be counted there: +-1."""
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork as DCN
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_ref.npz"))
    res_a, res_b, mask = torch.tensor(z["res_a"]), torch.tensor(z["res_b"]), torch.tensor(z["mask_b"])
    uv = torch.tensor(z["uv"])
    s = DCN.compute_match_statistics(uv, uv, res_a, res_b, mask)
    assert np.array_equal(s["uv_b_pred"].numpy(), z["uv_b_pred"].astype(np.int64))
    assert np.array_equal(s["uv_b_pred_masked"].numpy(), z["uv_b_pred_masked"].astype(np.int64))
    np.testing.assert_allclose(s["norm_diff_pred"].numpy(), z["best_match_diff"], rtol=1e-5)
    np.testing.assert_allclose(s["norm_diff_pred_masked"].numpy(), z["best_match_diff_masked"], rtol=1e-5)
    np.testing.assert_allclose(s["norm_diff_descriptor_ground_truth"].numpy(), z["norm_diff_descriptor_ground_truth"], rtol=1e-5)
    np.testing.assert_allclose(s["pixel_match_error_l2"].numpy(), z["pixel_match_error_l2"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(s["pixel_match_error_l2_masked"].numpy(), z["pixel_match_error_l2_masked"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(s["pixel_match_error_l1"].numpy(), z["pixel_match_error_l1"], rtol=1e-6, atol=1e-6)
    for name in ("", "_masked"):
        c = s["num_pixels_closer_than_ground_truth" + name].numpy().astype(np.int64)
        assert np.abs(c - z["num_pixels_closer_than_ground_truth" + name]).max() <= 1
        big = z["num_pixels_closer_than_ground_truth" + name] > 20      # the averages: where one tied pixel cannot matter
        np.testing.assert_allclose(s["average_l2_distance_for_false_positives" + name].numpy()[big],
                                   z["average_l2_distance_for_false_positives" + name][big], rtol=0.1)
    # no mask: the "masked" half equals the image half
    s2 = DCN.compute_match_statistics(uv, uv, res_a, res_b)
    assert torch.equal(s2["uv_b_pred"], s2["uv_b_pred_masked"])
    assert torch.equal(s2["num_pixels_closer_than_ground_truth"], s2["num_pixels_closer_than_ground_truth_masked"])


def test_match_statistics_wide_descriptors_and_degenerate_masks(L):
    """D = 16 (specialised instance) and D = 5 (runtime width); a mask of one pixel; the oracle per query."""
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork as DCN
    from oracle import evaluation_oracle as eo
    g = torch.Generator().manual_seed(1)
    for D in (16, 5):
        H, W, Q = 20, 28, 6
        res_a = torch.randn(H, W, D, generator=g)
        res_b = res_a + 0.4 * torch.randn(H, W, D, generator=g)
        mask = torch.zeros(H, W)
        mask[7, 9] = 1
        uv = torch.stack([torch.randint(0, W, (Q,), generator=g), torch.randint(0, H, (Q,), generator=g)], 1)
        s = DCN.compute_match_statistics(uv, uv, res_a, res_b, mask)
        for q in range(Q):
            o = eo.match_statistics((int(uv[q, 0]), int(uv[q, 1])), (int(uv[q, 0]), int(uv[q, 1])), res_a.numpy(), res_b.numpy(),
                                    mask.numpy())
            assert tuple(s["uv_b_pred"][q].tolist()) == tuple(int(x) for x in o["uv_b_pred"])
            assert tuple(s["uv_b_pred_masked"][q].tolist()) == (9, 7)
            assert abs(int(s["num_pixels_closer_than_ground_truth"][q]) - o["num_pixels_closer_than_ground_truth"]) <= 1
            assert int(s["num_pixels_closer_than_ground_truth_masked"][q]) in (0, 1)


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, dil, groups, add, relu
    (1, 12, 16, 32, 24, 3, 1, 1, True, True),      # UNI path (cs = ldc % 32 != 0 here: generic path), one M tile
    (2, 16, 16, 64, 32, 3, 2, 1, True, True),      # dilation 2, several M tiles, uniform-tap path (ldc = 32)
    (2, 8, 16, 8, 64, 1, 1, 1, False, True),       # 1x1 (bottleneck conv3 -> bn2), no residual gradient
    (2, 16, 16, 16, 32, 3, 1, 2, True, True),      # two statistics groups: tiles must not straddle them
    (1, 10, 12, 12, 8, 3, 1, 1, True, False),      # no ReLU behind the batch norm (mask NULL), ragged tile
])
@pytest.mark.parametrize("sk", ["0", "3"])
def test_dgrad_with_fused_bn_backward_reduction(L, case, sk, dcn_env):
    import kernel_checks
    dcn_env(DCN_GEMM_SK=sk, DCN_GEMM_TILE_M="64")
    n, h, w, cin, cout, k, dil, groups, add, relu = case
    kernel_checks.check_dgrad_with_bn_backward(L, "cpu", n, h, w, cin, cout, k, dil, groups, add, relu)


def test_stream_k_inline_completion(L, dcn_env):
    import kernel_checks
    kernel_checks.check_stream_k_inline(L, "cpu", dcn_env, 1, 12, 12, 32, 136, 3, 2, sk="5", tile_m="256")
    kernel_checks.check_stream_k_inline(L, "cpu", dcn_env, 2, 9, 9, 32, 40, 3, 1, sk="7", tile_m="64")


@pytest.mark.parametrize("case", [
    (1, 8, 36, 8, 16, 3, 1, 1, 1),       # 64-channel tile, carried-position path, several stages
    (2, 9, 7, 4, 12, 7, 2, 3, 1),        # stem-like, division path
    (1, 6, 40, 16, 136, 3, 1, 2, 2),     # 128-channel tile, two channel tiles
    (1, 4, 36, 8, 256, 3, 1, 1, 1),      # 256-channel / 8-wavefront tile
    (2, 5, 6, 32, 512, 1, 1, 0, 1),
    (1, 1, 32, 8, 256, 3, 1, 1, 1),      # a single 32-pixel stage per split: prologue loads run past the end
    (2, 3, 64, 16, 512, 3, 1, 2, 2),     # two 256-channel tiles, dilation
], ids=str)
def test_wgrad_f16_deep_prefetch_is_bit_identical(L, case, dcn_env):
    """DCN_WGRAD_DEEP: the global loads of a stage travel through two register sets (issued 1.5 iterations ahead) instead of
    one -- the same values reach the same MFMAs in the same order: identical bits, odd and even stage counts, stages past
    the end of a split (out-of-range loads that are never stored)."""
    lib = L.get()
    n, hin, win, cin, cout, k, stride, pad, dil = case
    hout = (hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wout = (win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    d = L.ConvDesc(n, hin, win, cin, hout, wout, cout, k, k, stride, pad, dil, cout)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, hin, win, cin, generator=g)
    dout = torch.randn(n, hout, wout, cout, generator=g)
    M = n * hout * wout
    amax = dout.abs().max().reshape(1).clone()
    dq = torch.empty(lib.dcn_grad_blocked_bytes(M, cout) // 4)
    assert lib.dcn_split_grad_blocked_f16(L.ptr(dout), M, cout, L.ptr(amax), L.ptr(dq), None) == 0
    res = []
    # (DCN_WGRAD_ROLES: wide tile only -- wavefronts 0-3 stage the activations, 4-7 the whole gradient tile)
    for deep, roles in ((0, 0), (7, 0), (4, 1), (0, 1)):
        dcn_env(DCN_WGRAD_DEEP=deep, DCN_WGRAD_ROLES=roles)
        slabs = torch.empty(max(lib.dcn_conv_wgrad_workspace_f16(ctypes.byref(d)), 4) // 4)
        dw = torch.full((cout, k, k, cin), float("nan"))
        assert lib.dcn_conv_wgrad_f16(ctypes.byref(d), L.ptr(x), 1, None, L.ptr(dq), L.ptr(amax), L.ptr(dw), L.ptr(slabs), None) == 0
        res.append(dw)
    ref = F.conv2d(x.permute(0, 3, 1, 2).requires_grad_(False), torch.zeros(cout, cin, k, k), None, stride, pad, dil)  # shape check only
    assert ref.shape[2:] == (hout, wout)
    w = torch.zeros(cout, cin, k, k, requires_grad=True)
    (F.conv2d(x.permute(0, 3, 1, 2), w, None, stride, pad, dil) * dout.permute(0, 3, 1, 2)).sum().backward()
    assert rel_err(res[0], w.grad.permute(0, 2, 3, 1)) < 5e-6
    assert all(torch.equal(res[0], r) for r in res[1:])


@pytest.mark.parametrize("shape", [(2, 18, 26), (1, 9, 8), (1, 32, 40)], ids=str)
def test_stem_through_uniform_tap_path(L, shape):
    import kernel_checks
    kernel_checks.check_stem_uniform_tap(L, "cpu", *shape)


HL_CASES = [
    # n, h, w, cin, cout, k, dil, stream-K workgroups (None: as decided -> plain launch at these sizes), x scale
    (1, 12, 20, 32, 40, 3, 2, None, 1.0),     # one ragged 256 x 256 tile, ragged channels, dilation 2, one chunk per tap
    (2, 16, 24, 64, 288, 3, 1, None, 1.0),    # 3 x 2 tiles, second N tile ragged, two chunks per tap (chunk groups of 2)
    (2, 16, 24, 64, 288, 3, 1, "5", 1e4),     # the same through stream-K: segments cross tiles, partials completed in-launch
    (1, 20, 20, 128, 256, 1, 1, "5", 1.0),    # 1x1: taps = 1, chunk groups of 4; M = 400 (ragged second tile)
    (1, 9, 30, 96, 256, 3, 4, "3", 1e-6),     # dilation 4 with halos wider than the border, 3 chunks per tap, tiny operands
    (1, 10, 30, 32, 64, 1, 1, None, 1.0),     # K = 32: ONE stage (prologue straight into the last-stage phases), 2 M tiles
    (1, 10, 30, 64, 64, 1, 1, None, 1.0),     # K = 64: two stages (no steady-state stage)
]


@pytest.mark.parametrize("rows", ["256", "192", "320"])
@pytest.mark.parametrize("dma", ["late", "early"])
@pytest.mark.parametrize("case", HL_CASES, ids=[str(c) for c in HL_CASES])
def test_conv_hl32_lds_dma_gather_gemm(L, case, dma, rows, dcn_env, monkeypatch):
    """conv_hl_kernels.hip on the host: LDS-DMA pieces land either when the issuing work-item's counted wait retires them
    (the latest moment the hardware allows: catches reads that are not covered by wait + barrier) or at issue (the earliest:
    catches a buffer restaged while it is still being read)."""
    import kernel_checks
    if dma == "early" and case[7] is None and case[1] == 12:
        pytest.skip("one tile, plain launch: covered by the late mode")
    monkeypatch.setenv("HIPEMU_LDS_DMA", dma)
    n, h, w, cin, cout, k, dil, sk, sx = case
    if rows == "320" and sk is not None:
        pytest.skip("320-row tiles run data-parallel launches only (hl_shape)")
    kernel_checks.check_conv_hl(L, "cpu", n, h, w, cin, cout, k, dil, set_env=dcn_env, sk=sk, scale_x=sx, seed=len(str(case)),
                                rows=rows)


HLX_CASES = [
    # n, h, w, cin, cout, k, dil, "kg,splits" forced, x scale
    (1, 12, 20, 32, 40, 3, 2, "1,1", 1.0),     # 160 x 256 tile: two ragged M tiles (240 rows), ragged channels, one chunk per tap
    (2, 16, 24, 64, 288, 3, 1, "1,1", 1.0),    # 5 x 2 tiles, second N tile ragged, chunk groups of 2
    (2, 16, 24, 64, 288, 3, 1, "1,3", 1e4),    # ... K split over 3 workgroups per tile: partials completed in-launch, fixed order
    (2, 16, 24, 64, 288, 3, 1, "2,1", 1.0),    # 160 x 128 tile, two K groups added through LDS; 3 N tiles (third ragged)
    (1, 20, 20, 128, 256, 1, 1, "2,2", 1.0),   # 1x1: ONE tap, chunk groups of 4 = two stages of the K-group kernel, split in two
    (1, 9, 30, 64, 256, 3, 4, "2,3", 1e-6),    # dilation 4 with halos wider than the border, tiny operands, 3 splits of 3 stages
    (1, 10, 30, 32, 64, 1, 1, "1,1", 1.0),     # K = 32: ONE stage
    (1, 10, 30, 64, 64, 1, 1, "2,1", 1.0),     # K = 64: ONE stage of the K-group kernel
]


@pytest.mark.parametrize("dma", ["late", "early"])
@pytest.mark.parametrize("case", HLX_CASES, ids=[str(c) for c in HLX_CASES])
def test_conv_hlx_small_tiles(L, case, dma, dcn_env, monkeypatch):
    """conv_hlx_kernels.hip on the host (160 x 256 and 160 x 128 tiles of 16 x 16 x 32 MFMAs, K groups inside the workgroup, K
    split over workgroups): same checks as the big tiles, LDS-DMA landing as late and as early as the hardware allows."""
    import kernel_checks
    monkeypatch.setenv("HIPEMU_LDS_DMA", dma)
    n, h, w, cin, cout, k, dil, hlx, sx = case
    kernel_checks.check_conv_hl(L, "cpu", n, h, w, cin, cout, k, dil, set_env=dcn_env, scale_x=sx, seed=len(str(case)), hlx=hlx)


@pytest.mark.parametrize("switch", ["DCN_HLX_STAGGER=0", "DCN_HLX_STAGGER=2", "DCN_HLX_COUNTERS=0"])
@pytest.mark.parametrize("dma", ["late", "early"])
@pytest.mark.parametrize("case", [HLX_CASES[2], HLX_CASES[5], HLX_CASES[7]], ids=str)
def test_conv_hlx_schedule_and_counter_switches(L, case, switch, dma, dcn_env, monkeypatch):
    """The switches of the small-tile kernel: every wavefront issuing its LDS-DMA in front of the compute slot (instead of
    wavefronts 4-7 between its two parts); two wavefront groups one slot apart (LOAD slot | barrier | COMPUTE slot | barrier --
    its own RAW / WAR argument, hence both DMA landing modes); the K splits' arrival words in the caller's scratch behind a fill
    launch (instead of the library's clean per-stream buffer)."""
    import kernel_checks
    monkeypatch.setenv("HIPEMU_LDS_DMA", dma)
    monkeypatch.setenv(*switch.split("="))
    n, h, w, cin, cout, k, dil, hlx, sx = case
    kernel_checks.check_conv_hl(L, "cpu", n, h, w, cin, cout, k, dil, set_env=dcn_env, scale_x=sx, seed=len(str(case)), hlx=hlx)


WGRAD_HLR_CASES = [
    # n, h, w, cin, cout, k, dil, forced splits   (row-window kernel of the narrow 3 x 3 layers: DCN_WGRAD_HLR=2)
    (1, 3, 40, 64, 64, 3, 1, None),      # cin = 64: two segments per row (32 + 8 pixels), image borders on every side of every stage
    (2, 4, 33, 128, 128, 3, 1, "3"),     # cin = 128 (two input blocks per wavefront), two 64-channel tiles, three stage ranges over two images
    (1, 2, 20, 64, 192, 3, 1, "1"),      # rows shorter than a stage, three tiles, ONE range: the kernel writes dw itself (no slabs)
    (1, 5, 64, 128, 64, 3, 1, "5"),      # whole segments only (no ragged one), five ranges of two stages
    (2, 20, 32, 64, 64, 3, 1, "40"),     # 40 ranges of ONE stage: the slabs are summed by the many-splits reduce kernel (>= 32)
]


@pytest.mark.parametrize("pairs", [1, 0])
@pytest.mark.parametrize("dma", ["late", "early"])
@pytest.mark.parametrize("case", WGRAD_HLR_CASES, ids=[str(c) for c in WGRAD_HLR_CASES])
def test_wgrad_hl32_row_window_kernel(L, case, dma, pairs, dcn_env, monkeypatch):
    """conv_wgrad_hlrp_kernel / conv_wgrad_hlr_kernel (wgrad_hl_kernels.hip): 64 output channels x nine taps x 64 input channels
    per workgroup, row windows of x per 32-pixel stage (row pairs: two image rows per stage, one per wavefront group, sums joined
    through LDS; DCN_WGRAD_HLR_PAIRS=0: one row) -- against float64 autograd and the fp32-operand kernel, both LDS-DMA landing
    modes; odd image heights leave the last pair's second row empty."""
    if pairs == 0 and case not in (WGRAD_HLR_CASES[0], WGRAD_HLR_CASES[1]):
        pytest.skip("single-row form: two cases")
    monkeypatch.setenv("DCN_WGRAD_HLR_PAIRS", str(pairs))
    import kernel_checks
    monkeypatch.setenv("HIPEMU_LDS_DMA", dma)
    n, h, w, cin, cout, k, dil, splits = case
    dcn_env(DCN_WGRAD_HLR=2)
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, dil * (k - 1) // 2, dil, cout, 0)
    assert L.get().dcn_conv_wgrad_hl_kind(ctypes.byref(d)) == 2
    kernel_checks.check_wgrad_hl(L, "cpu", n, h, w, cin, cout, k, dil, set_env=dcn_env, splits=splits, seed=len(str(case)))


def test_wgrad_hl32_kernel_choice(L, dcn_env):
    """Which weight-gradient kernel dcn_conv_wgrad_hl launches (dcn_conv_wgrad_hl_kind / _eligible; host logic only): the row-window
    kernel for the 64 / 128-channel 3 x 3 stride-1 dilation-1 layers from four images at 640 x 480, the 256 x 256 tile kernel for
    whole 256-channel output tiles, neither for the rest (strided, 1 x 1, dilated narrow layers stay on the fp32-operand kernel)."""
    lib = L.get()
    dcn_env(DCN_WGRAD_HLR=1)

    def kind(n, h, w, cin, cout, k, stride=1, dil=1):
        pad = dil * (k - 1) // 2
        d = L.ConvDesc(n, h, w, cin, (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1,
                       cout, k, k, stride, pad, dil, cout, 0)
        return lib.dcn_conv_wgrad_hl_kind(ctypes.byref(d)), lib.dcn_conv_wgrad_hl_eligible(ctypes.byref(d))
    assert kind(8, 120, 160, 64, 64, 3) == (2, 1)          # layer 1 of config 2
    assert kind(8, 60, 80, 128, 128, 3) == (2, 1)          # layer 2
    assert kind(4, 120, 160, 64, 64, 3) == (2, 1)          # the two-call pattern: four images per call
    assert kind(4, 240, 320, 64, 64, 3) == (2, 1)          # ResNet50-8s at 1280 x 960 (config 5)
    assert kind(2, 120, 160, 64, 64, 3) == (1, 0)          # config 1: too few stages per workgroup -- fp32-operand kernel
    assert kind(8, 60, 80, 128, 256, 3, dil=2) == (1, 1)   # layer3.0.conv1: whole 256-channel tiles -- tile kernel
    assert kind(8, 60, 80, 256, 256, 3, dil=2) == (1, 1)
    assert kind(8, 120, 160, 64, 128, 3, stride=2)[1] == 0   # strided
    assert kind(8, 120, 160, 64, 64, 1)[1] == 0              # 1 x 1
    assert kind(8, 60, 80, 128, 128, 3, dil=2)[1] == 0       # dilated narrow layer
    dcn_env(DCN_WGRAD_HLR=0)
    assert kind(8, 120, 160, 64, 64, 3) == (1, 0)
    dcn_env(DCN_WGRAD_HLR=2)
    assert kind(1, 8, 20, 64, 64, 3) == (2, 1) and kind(8, 60, 80, 128, 256, 3) == (2, 1)


WGRAD_HL_CASES = [
    # n, h, w, cin, cout, k, dil, forced splits
    (1, 3, 40, 32, 64, 3, 1, None),      # one ragged tile (64 of 256 output channels, K = 288 of 2 x 256), 120 pixels = 4 stages
    (2, 5, 36, 64, 288, 3, 2, "3"),      # two channel tiles (second ragged), dilation 2, rows wrap inside stages, 3 pixel splits
    (1, 8, 32, 256, 256, 1, 1, "2"),     # 1x1: one tap, full tiles, two splits of 4 stages
    (1, 4, 48, 96, 32, 3, 4, None),      # dilation 4 with halos wider than the image, 3 chunks per tap (taps change inside a K tile)
]


@pytest.mark.parametrize("dma", ["late", "early"])
@pytest.mark.parametrize("case", WGRAD_HL_CASES, ids=[str(c) for c in WGRAD_HL_CASES])
def test_wgrad_hl32_transposing_lds_reads(L, case, dma, dcn_env, monkeypatch):
    import kernel_checks
    monkeypatch.setenv("HIPEMU_LDS_DMA", dma)
    n, h, w, cin, cout, k, dil, splits = case
    kernel_checks.check_wgrad_hl(L, "cpu", n, h, w, cin, cout, k, dil, set_env=dcn_env, splits=splits, seed=len(str(case)))
