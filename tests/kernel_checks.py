"""Kernel-level checks shared by the CPU (host-emulated) and the -m gpu suites (not a test module): the same C-ABI calls
against torch CPU ops, on `device` ("cpu" with the emulation library loaded, "cuda" with libdcn_hip.so)."""
import ctypes

import torch
import torch.nn.functional as F

from helpers import rel_err


def _split_rows(L, lib, w2d, dev):
    rows, K = w2d.shape
    kp = lib.dcn_f16_kpad(K)
    hi = torch.empty(rows, kp, dtype=torch.float16, device=dev)
    lo = torch.empty(rows, kp, dtype=torch.float16, device=dev)
    assert lib.dcn_split_rows_f16(L.ptr(w2d), L.ptr(hi), L.ptr(lo), rows, K, 64.0, None) == 0
    return hi, lo


def garbage(nbytes, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(-2 ** 31, 2 ** 31 - 1, (max(nbytes, 8) // 4 + 2,), generator=g, dtype=torch.int32)
    return t.to(dev)


def check_dgrad_with_bn_backward(L, dev, n, h, w, cin, cout, k, dil, groups=1, with_add=True, relu=True, seed=0):
    """conv (stride 1, `same` padding) behind a train-mode batch norm + ReLU:  x -> BN -> ReLU -> y -> conv -> out.
    dcn_conv_dgrad_bn_f16 must deliver the ReLU-masked gradient w.r.t. y and the per-tile sums from which
    dcn_bn_backward_from_partial produces dx / dgamma / dbeta -- against torch autograd on the CPU (float64)."""
    lib = L.get()
    g = torch.Generator().manual_seed(seed)
    pad = dil * (k - 1) // 2
    x = torch.randn(n, h, w, cin, generator=g) * 1.5 + 0.3            # NHWC: the batch norm's input (a conv output)
    gamma = torch.rand(cin, generator=g) + 0.5
    beta = torch.randn(cin, generator=g) * 0.2
    wt_ = torch.randn(cout, cin, k, k, generator=g) * 0.1
    dout = torch.randn(n, h, w, cout, generator=g) * 1e-3
    add = torch.randn(n, h, w, cin, generator=g) * 1e-3 if with_add else None
    rows = n * h * w
    rpg = rows // groups
    # ---- reference (float64, per statistics group)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    ys = []
    for gi in range(groups):
        xg = xd.reshape(groups, rpg, cin)[gi]
        ys.append(F.batch_norm(xg, None, None, gd, bd, True, 0.1, 1e-5))
    y = torch.stack(ys).reshape(n, h, w, cin)
    if relu:
        y = torch.relu(y)
    y.retain_grad()
    out = F.conv2d(y.permute(0, 3, 1, 2), wt_.double(), None, 1, pad, dil)
    loss = (out * dout.double().permute(0, 3, 1, 2)).sum()
    if add is not None:
        loss = loss + (y * add.double()).sum()
    loss.backward()
    g_ref = y.grad * ((y > 0) if relu else 1.0)                         # masked gradient w.r.t. the BN output
    # ---- device: forward statistics through dcn_bn_forward (partial sums of ONE tile per group = the group's column sums)
    t = lambda a: a.to(dev).contiguous()
    xdev = t(x)
    gam_d, bet_d, dout_d = t(gamma), t(beta), t(dout)   # (kept alive: the launches are asynchronous on the GPU)
    add_d = t(add) if add is not None else None
    keep = []
    stats = torch.empty(groups, 4, cin, device=dev)
    ydev = torch.empty(n, h, w, cin, device=dev)
    mask = torch.zeros(rows * cin // 4, dtype=torch.uint8, device=dev)
    for gi in range(groups):
        xg = x.reshape(groups, rpg, cin)[gi]
        part = t(torch.stack([xg.double().sum(0), (xg.double() ** 2).sum(0), xg.abs().amax(0).double()]).float().reshape(1, 3, cin))
        keep.append(part)
        xs = xdev.reshape(groups, rpg, cin)[gi]
        rc = lib.dcn_bn_forward(L.ptr(xs), L.ptr(part), 1, cin, rpg, L.ptr(gam_d), L.ptr(bet_d), None, None, 0.1, 1e-5, 1,
                                None, 1 if relu else 0, L.ptr(ydev.reshape(groups, rpg, cin)[gi]),
                                L.ptr(mask.reshape(groups, rpg * cin // 4)[gi]) if relu else None, L.ptr(stats[gi]), None)
        assert rc == 0
    assert rel_err(ydev.cpu(), y.detach()) < 1e-5
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, pad, dil, cout, rpg if groups > 1 else 0)
    w_k = wt_.permute(0, 2, 3, 1).contiguous()                           # [cout][kh][kw][cin]
    wtr = t(w_k.reshape(cout, k * k, cin).permute(2, 1, 0).contiguous().reshape(cin, k * k * cout))
    wth, wtl = _split_rows(L, lib, wtr, dev)
    amax = t(dout.abs().max().reshape(1))
    mt = lib.dcn_conv_dgrad_bn_num_mtiles_f16(ctypes.byref(d))
    assert mt >= groups and mt % groups == 0
    bn_part = torch.full((mt, cin, 4), float("nan"), device=dev)
    ws = garbage(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 1), dev, seed + 1)
    din = torch.full((n, h, w, cin), float("nan"), device=dev)
    rc = lib.dcn_conv_dgrad_bn_f16(ctypes.byref(d), L.ptr(dout_d), L.ptr(wth), L.ptr(wtl), 64.0, L.ptr(amax),
                                   L.ptr(add_d), L.ptr(din), L.ptr(xdev),
                                   L.ptr(mask) if relu else None, L.ptr(stats), L.ptr(bn_part), L.ptr(ws), None)
    assert rc == 0
    e_g = rel_err(din.cpu(), g_ref)
    assert e_g < 1e-5, e_g
    # per-tile sums: column 0 totals = dbeta, column 1 totals = dgamma (per group; the parameters are shared)
    bp = bn_part.cpu().double()
    assert bool(torch.isfinite(bp).all())
    assert rel_err(bp[:, :, 0].sum(0), bd.grad) < 2e-5
    assert rel_err(bp[:, :, 1].sum(0), gd.grad) < 2e-5
    assert rel_err(bp[:, :, 2].amax(0), g_ref.abs().amax((0, 1, 2))) < 1e-5
    out = {"masked_grad": e_g, "mtiles": mt}
    if groups == 1:
        dgam = torch.empty(cin, device=dev)
        dbet = torch.empty(cin, device=dev)
        dx = torch.full((n, h, w, cin), float("nan"), device=dev)
        ws3 = torch.empty(3 * cin, device=dev)
        rc = lib.dcn_bn_backward_from_partial(L.ptr(din), L.ptr(bn_part), mt, L.ptr(xdev), L.ptr(stats), L.ptr(gam_d), cin,
                                              rows, L.ptr(dgam), L.ptr(dbet), L.ptr(dx), L.ptr(ws3), None)
        assert rc == 0
        out.update(dx=rel_err(dx.cpu(), xd.grad), dgamma=rel_err(dgam.cpu(), gd.grad), dbeta=rel_err(dbet.cpu(), bd.grad))
        assert out["dx"] < 2e-5 and out["dgamma"] < 2e-5 and out["dbeta"] < 2e-5, out
    return out


def check_stream_k_inline(L, dev, set_env, n, h, w, cin, cout, k, dil, sk, tile_m=None, repeats=3, seed=0):
    """Stream-K tiles completed inside the GEMM launch (default) vs by the separate fix-up kernel
    (DCN_GEMM_SK_FIXUP=kernel): bit-identical outputs and batch-norm partial sums, on a garbage-filled workspace, again
    and again on the workspace the previous launches left behind."""
    lib = L.get()
    g = torch.Generator().manual_seed(seed)
    pad = dil * (k - 1) // 2
    t = lambda a: a.to(dev).contiguous()
    x = t(torch.randn(n, h, w, cin, generator=g))
    wt_ = torch.randn(cout, k, k, cin, generator=g) * 0.1
    w2d = t(wt_.reshape(cout, k * k * cin))
    wh, wl = _split_rows(L, lib, w2d, dev)
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, pad, dil, cout, 0)
    env = {"DCN_GEMM_SK": sk}
    if tile_m:
        env["DCN_GEMM_TILE_M"] = tile_m
    res = {}
    for mode in ("kernel", "inline"):
        set_env(DCN_GEMM_SK_FIXUP=mode, **env)
        mt = lib.dcn_conv_num_mtiles_f16(ctypes.byref(d))
        nbytes = lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 0)
        assert nbytes > 8, "stream-K is not exercised by this shape"
        ws = garbage(nbytes, dev, seed + 7)
        runs = []
        for _ in range(repeats if mode == "inline" else 1):
            out = torch.full((n, h, w, cout), float("nan"), device=dev)
            part = torch.full((mt, 3, cout), float("nan"), device=dev)
            assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(x), None, L.ptr(wh), L.ptr(wl), 64.0, None, L.ptr(out),
                                            L.ptr(part), L.ptr(ws), None) == 0
            runs.append((out.cpu(), part.cpu()))
        res[mode] = runs
    ref = F.conv2d(x.cpu().permute(0, 3, 1, 2), wt_.permute(0, 3, 1, 2), None, 1, pad, dil).permute(0, 2, 3, 1)
    o_k, p_k = res["kernel"][0]
    assert rel_err(o_k, ref) < 5e-6
    for o_i, p_i in res["inline"]:
        assert torch.equal(o_i, o_k) and torch.equal(p_i, p_k)
    return True


def check_stem_uniform_tap(L, dev, n, hin, win):
    """dcn_conv_stem_forward_f16: the 7x7 / 2 stem with a filter row as one 32-K chunk (8 pixels x 4 channels, 8th pixel zero
    weights, per-work-item column validity) == the generic gather path == F.conv2d, borders and odd sizes included; a NaN in
    the 8th (unused) column of a window must not leak into the result."""
    lib = L.get()
    cout = 16
    hout, wout = (hin + 6 - 7) // 2 + 1, (win + 6 - 7) // 2 + 1
    d = L.ConvDesc(n, hin, win, 4, hout, wout, cout, 7, 7, 2, 3, 1, cout)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, hin, win, 4, generator=g)
    x[..., 3] = 0
    xd = x.to(dev)
    w4 = torch.randn(cout, 7, 7, 4, generator=g) * 0.1
    w4[..., 3] = 0
    w4d = w4.to(dev)
    hi = torch.empty(cout, 224, dtype=torch.float16, device=dev)
    lo = torch.empty(cout, 224, dtype=torch.float16, device=dev)
    assert lib.dcn_split_stem_weights_f16(L.ptr(w4d), L.ptr(hi), L.ptr(lo), cout, 64.0, None) == 0
    rec = ((hi.float() + lo.float()) / 64.0).reshape(cout, 7, 8, 4).cpu()
    assert rel_err(rec[:, :, :7], w4) < 1e-6 and float(rec[:, :, 7].abs().max()) == 0
    mt = lib.dcn_conv_num_mtiles_f16(ctypes.byref(d))
    out = torch.full((n, hout, wout, cout), float("nan"), device=dev)
    part = torch.full((mt, 3, cout), float("nan"), device=dev)
    assert lib.dcn_conv_stem_forward_f16(ctypes.byref(d), L.ptr(xd), None, L.ptr(hi), L.ptr(lo), 64.0, L.ptr(out), L.ptr(part), None) == 0
    ref = F.conv2d(x.permute(0, 3, 1, 2), w4.permute(0, 3, 1, 2), None, 2, 3, 1).permute(0, 2, 3, 1)
    assert rel_err(out.cpu(), ref) < 3e-6
    assert rel_err(part.cpu().sum(0)[0], ref.sum((0, 1, 2))) < 1e-5
    # generic path on the same tensors
    K = 196
    wh = torch.empty(cout, lib.dcn_f16_kpad(K), dtype=torch.float16, device=dev)
    wl = torch.empty_like(wh)
    w2 = w4d.reshape(cout, K).contiguous()
    assert lib.dcn_split_rows_f16(L.ptr(w2), L.ptr(wh), L.ptr(wl), cout, K, 64.0, None) == 0
    out2 = torch.full_like(out, float("nan"))
    assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(xd), None, L.ptr(wh), L.ptr(wl), 64.0, None, L.ptr(out2), None, None, None) == 0
    assert rel_err(out.cpu(), out2.cpu()) < 2e-6
    if win >= 12:   # a NaN that only ever sits in the 8th column of some window: x = 2*ox - 3 + 7 for ox = 1 -> column 6... use
        xb = x.clone()   # the last pixel of a row seen as column 7 by the window of ox = (win - 1 - 4) / 2 when that is integral
        col = win - 1
        if (col - 4) % 2 == 0:
            ox7 = (col - 4) // 2          # window of ox7 starts at 2*ox7 - 3 = col - 7: col is its 8th pixel
            xb[0, 5 if hin > 5 else 0, col, 0] = float("nan")
            xbd = xb.to(dev)
            out3 = torch.full_like(out, float("nan"))
            assert lib.dcn_conv_stem_forward_f16(ctypes.byref(d), L.ptr(xbd), None, L.ptr(hi), L.ptr(lo), 64.0, L.ptr(out3), None, None) == 0
            refb = F.conv2d(xb.permute(0, 3, 1, 2), w4.permute(0, 3, 1, 2), None, 2, 3, 1).permute(0, 2, 3, 1)
            assert torch.equal(torch.isnan(out3.cpu()), torch.isnan(refb))   # NaN exactly where the true convolution has it


def hl32_image(L, lib, x2d, absmax, dev):
    """[rows][C] fp32 (device) -> its hl32 image (dcn_split_act_hl32) as a float32-typed byte buffer of the same size."""
    rows, C = x2d.shape
    out = torch.empty(rows * C, dtype=torch.float32, device=dev)
    assert lib.dcn_split_act_hl32(L.ptr(x2d), L.ptr(absmax) if absmax is not None else None, L.ptr(out), rows, C, None) == 0
    return out


def hl32_decode(img, rows, C):
    """hl32 byte buffer -> (hi, lo) float32 [rows][C] (host)."""
    h = img.cpu().view(torch.float16).reshape(rows, C // 32, 2, 32).float()
    return h[:, :, 0, :].reshape(rows, C), h[:, :, 1, :].reshape(rows, C)


def check_conv_hl(L, dev, n, h, w, cin, cout, k, dil, set_env=None, sk=None, scale_x=1.0, seed=0, rows=None, hlx=None):
    """The pre-split (hl32) LDS-DMA gather-GEMM, forward and dgrad, against F.conv2d / its autograd in float64 and against
    the fp32-operand split-fp16 kernel; the operand split, the weight images and the batch-norm partial sums on the way."""
    lib = L.get()
    g = torch.Generator().manual_seed(seed)
    pad = dil * (k - 1) // 2
    t = lambda a: a.to(dev).contiguous()
    x = torch.randn(n, h, w, cin, generator=g) * scale_x
    x = torch.relu(x) if seed % 2 else x
    wt_ = torch.randn(cout, k, k, cin, generator=g) * 0.1
    dout = torch.randn(n, h, w, cout, generator=g) * 1e-3 * scale_x
    M = n * h * w
    env = {}
    if sk is not None:
        env["DCN_GEMM_SK"] = sk
    if rows is not None:
        env["DCN_GEMM_HL_ROWS"] = rows      # tile height 256 / 192 / 320 (conv_hl_kernels.hip: three software pipelines)
    if hlx is not None:
        env["DCN_GEMM_HLX"] = hlx           # "kg,splits" of the small-tile kernel (conv_hlx_kernels.hip), 160-row tiles
    if env:
        set_env(**env)
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, pad, dil, cout, 0)
    xd, dd = t(x), t(dout)
    ax, ad = t(x.abs().max().reshape(1)), t(dout.abs().max().reshape(1))
    # ---- operand images
    xi = hl32_image(L, lib, xd.reshape(M, cin), ax, dev)
    hi, lo = hl32_decode(xi, M, cin)
    s = 2.0 ** torch.floor(torch.log2(4096.0 / x.abs().max()))
    assert rel_err((hi + lo) / s, x.reshape(M, cin)) < 3e-7 and float(hi.abs().max()) <= 4096.0
    do_dgrad = cout % 32 == 0            # (the gathered operand of dgrad is the gradient: whole 32-channel chunks)
    di = hl32_image(L, lib, dd.reshape(M, cout), ad, dev) if do_dgrad else None
    K = k * k * cin
    w2d = t(wt_.reshape(cout, K))
    wtr = t(wt_.reshape(cout, k * k, cin).permute(2, 1, 0).contiguous().reshape(cin, k * k * cout))
    w_hl = torch.empty(cout * K, dtype=torch.float32, device=dev)
    wt_hl = torch.empty(cin * k * k * cout, dtype=torch.float32, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    arr = lambda ty, v: (ty * 1)(v)
    assert lib.dcn_split_weights_hl32(1, arr(P, w2d.data_ptr()), arr(P, w_hl.data_ptr()), arr(I, cout), arr(I, k * k), arr(I, cin),
                                      arr(I, cout), 0, 64.0, None) == 0
    whi, wlo = hl32_decode(w_hl, cout, K)
    assert rel_err((whi + wlo) / 64.0, wt_.reshape(cout, K)) < 3e-7
    w4 = t(wt_)
    if do_dgrad:
        assert lib.dcn_split_weights_hl32(1, arr(P, w4.data_ptr()), arr(P, wt_hl.data_ptr()), arr(I, cout), arr(I, k * k),
                                          arr(I, cin), arr(I, cout), 1, 64.0, None) == 0
        thi, tlo = hl32_decode(wt_hl, cin, k * k * cout)
        assert rel_err((thi + tlo) / 64.0, wtr.cpu()) < 3e-7
    # ---- forward
    mt = lib.dcn_conv_num_mtiles_hl(ctypes.byref(d))
    tr = lib.dcn_conv_tile_rows_hl(ctypes.byref(d), 0)
    assert tr in (160, 192, 256, 320) and (rows is None or tr == int(rows)) and mt == (M + tr - 1) // tr
    assert hlx is None or tr == 160
    assert lib.dcn_conv_tile_rows_hl(ctypes.byref(d), 1) in ((160, 192, 256, 320) if rows is None else (int(rows),))
    nws = max(lib.dcn_conv_gemm_workspace_hl(ctypes.byref(d), 0), lib.dcn_conv_gemm_workspace_hl(ctypes.byref(d), 1))
    if (sk is not None and str(sk) != "0") or (hlx is not None and "," in str(hlx) and int(str(hlx).split(",")[1]) > 1):
        assert nws > 8, "stream-K / the K split is not exercised by this shape"
    ws = garbage(nws, dev, seed + 3)
    out = torch.full((n, h, w, cout), float("nan"), device=dev)
    part = torch.full((mt, 3, cout), float("nan"), device=dev)
    assert lib.dcn_conv_forward_hl(ctypes.byref(d), L.ptr(xi), L.ptr(ax), L.ptr(w_hl), 64.0, None, L.ptr(out), L.ptr(part),
                                   L.ptr(ws), None) == 0
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xr, wt_.double().permute(0, 3, 1, 2), None, 1, pad, dil)
    refn = ref.detach().permute(0, 2, 3, 1)
    res = {"fwd": rel_err(out.cpu(), refn)}
    assert res["fwd"] < 5e-6, res
    pc = part.cpu().double()
    assert rel_err(pc[:, 0].sum(0), refn.sum((0, 1, 2))) < 2e-5
    assert rel_err(pc[:, 1].sum(0), (refn ** 2).sum((0, 1, 2))) < 2e-5
    assert rel_err(pc[:, 2].amax(0), refn.abs().amax((0, 1, 2))) < 5e-6
    # same products as the fp32-operand kernel, other summation order
    wh, wl = _split_rows(L, lib, w2d, dev)
    out2 = torch.full_like(out, float("nan"))
    ws2 = garbage(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 0), dev, seed + 4)
    assert lib.dcn_conv_forward_f16(ctypes.byref(d), L.ptr(xd), L.ptr(ax), L.ptr(wh), L.ptr(wl), 64.0, None, L.ptr(out2), None,
                                    L.ptr(ws2), None) == 0
    res["fwd_vs_f16"] = rel_err(out.cpu(), out2.cpu())
    assert res["fwd_vs_f16"] < 6e-6, res
    # ---- dgrad (second launch on the workspace the forward left behind)
    din = torch.full((n, h, w, cin), float("nan"), device=dev)
    if not do_dgrad:
        assert lib.dcn_conv_dgrad_hl(ctypes.byref(d), L.ptr(xi), L.ptr(wt_hl), 64.0, L.ptr(ad), None, L.ptr(din), L.ptr(ws), None) == -3
        return res
    assert lib.dcn_conv_dgrad_hl(ctypes.byref(d), L.ptr(di), L.ptr(wt_hl), 64.0, L.ptr(ad), None, L.ptr(din), L.ptr(ws), None) == 0
    (ref * dout.double().permute(0, 3, 1, 2)).sum().backward()
    res["dgrad"] = rel_err(din.cpu(), xr.grad.permute(0, 2, 3, 1))
    assert res["dgrad"] < 5e-6, res
    # bit-reproducible (stream-K completion order is fixed)
    out3 = torch.full_like(out, float("nan"))
    assert lib.dcn_conv_forward_hl(ctypes.byref(d), L.ptr(xi), L.ptr(ax), L.ptr(w_hl), 64.0, None, L.ptr(out3), None,
                                   L.ptr(ws), None) == 0
    assert torch.equal(out3.cpu(), out.cpu())
    return res


def check_wgrad_hl(L, dev, n, h, w, cin, cout, k, dil, set_env=None, splits=None, seed=0):
    """The weight gradient on hl32 operands (wgrad_hl_kernels.hip: pixel-major tiles by LDS-DMA, k-major fragments by
    transposing LDS reads) against autograd in float64 and against the fp32-operand split-fp16 wgrad kernel."""
    lib = L.get()
    g = torch.Generator().manual_seed(seed)
    pad = dil * (k - 1) // 2
    t = lambda a: a.to(dev).contiguous()
    x = torch.randn(n, h, w, cin, generator=g)
    x = torch.relu(x) if seed % 2 else x
    dout = torch.randn(n, h, w, cout, generator=g) * 1e-3
    M, K = n * h * w, k * k * cin
    if splits is not None:
        set_env(DCN_WGRAD_SPLITS=splits)
    d = L.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, pad, dil, cout, 0)
    xd, dd = t(x), t(dout)
    ax, ad = t(x.abs().max().reshape(1)), t(dout.abs().max().reshape(1))
    xi = hl32_image(L, lib, xd.reshape(M, cin), ax, dev)
    di = hl32_image(L, lib, dd.reshape(M, cout), ad, dev)
    slab = garbage(lib.dcn_conv_wgrad_workspace_hl(ctypes.byref(d)), dev, seed + 5)
    dw = torch.full((cout, k, k, cin), float("nan"), device=dev)
    assert lib.dcn_conv_wgrad_hl(ctypes.byref(d), L.ptr(xi), L.ptr(ax), L.ptr(di), L.ptr(ad), L.ptr(dw), L.ptr(slab), None) == 0
    wz = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(x.double().permute(0, 3, 1, 2), wz, None, 1, pad, dil)
    (out * dout.double().permute(0, 3, 1, 2)).sum().backward()
    ref = wz.grad.permute(0, 2, 3, 1)
    res = {"wgrad": rel_err(dw.cpu(), ref)}
    assert res["wgrad"] < 5e-6, res
    # the fp32-operand kernel on the same tensors
    dq = torch.empty(lib.dcn_grad_blocked_bytes(M, cout) // 4, device=dev)
    assert lib.dcn_split_grad_blocked_f16(L.ptr(dd), M, cout, L.ptr(ad), L.ptr(dq), None) == 0
    slab2 = garbage(lib.dcn_conv_wgrad_workspace_f16(ctypes.byref(d)), dev, seed + 6)
    dw2 = torch.full_like(dw, float("nan"))
    assert lib.dcn_conv_wgrad_f16(ctypes.byref(d), L.ptr(xd), 1, L.ptr(ax), L.ptr(dq), L.ptr(ad), L.ptr(dw2), L.ptr(slab2), None) == 0
    res["vs_f16"] = rel_err(dw.cpu(), dw2.cpu())
    assert res["vs_f16"] < 6e-6, res
    dw3 = torch.full_like(dw, float("nan"))   # bit-reproducible
    assert lib.dcn_conv_wgrad_hl(ctypes.byref(d), L.ptr(xi), L.ptr(ax), L.ptr(di), L.ptr(ad), L.ptr(dw3), L.ptr(slab), None) == 0
    assert torch.equal(dw3.cpu(), dw.cpu())
    return res
