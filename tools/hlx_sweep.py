#!/usr/bin/env python3
"""Forward / dgrad of the wide layers through the C ABI at several batch sizes, one process, one table: which kernel / tile
the library picks by default, the round-4 choice (DCN_GEMM_HLX=0), and the small-tile kernel of conv_hlx_kernels.hip forced
into every (K groups, K splits) combination -- the measurements hlx_shape's cost model is calibrated on.  GPU only.
    python tools/hlx_sweep.py [--n 1,2,4,8] [--reps 30] [--variants default,old,1:1,2:1,2:2,...] [--only layer4]
The library re-reads DCN_* between variants (dcn_reload_env)."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-dense-correspondence_amd"))
import torch  # noqa: E402
from dcn_hip import _lib  # noqa: E402

SHAPES = [  # name, cin, cout, k, dil   (60 x 80 maps: 640 x 480 images at stride 8)
    ("layer2 3x3 128->128", 128, 128, 3, 1),
    ("layer3.0 3x3 d2 128->256", 128, 256, 3, 2),
    ("layer3 3x3 d2 256->256", 256, 256, 3, 2),
    ("layer4.0 3x3 d4 256->512", 256, 512, 3, 4),
    ("layer4 3x3 d4 512->512", 512, 512, 3, 4),
    ("layer4 down 1x1 256->512", 256, 512, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--warm", type=int, default=15)
    ap.add_argument("--variants", default="default,old,1:1,1:2,1:3,2:1,2:2,2:3,2:4")
    ap.add_argument("--only", default="")
    ap.add_argument("--kinds", default="fwd,dgrad")
    ap.add_argument("--narrow", action="store_true", help="set DCN_GEMM_HLX_NARROW=1 for every variant (128-channel destinations)")
    a = ap.parse_args()
    lib = _lib.get()
    dev = torch.device("cuda")
    st = _lib.stream_ptr()
    P, I = ctypes.c_void_p, ctypes.c_int
    arr = lambda ty, v: (ty * 1)(v)
    h, w = 60, 80

    def set_env(**kw):
        for k_ in ("DCN_GEMM_HLX", "DCN_GEMM_HL_ROWS", "DCN_GEMM_HL"):
            os.environ.pop(k_, None)
        for k_, v in kw.items():
            os.environ[k_] = str(v)
        if a.narrow:
            os.environ["DCN_GEMM_HLX_NARROW"] = "1"
        lib.dcn_reload_env()

    for n in [int(v) for v in a.n.split(",")]:
        print("=== N = %d images (M = %d rows)" % (n, n * h * w), flush=True)
        for name, cin, cout, k, dil in SHAPES:
            if a.only and a.only not in name:
                continue
            pad = dil * (k - 1) // 2
            d = _lib.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, pad, dil, cout, (n // 2) * h * w if n > 1 else 0)
            M, K, Kt = n * h * w, k * k * cin, k * k * cout
            x = torch.relu(torch.randn(n, h, w, cin, device=dev))
            wgt = torch.randn(cout, k, k, cin, device=dev) * 0.05
            dy = torch.randn(n, h, w, cout, device=dev) * 1e-3
            y = torch.empty(n, h, w, cout, device=dev)
            dx = torch.empty_like(x)
            ax, ad = x.abs().max().reshape(1), dy.abs().max().reshape(1)
            x_hl, dy_hl = torch.empty(x.numel(), device=dev), torch.empty(dy.numel(), device=dev)
            w_hl, wt_hl = torch.empty(cout * K, device=dev), torch.empty(cin * Kt, device=dev)
            assert lib.dcn_split_act_hl32(_lib.ptr(x), _lib.ptr(ax), _lib.ptr(x_hl), M, cin, st) == 0
            assert lib.dcn_split_act_hl32(_lib.ptr(dy), _lib.ptr(ad), _lib.ptr(dy_hl), M, cout, st) == 0
            assert lib.dcn_split_weights_hl32(1, arr(P, wgt.data_ptr()), arr(P, w_hl.data_ptr()), arr(I, cout), arr(I, k * k), arr(I, cin), arr(I, cout), 0, 64.0, st) == 0
            assert lib.dcn_split_weights_hl32(1, arr(P, wgt.data_ptr()), arr(P, wt_hl.data_ptr()), arr(I, cout), arr(I, k * k), arr(I, cin), arr(I, cout), 1, 64.0, st) == 0
            wt = wgt.reshape(cout, k * k, cin).permute(2, 1, 0).contiguous()          # [cin][taps][cout]
            kp, kpt = lib.dcn_f16_kpad(K), lib.dcn_f16_kpad(Kt)
            wh = torch.empty(cout, kp, dtype=torch.float16, device=dev); wl = torch.empty_like(wh)
            wth = torch.empty(cin, kpt, dtype=torch.float16, device=dev); wtl = torch.empty_like(wth)
            assert lib.dcn_split_rows_f16(_lib.ptr(wgt), _lib.ptr(wh), _lib.ptr(wl), cout, K, 64.0, st) == 0
            assert lib.dcn_split_rows_f16(_lib.ptr(wt), _lib.ptr(wth), _lib.ptr(wtl), cin, Kt, 64.0, st) == 0
            part = torch.empty((M + 31) // 32 + 2, 3, cout, device=dev)
            flops = 2.0 * M * cout * K
            ref = {}
            for var in a.variants.split(","):
                if var == "default":
                    set_env()
                elif var == "old":
                    set_env(DCN_GEMM_HLX=0)
                else:
                    kg, sp = var.split(":")
                    set_env(DCN_GEMM_HLX="%s,%s" % (kg, sp), DCN_GEMM_HL=2)
                line = "  %-26s %-8s" % (name, var)
                for kind in a.kinds.split(","):
                    dg = 1 if kind == "dgrad" else 0
                    info = (ctypes.c_int * 6)()
                    el = lib.dcn_conv_hl_eligible(ctypes.byref(d), dg)
                    rc = lib.dcn_conv_hl_shape_info(ctypes.byref(d), dg, info)
                    if var not in ("default", "old") and (rc != 0 or info[0] != 160 or "%d:%d" % (info[2], info[3]) != var):
                        line += " | %-5s %-28s" % (kind, "-- (shape not available)")
                        continue
                    if el:
                        ws = torch.empty(max(lib.dcn_conv_gemm_workspace_hl(ctypes.byref(d), dg), 4) // 4, device=dev)
                        if dg:
                            fn = lambda: lib.dcn_conv_dgrad_hl(ctypes.byref(d), _lib.ptr(dy_hl), _lib.ptr(wt_hl), 64.0, _lib.ptr(ad), None, _lib.ptr(dx), _lib.ptr(ws), st)
                        else:
                            fn = lambda: lib.dcn_conv_forward_hl(ctypes.byref(d), _lib.ptr(x_hl), _lib.ptr(ax), _lib.ptr(w_hl), 64.0, None, _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws), st)
                        tag = "hl %dx%d kg%d s%d %3dwg" % (info[0], info[1], info[2], info[3], info[3] * info[4] * info[5])
                    else:
                        ws = torch.empty(max(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), dg), 4) // 4, device=dev)
                        if dg:
                            fn = lambda: lib.dcn_conv_dgrad_f16(ctypes.byref(d), _lib.ptr(dy), _lib.ptr(wth), _lib.ptr(wtl), 64.0, _lib.ptr(ad), None, _lib.ptr(dx), _lib.ptr(ws), st)
                        else:
                            fn = lambda: lib.dcn_conv_forward_f16(ctypes.byref(d), _lib.ptr(x), _lib.ptr(ax), _lib.ptr(wh), _lib.ptr(wl), 64.0, None, _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws), st)
                        tag = "f16 (fp32 operands)"
                    for _ in range(a.warm):
                        assert fn() == 0
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    us = 1e3 * e0.elapsed_time(e1) / a.reps
                    out = (dx if dg else y).clone()
                    err = ""
                    if kind in ref:   # every variant computes the same convolution
                        e = float((out - ref[kind]).abs().max() / ref[kind].abs().max())
                        err = " d%.0e" % e if e > 0 else " ="
                        if not e < 1e-5:
                            err += " MISMATCH"
                    else:
                        ref[kind] = out
                    line += " | %-5s %-24s %6.1f us %5.0f TF%s" % (kind, tag, us, flops / us / 1e6, err)
                print(line, flush=True)
    set_env()


if __name__ == "__main__":
    main()
