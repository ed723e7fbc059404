#!/usr/bin/env python3
"""Per-shape micro-benchmark of the matrix-core kernels through the C ABI (GPU only): every distinct convolution
of Resnet34_8s at a given batch, forward / dgrad / wgrad, timed with events on the launch stream.
    python tools/conv_bench.py [--n 4] [--reps 10] [--mode f16 [--check] [--x-direct] [--no-split]] [--only NAME] [--kinds fwd,dgrad,wgrad]
Prints TFLOP/s (algorithmic; the totals also as a fraction of the fp32-MFMA peak, 157.3 TF -- the f16x3 kernels exceed it).
Use one process for a whole sweep: a fresh process measures the first layers at clocks that have not ramped up yet."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-dense-correspondence_amd"))
import torch  # noqa: E402
from dcn_hip import _lib  # noqa: E402

SHAPES = [  # name, count in the net, hin, win, cin, cout, k, stride, pad, dil
    ("stem 7x7/2 3(4)->64", 1, 480, 640, 4, 64, 7, 2, 3, 1),
    ("layer1 3x3 64->64", 6, 120, 160, 64, 64, 3, 1, 1, 1),
    ("layer2.0 3x3/2 64->128", 1, 120, 160, 64, 128, 3, 2, 1, 1),
    ("layer2 3x3 128->128", 7, 60, 80, 128, 128, 3, 1, 1, 1),
    ("layer2 down 1x1/2 64->128", 1, 120, 160, 64, 128, 1, 2, 0, 1),
    ("layer3.0 3x3 d2 128->256", 1, 60, 80, 128, 256, 3, 1, 2, 2),
    ("layer3 3x3 d2 256->256", 11, 60, 80, 256, 256, 3, 1, 2, 2),
    ("layer3 down 1x1 128->256", 1, 60, 80, 128, 256, 1, 1, 0, 1),
    ("layer4.0 3x3 d4 256->512", 1, 60, 80, 256, 512, 3, 1, 4, 4),
    ("layer4 3x3 d4 512->512", 5, 60, 80, 512, 512, 3, 1, 4, 4),
    ("layer4 down 1x1 256->512", 1, 60, 80, 256, 512, 1, 1, 0, 1),
]


SHAPES_R50 = [  # Resnet50_8s at 1280 x 960 (BASELINE config 5): the bottleneck layers 3-4 (160 x 120 maps)
    ("r50 l3 conv1 1x1 1024->256", 5, 120, 160, 1024, 256, 1, 1, 0, 1),
    ("r50 l3 conv2 3x3 d2 256->256", 6, 120, 160, 256, 256, 3, 1, 2, 2),
    ("r50 l3 conv3 1x1 256->1024", 6, 120, 160, 256, 1024, 1, 1, 0, 1),
    ("r50 l3.0 conv1 1x1 512->256", 1, 120, 160, 512, 256, 1, 1, 0, 1),
    ("r50 l3.0 down 1x1 512->1024", 1, 120, 160, 512, 1024, 1, 1, 0, 1),
    ("r50 l4 conv1 1x1 2048->512", 2, 120, 160, 2048, 512, 1, 1, 0, 1),
    ("r50 l4 conv2 3x3 d4 512->512", 3, 120, 160, 512, 512, 3, 1, 4, 4),
    ("r50 l4 conv3 1x1 512->2048", 3, 120, 160, 512, 2048, 1, 1, 0, 1),
    ("r50 l4.0 conv1 1x1 1024->512", 1, 120, 160, 1024, 512, 1, 1, 0, 1),
    ("r50 l4.0 down 1x1 1024->2048", 1, 120, 160, 1024, 2048, 1, 1, 0, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="r34", choices=["r34", "r50"], help="Resnet34_8s at 640x480 (default) or Resnet50_8s layers 3-4 at 1280x960")
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--json", default="")
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--x-direct", action="store_true", help="f16 wgrad: fp32 activation operand split on the fly")
    ap.add_argument("--no-split", action="store_true", help="f16 wgrad: time the GEMM kernel alone (planes prepared once)")
    ap.add_argument("--check", action="store_true", help="with --mode f16: compare against the fp32 kernels")
    ap.add_argument("--mode", default="fp32", choices=["fp32", "f16", "hl"],
                    help="fp32 MFMA kernels, the split-fp16 (f16x3) ones, or f16 with the pre-split (hl32) LDS-DMA kernel where eligible")
    ap.add_argument("--hl-split", action="store_true", help="hl mode: time the stand-alone operand split pass with the GEMM")
    ap.add_argument("--relu-x", action="store_true", help="activations = relu(randn) (half zeros, one sign) instead of randn")
    a = ap.parse_args()
    lib = _lib.get()
    dev = torch.device("cuda")
    st = _lib.stream_ptr()
    rows = []
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    for name, count, hin, win, cin, cout, k, stride, pad, dil in (SHAPES if a.shapes == "r34" else SHAPES_R50):
        if a.only and a.only not in name:
            continue
        n = a.n
        hout = (hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
        wout = (win + 2 * pad - dil * (k - 1) - 1) // stride + 1
        d = _lib.ConvDesc(n, hin, win, cin, hout, wout, cout, k, k, stride, pad, dil, cout)
        x = torch.randn(n, hin, win, cin, device=dev)
        if a.relu_x:
            x = torch.relu(x)
        w = torch.randn(cout, k, k, cin, device=dev) * 0.05
        wt = torch.randn(cin, k, k, cout, device=dev) * 0.05
        y = torch.empty(n, hout, wout, cout, device=dev)
        dy = torch.randn(n, hout, wout, cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        part = torch.empty(lib.dcn_conv_num_mtiles(ctypes.byref(d)), 3, cout, device=dev)
        slab = torch.empty(max(lib.dcn_conv_wgrad_workspace(ctypes.byref(d)), 4) // 4, device=dev)
        flops = 2.0 * n * hout * wout * cout * k * k * (3 if cin == 4 else cin)
        wsf = torch.empty(max(lib.dcn_conv_gemm_workspace(ctypes.byref(d), 0), 4) // 4, device=dev)
        wsd = torch.empty(max(lib.dcn_conv_gemm_workspace(ctypes.byref(d), 1), 4) // 4, device=dev)
        if a.mode in ("f16", "hl"):
            K, Kt = k * k * cin, k * k * cout
            kp, kpt = lib.dcn_f16_kpad(K), lib.dcn_f16_kpad(Kt)
            wh = torch.empty(cout, kp, dtype=torch.float16, device=dev); wl = torch.empty_like(wh)
            wth = torch.empty(cin, kpt, dtype=torch.float16, device=dev); wtl = torch.empty_like(wth)
            assert lib.dcn_split_rows_f16(_lib.ptr(w), _lib.ptr(wh), _lib.ptr(wl), cout, K, 64.0, st) == 0
            assert lib.dcn_split_rows_f16(_lib.ptr(wt), _lib.ptr(wth), _lib.ptr(wtl), cin, Kt, 64.0, st) == 0
            amax = dy.abs().max().reshape(1)
            wsf = torch.empty(max(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 0), 4) // 4, device=dev)
            wsd = torch.empty(max(lib.dcn_conv_gemm_workspace_f16(ctypes.byref(d), 1), 4) // 4, device=dev)
            # (sized for whichever kernel writes more rows of batch-norm partial sums: the hl32 kernel's 192-row tiles make
            # more M tiles than the 256-row ones of the fp32-operand kernel -- 50 against 38 at two images)
            part = torch.empty(max(lib.dcn_conv_num_mtiles_f16(ctypes.byref(d)), lib.dcn_conv_num_mtiles_hl(ctypes.byref(d)),
                                   (n * hout * wout + 31) // 32), 3, cout, device=dev)
        calls = {
            "fwd": lambda: lib.dcn_conv_forward(ctypes.byref(d), _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y), _lib.ptr(part), _lib.ptr(wsf), st),
            "dgrad": lambda: lib.dcn_conv_dgrad(ctypes.byref(d), _lib.ptr(dy), _lib.ptr(wt), None, _lib.ptr(dx), _lib.ptr(wsd), st),
            "wgrad": lambda: lib.dcn_conv_wgrad(ctypes.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(slab), st),
        }
        if a.mode in ("f16", "hl"):
            calls["fwd"] = lambda: lib.dcn_conv_forward_f16(ctypes.byref(d), _lib.ptr(x), None, _lib.ptr(wh), _lib.ptr(wl), 64.0,
                                                            None, _lib.ptr(y), _lib.ptr(part), _lib.ptr(wsf), st)
            calls["dgrad"] = lambda: lib.dcn_conv_dgrad_f16(ctypes.byref(d), _lib.ptr(dy), _lib.ptr(wth), _lib.ptr(wtl), 64.0,
                                                            _lib.ptr(amax), None, _lib.ptr(dx), _lib.ptr(wsd), st)
            slab = torch.empty(max(lib.dcn_conv_wgrad_workspace_f16(ctypes.byref(d)), 4) // 4, device=dev)
            M = n * hout * wout
            xs = torch.empty(x.numel(), device=dev)
            dq = torch.empty(lib.dcn_grad_blocked_bytes(M, cout) // 4, device=dev)

            def wgrad_f16(split=not a.no_split):   # the two split passes are part of the cost unless --no-split
                rc = 0
                if a.x_direct:   # activation operand = the fp32 tensor, split on the fly (no split pass)
                    if split:
                        rc |= lib.dcn_split_grad_blocked_f16(_lib.ptr(dy), M, cout, _lib.ptr(amax), _lib.ptr(dq), st)
                    return rc | lib.dcn_conv_wgrad_f16(ctypes.byref(d), _lib.ptr(x), 1, None, _lib.ptr(dq), _lib.ptr(amax), _lib.ptr(dw),
                                                       _lib.ptr(slab), st)
                if split:
                    rc |= lib.dcn_split_act_f16(_lib.ptr(x), _lib.ptr(xs), x.numel(), st)
                    rc |= lib.dcn_split_grad_blocked_f16(_lib.ptr(dy), M, cout, _lib.ptr(amax), _lib.ptr(dq), st)
                return rc | lib.dcn_conv_wgrad_f16(ctypes.byref(d), _lib.ptr(xs), 0, None, _lib.ptr(dq), _lib.ptr(amax), _lib.ptr(dw),
                                                   _lib.ptr(slab), st)
            assert wgrad_f16(True) == 0
            calls["wgrad"] = wgrad_f16
            hl_f = a.mode == "hl" and lib.dcn_conv_hl_eligible(ctypes.byref(d), 0)
            hl_d = a.mode == "hl" and lib.dcn_conv_hl_eligible(ctypes.byref(d), 1)
            hl_w = a.mode == "hl" and lib.dcn_conv_wgrad_hl_eligible(ctypes.byref(d))
            if hl_w:
                axw = x.abs().max().reshape(1)
                xw_hl = torch.empty(x.numel(), device=dev)
                dyw_hl = torch.empty(dy.numel(), device=dev)
                slab_hl = torch.empty(max(lib.dcn_conv_wgrad_workspace_hl(ctypes.byref(d)), 4) // 4, device=dev)
                assert lib.dcn_split_act_hl32(_lib.ptr(x), _lib.ptr(axw), _lib.ptr(xw_hl), n * hin * win, cin, st) == 0
                assert lib.dcn_split_act_hl32(_lib.ptr(dy), _lib.ptr(amax), _lib.ptr(dyw_hl), n * hout * wout, cout, st) == 0

                def wgrad_hl():
                    rc = 0
                    if a.hl_split:
                        rc |= lib.dcn_split_act_hl32(_lib.ptr(dy), _lib.ptr(amax), _lib.ptr(dyw_hl), n * hout * wout, cout, st)
                    return rc | lib.dcn_conv_wgrad_hl(ctypes.byref(d), _lib.ptr(xw_hl), _lib.ptr(axw), _lib.ptr(dyw_hl), _lib.ptr(amax),
                                                     _lib.ptr(dw), _lib.ptr(slab_hl), st)
                calls["wgrad"] = wgrad_hl
                name = name + (" [hlr wgrad]" if lib.dcn_conv_wgrad_hl_kind(ctypes.byref(d)) == 2 else " [hl wgrad]")
            if hl_f or hl_d:
                P, I = ctypes.c_void_p, ctypes.c_int
                arr = lambda ty, v: (ty * 1)(v)
                ws_hl = torch.empty(max(lib.dcn_conv_gemm_workspace_hl(ctypes.byref(d), 0), lib.dcn_conv_gemm_workspace_hl(ctypes.byref(d), 1), 4) // 4, device=dev)
                ax = x.abs().max().reshape(1)
            if hl_f:
                x_hl = torch.empty(x.numel(), device=dev)
                w_hl = torch.empty(cout * K, device=dev)
                assert lib.dcn_split_weights_hl32(1, arr(P, w.data_ptr()), arr(P, w_hl.data_ptr()), arr(I, cout), arr(I, k * k), arr(I, cin), arr(I, cout), 0, 64.0, st) == 0
                assert lib.dcn_split_act_hl32(_lib.ptr(x), _lib.ptr(ax), _lib.ptr(x_hl), n * hin * win, cin, st) == 0

                def fwd_hl():
                    rc = lib.dcn_split_act_hl32(_lib.ptr(x), _lib.ptr(ax), _lib.ptr(x_hl), n * hin * win, cin, st) if a.hl_split else 0
                    return rc | lib.dcn_conv_forward_hl(ctypes.byref(d), _lib.ptr(x_hl), _lib.ptr(ax), _lib.ptr(w_hl), 64.0, None, _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws_hl), st)
                calls["fwd"] = fwd_hl
                name = name + " [hl fwd]"
            if hl_d:
                dy_hl = torch.empty(dy.numel(), device=dev)
                wt_hl = torch.empty(cin * Kt, device=dev)
                w_nat = wt.reshape(cin, k * k, cout).permute(2, 1, 0).contiguous()   # [cout][taps][cin] whose transpose is wt
                assert lib.dcn_split_weights_hl32(1, arr(P, w_nat.data_ptr()), arr(P, wt_hl.data_ptr()), arr(I, cout), arr(I, k * k), arr(I, cin), arr(I, cout), 1, 64.0, st) == 0
                assert lib.dcn_split_act_hl32(_lib.ptr(dy), _lib.ptr(amax), _lib.ptr(dy_hl), n * hout * wout, cout, st) == 0

                def dgrad_hl():
                    rc = lib.dcn_split_act_hl32(_lib.ptr(dy), _lib.ptr(amax), _lib.ptr(dy_hl), n * hout * wout, cout, st) if a.hl_split else 0
                    return rc | lib.dcn_conv_dgrad_hl(ctypes.byref(d), _lib.ptr(dy_hl), _lib.ptr(wt_hl), 64.0, _lib.ptr(amax), None, _lib.ptr(dx), _lib.ptr(ws_hl), st)
                calls["dgrad"] = dgrad_hl
                name = name + " [hl dgrad]"
            if a.check:   # f16x3 vs the fp32 MFMA kernels on the same operands
                y2, dx2, dw2 = torch.empty_like(y), torch.empty_like(dx), torch.empty_like(dw)
                part2 = torch.empty(lib.dcn_conv_num_mtiles(ctypes.byref(d)), 3, cout, device=dev)
                w2 = torch.empty(max(lib.dcn_conv_gemm_workspace(ctypes.byref(d), 0), lib.dcn_conv_gemm_workspace(ctypes.byref(d), 1),
                                     lib.dcn_conv_wgrad_workspace(ctypes.byref(d)), 4) // 4, device=dev)
                assert lib.dcn_conv_forward(ctypes.byref(d), _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y2), _lib.ptr(part2), _lib.ptr(w2), st) == 0
                assert lib.dcn_conv_dgrad(ctypes.byref(d), _lib.ptr(dy), _lib.ptr(wt), None, _lib.ptr(dx2), _lib.ptr(w2), st) == 0
                assert lib.dcn_conv_wgrad(ctypes.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw2), _lib.ptr(w2), st) == 0
                for fn in calls.values():
                    assert fn() == 0
                torch.cuda.synchronize()
                rel = lambda p_, q_: float((p_ - q_).abs().max() / q_.abs().max())
                print("  check %-24s y %.2e  dx %.2e  dw %.2e" % (name, rel(y, y2), rel(dx, dx2), rel(dw, dw2)), flush=True)
        row = {"shape": name, "count": count, "gflop": flops / 1e9}
        for kind in ("fwd", "dgrad", "wgrad"):
            row[kind + "_us"] = float("nan")
            row[kind + "_tf"] = float("nan")
        for kind, fn in calls.items():
            if kind not in a.kinds.split(","):
                continue
            for _ in range(2):
                assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            row[kind + "_us"] = 1e3 * ms
            row[kind + "_tf"] = flops / (ms * 1e-3) / 1e12
            if not (kind == "dgrad" and cin == 4):
                tot[kind][0] += count * ms
                tot[kind][1] += count * flops
        rows.append(row)
        print("%-28s x%-2d %7.2f GF | fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" %
              (name, count, row["gflop"], row["fwd_us"], row["fwd_tf"], row["dgrad_us"], row["dgrad_tf"],
               row["wgrad_us"], row["wgrad_tf"]), flush=True)
    for kind, (ms, fl) in tot.items():
        if ms <= 0:
            continue
        print("net total %-6s %8.2f ms  %6.1f TF/s  (%.1f%% of fp32 MFMA peak)" % (kind, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573))
    if a.json:
        json.dump({"n": a.n, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
