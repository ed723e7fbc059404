#!/usr/bin/env python3
"""Prints (and optionally stores as JSON) the measured deviation of one full-size training step of every BASELINE config
from its committed oracle fixture, in both convolution arithmetics -- the numbers behind tests/test_gpu_configs.py.

    python tools/parity_report.py [--config 1 2 3 5] [--json profiles/rN_parity_report.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-dense-correspondence_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, nargs="+", default=[1, 2, 3, 5])
    ap.add_argument("--json", default=None)
    ap.add_argument("--pair-call", action="store_true", help="forward_pair instead of two forward calls")
    a = ap.parse_args()
    import torch
    from dcn_hip import backbone as bb
    import parity_common as pc
    rows = []
    for cfg in a.config:
        if not os.path.exists(pc.fixture_path(cfg)):
            print("config %d: no fixture" % cfg)
            continue
        for mode in ("f16x3", "fp32"):
            bb.set_conv_mode(mode)
            r = pc.run_config_against_fixture(cfg, pair_call=a.pair_call)
            per = r.pop("per_tensor")
            worst = sorted(per, key=lambda t: -t[2])[:3]
            r["conv_mode"] = mode
            r["worst_tensors_l2"] = [(k, "%.2e" % e, "%.2f x" % y) for k, e, y, _, _, _ in worst]
            import math
            tot = math.sqrt(sum((e * 1.0) ** 2 for _, e, _, _, _, _ in per) / len(per))
            r["grad_l2_rel_rms_over_tensors"] = tot
            rows.append(r)
            print(json.dumps(r), flush=True)
            torch.cuda.empty_cache()
    bb.set_conv_mode(None)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
