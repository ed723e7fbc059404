"""Inference-side measurements on one MI355X:

  * eval-mode forward (running BN statistics) images/s of DenseCorrespondenceNetwork at 640x480,
  * ``find_best_matches``: Q queries against one [H, W, D] descriptor image in ONE pass over the image (each work-item keeps
    its pixel's descriptor in registers and walks the queries): algorithmic bytes = HW*D*4 (+ Q*HW*4 when the distance
    images of the reference's ``find_best_match`` are requested -- then it is a pure HBM write stream).

    python tools/infer_bench.py [--d 3] [--queries 100] [--batch 8]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-dense-correspondence_amd"))


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=3)
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    H, W = 480, 640
    cfg = {"descriptor_dimension": a.d, "image_width": W, "image_height": H,
           "backbone": {"model_class": "Resnet", "resnet_name": "Resnet34_8s"}}
    dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False)
    dcn.eval()
    out = {"device": torch.cuda.get_device_name(0)}
    x = torch.randn(a.batch, 3, H, W, device="cuda")
    with torch.no_grad():
        ms = timed(lambda: dcn.forward(x), a.reps)
    flops = dcn.fcn.forward_flops(a.batch, H, W)
    out["forward_eval"] = {"batch": a.batch, "ms": ms, "images_per_s": 1e3 * a.batch / ms, "conv_tflops": flops / ms / 1e9}
    with torch.no_grad():
        res = dcn.forward(x[:1])[0].permute(1, 2, 0).contiguous()      # [H, W, D]
    q = res.reshape(-1, a.d)[torch.randint(0, H * W, (a.queries,), device="cuda")].contiguous()
    from dcn_hip import match
    for nd in (False, True):
        ms = timed(lambda: match.find_best_matches(res, q, return_norm_diffs=nd), a.reps)
        algo = H * W * a.d * 4 + (a.queries * H * W * 4 if nd else 0)
        out["find_best_match" + ("_with_distance_images" if nd else "")] = {
            "queries": a.queries, "us": 1e3 * ms, "us_per_query": 1e3 * ms / a.queries,
            "algorithmic_GB_per_s": algo / (ms * 1e-3) / 1e9, "frac_of_8TBps": algo / (ms * 1e-3) / 8e12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
