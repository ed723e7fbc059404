#!/usr/bin/env python3
"""Host-side cost of one training step without a GPU: the host-emulation build with HIPEMU_NOEXEC=1 (every kernel launch
returns at once), so what is timed is the engine's own C++ (plan logic, tile selection, launch bookkeeping), the ctypes
bindings and the Python / autograd glue -- everything of `host_enqueue_ms_per_step` except the HIP runtime's launch cost.
TEST / ANALYSIS TOOL: results are meaningless as numerics (nothing is computed).
    HIPEMU_NOEXEC=1 python tools/host_overhead.py [--batch 1] [--separate] [--steps 20]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-dense-correspondence_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("HIPEMU_NOEXEC", "1")
from helpers import use_emulation_library  # noqa: E402
use_emulation_library()
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--separate", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--profile", action="store_true", help="cProfile of the timed steps")
    a = ap.parse_args()
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    from dcn_hip.optim import Adam
    from oracle import synth
    H, W, D, B = 480, 640, 3, a.batch
    cfg = {"descriptor_dimension": D, "image_width": W, "image_height": H}
    dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False)
    img_a, img_b, lists = synth.make_batch(B, H, W, 5000, 2500, 2500, seed=1)
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    from dcn_hip.distributed import FlatGradients
    from dcn_hip.loss import PairLists
    opt = FlatGradients(dcn).attach(Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4))   # (as bench.py: one flat gradient buffer)
    keys = ("matches_a", "matches_b", "masked_non_matches_a", "masked_non_matches_b", "background_non_matches_a",
            "background_non_matches_b", "blind_non_matches_a", "blind_non_matches_b")
    tup = PairLists.from_lists([tuple(Ld[k] for k in keys) for Ld in lists], torch.device("cpu"), hw=H * W)

    def step():
        opt.zero_grad()
        if a.separate:
            ya, yb = dcn.forward(img_a), dcn.forward(img_b)
        else:
            ya, yb = dcn.forward_pair(img_a, img_b)
        pa, pb = dcn.process_network_output(ya, B), dcn.process_network_output(yb, B)
        loss = loss_composer.get_loss_batched(pcl, 0, pa, pb, tup)[0]
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    if a.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(a.steps):
            step()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
        return
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    print("host side of one step (B = %d, %s): %.2f ms  (launches return at once: engine C++ + bindings + Python / autograd glue)" %
          (B, "two forward calls" if a.separate else "forward_pair", ms))


if __name__ == "__main__":
    main()
