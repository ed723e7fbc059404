#!/usr/bin/env python3
"""The contrastive-loss call (forward + backward) alone at BASELINE configs[2]'s list sizes (32 image pairs, D = 16, 10 000 +
50 000 + 50 000 pixel pairs each) or configs[1]'s, in the four combinations of SAVE_PAIR_RECORDS x PREFILL_GRADIENTS
(dcn_hip/loss.py).   python tools/loss_bench.py [--config 3] [--modes 00,10,11] [--reps 10]
Under `rocprofv3 --kernel-trace --stats` with ONE mode the kernel table shows where the call's time goes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=[2, 3])
    ap.add_argument("--modes", default="00,10,01,11", help="save-records / prefill bits per variant")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    from dcn_hip import loss as K
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    c = bench.WORKLOADS["config%d" % a.config]
    B, D, HW = c["B"], c["D"], c["H"] * c["W"]
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(3)
    lists = []
    for _ in range(B):
        t8 = []
        for n in (c["Pm"], c["Pk"], c["Pg"]):
            t8 += [torch.randint(0, HW, (n,), generator=gen, dtype=torch.int64) for _ in range(2)]
        t8 += [torch.tensor([-1], dtype=torch.int64)] * 2
        lists.append(tuple(t8))
    pl = K.PairLists.from_lists(lists, dev, hw=HW)
    gg = torch.Generator(device=dev).manual_seed(4)
    da = ((torch.rand(B, HW, D, device=dev, generator=gg) * 2 - 1) * 0.12).requires_grad_(True)
    db = ((torch.rand(B, HW, D, device=dev, generator=gg) * 2 - 1) * 0.12).requires_grad_(True)
    pcl = PixelwiseContrastiveLoss(image_shape=[c["H"], c["W"]], config=bench.LOSS_CONFIG)
    pair_bytes = (16 * D + 16) * pl.total
    fill_bytes = 2 * B * HW * D * 4

    def call():
        l = loss_composer.get_loss_batched(pcl, 0, da, db, pl)[0]
        return torch.autograd.grad(l, [da, db])
    for m in a.modes.split(","):
        K.SAVE_PAIR_RECORDS, K.PREFILL_GRADIENTS = m[0] == "1", m[1] == "1"
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print("loss call config %d  records=%s prefill=%s  %.1f us/call  pair bytes %.3f of 8 TB/s (%.2f TB/s), with the zero-fill %.3f"
              % (a.config, m[0], m[1], 1e3 * ms, pair_bytes / (ms * 1e-3) / 8e12, pair_bytes / (ms * 1e-3) / 1e12,
                 (pair_bytes + fill_bytes) / (ms * 1e-3) / 8e12), flush=True)


if __name__ == "__main__":
    main()
