#!/usr/bin/env python3
"""Golden fixtures for the BASELINE configs at FULL size from the ORACLE (oracle/: CPU fp32 restatement; the backbone
itself is 'parity unpinned', see oracle/resnet_dilated_oracle.py):

    config 1: B=1  640x480   D=3  Resnet34_8s  1000/500/500 pairs      -> config1_oracle.npz
    config 2: B=4  640x480   D=3  Resnet34_8s  5000/2500/2500          -> config2_oracle.npz   (the headline workload)
    config 3: B=32 640x480   D=16 Resnet34_8s  10000/50000/50000       -> config3_oracle.npz
    config 4: B=8  640x480   D=3  Resnet34_8s  5000/2500/2500          -> config4_oracle.npz   (per-GPU share of the B=64 / 8-GPU config)
    config 5: B=2  1280x960  D=32 Resnet50_8s  masked / background     -> config5_oracle.npz   (per-GPU share of B=16)

Each stores a subsampled descriptor map of both image batches, the five loss terms and the hard-negative counts of every
pair, and -- from a float64 run of the same oracle -- per-parameter gradient norms, 16 sampled entries per tensor and the
float32 oracle's own deviation from its float64 self (the yard-stick of the gradient tolerances), so that the GPU tests
check a full-size step without /root/reference and without a long CPU run.

    python tests/golden/make_backbone_goldens.py [--config 1 2 3 4 5]

Configs 3, 4 and 5 do not fit this container's memory with every activation kept (64 images x ~1 GB): the oracle is run
with per-block activation checkpointing (torch.utils.checkpoint, exact same arithmetic, every block's forward is
recomputed during backward; BN running statistics are then updated twice and are not stored for those configs).
Wall time on 8 cores: config 1 ~20 s, 2 ~1 min, 3 ~9 min, 5 ~10 min.
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import loss_oracle, resnet_dilated_oracle, step as ostep, synth  # noqa: E402
from parity_common import probe_vectors  # noqa: E402  (tests/parity_common.py: the +-1 probe vectors shared with the GPU tests)

SUBSAMPLE = {1: 16, 2: 16, 3: 32, 4: 16, 5: 32}     # descriptor-map stride kept in the fixture
CHECKPOINT = {1: False, 2: False, 3: True, 4: True, 5: True}


def hard_negative_counts(da, db, lists, cfg):
    """Per pair: (#masked hard negatives, #background hard negatives, #pairs within 1e-5 of a margin = tie band)."""
    B = da.shape[0]
    pa, pb = ostep.process_network_output(da, B), ostep.process_network_output(db, B)
    PCL = loss_oracle.PixelwiseContrastiveLoss
    out = np.zeros((B, 3), np.int64)
    for b in range(B):
        L = lists[b]
        for j, (name, M) in enumerate((("masked_non_matches", cfg["M_masked"]), ("background_non_matches", cfg["M_background"]))):
            ia, ib = L[name + "_a"], L[name + "_b"]
            if loss_oracle.is_empty(ia):
                continue
            _, hn, xa, xb = PCL.non_match_descriptor_loss(pa[b:b + 1], pb[b:b + 1], ia, ib, M=M)
            out[b, j] = int(hn)
            d = (xa - xb).norm(2, 1)
            out[b, 2] += int(((d - M).abs() < 1e-5).sum())
    return out


def make(config):
    c = synth.CONFIGS[config]
    model = resnet_dilated_oracle.build(c["backbone"], c["D"], seed=0)
    model.train()
    resnet_dilated_oracle.set_checkpointing(model, CHECKPOINT[config])
    model64 = copy.deepcopy(model).double()
    img_a, img_b, lists = synth.make_batch(c["B"], c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=1,
                                           masked=c.get("masked", False))
    t0 = time.time()
    loss, terms, da, db = ostep.forward_loss(model, img_a, img_b, lists, synth.LOSS_CONFIG)
    loss.backward()
    with torch.no_grad():
        hard = hard_negative_counts(da.detach(), db.detach(), lists, synth.LOSS_CONFIG)
    s = SUBSAMPLE[config]
    rec = dict(desc_a=da.detach()[:, :, ::s, ::s].numpy().copy(), desc_b=db.detach()[:, :, ::s, ::s].numpy().copy(),
               desc_a_absmax=float(da.abs().max()), desc_stride=s, loss=float(loss.detach()),
               terms=np.array([[float(t.detach().sum()) for t in tb] for tb in terms]), hard=hard)
    if config == 1:
        rec["terms"] = rec["terms"][0]          # (kept as round 1 stored it)
    if not CHECKPOINT[config]:
        rec["running_mean_bn1"] = getattr(model, model._attr).bn1.running_mean.numpy().copy()
    da32 = da.detach()
    grads32 = [p.grad.detach().clone() for p in model.parameters()]
    del loss, terms, da, db
    t1 = time.time()
    print("config %d: float32 oracle step %.0f s" % (config, t1 - t0), flush=True)
    # float64 run of the same oracle: the yard-stick for gradient tolerances (gradients through 36 ReLU/BN layers are
    # ill-conditioned: the float32 oracle itself is only good to a few 1e-2 of max|g| on some tensors)
    loss64, _, da64, db64 = ostep.forward_loss(model64, img_a.double(), img_b.double(), lists, synth.LOSS_CONFIG)
    loss64.backward()
    rec["desc_err32_vs_64"] = float((da32.double() - da64.detach()).abs().max() / da64.detach().abs().max())
    # the float64 oracle's maps ("truth" for the 1e-4 bound), stored as their float32 difference from the float32 maps
    rec["desc_a64_minus_32"] = (da64.detach()[:, :, ::s, ::s].numpy() - rec["desc_a"].astype(np.float64)).astype(np.float32)
    rec["desc_b64_minus_32"] = (db64.detach()[:, :, ::s, ::s].numpy() - rec["desc_b"].astype(np.float64)).astype(np.float32)
    rec["desc64_absmax"] = float(np.abs(rec["desc_a"].astype(np.float64) + rec["desc_a64_minus_32"]).max())
    rec["loss64"] = float(loss64.detach())
    print("config %d: float64 oracle step %.0f s" % (config, time.time() - t1), flush=True)
    names, norms64, samples64, err32_max, err32_l2, gmax64, probes64 = [], [], [], [], [], [], []
    for i, ((k, p6), g32) in enumerate(zip(model64.named_parameters(), grads32)):
        names.append(k)
        g6 = p6.grad
        # random-sign probes <g, r_j>: for ANY other gradient g' of this tensor, E_r (<g' - g, r>)^2 = ||g' - g||_2^2, so four
        # numbers per tensor give an unbiased estimate of the L2 error of the WHOLE tensor (tests/parity_common.py)
        probes64.append([float((g6.reshape(-1) * r).sum()) for r in probe_vectors(i, g6.numel())])
        norms64.append(float(g6.norm()))
        gmax64.append(float(g6.abs().max()))
        flat = g6.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 16).long()
        samples64.append(flat[idx].numpy())
        err32_max.append(float((g32.double() - g6).abs().max()))
        err32_l2.append(float((g32.double() - g6).norm()))
    rec.update(grad_names=np.array(names), grad_norms64=np.array(norms64), grad_max64=np.array(gmax64),
               grad_probes64=np.array(probes64),
               grad_samples64=np.stack(samples64), grad_err32_max=np.array(err32_max), grad_err32_l2=np.array(err32_l2))
    out = os.path.join(HERE, "config%d_oracle.npz" % config)
    np.savez_compressed(out, **rec)
    worst = max(e / m for e, m, n in zip(err32_max, gmax64, names) if not n.endswith("fc.bias"))
    worst_l2 = max(e / m for e, m, n in zip(err32_l2, norms64, names) if not n.endswith("fc.bias"))
    print(out, os.path.getsize(out), "bytes; loss", rec["loss"], "desc err32 vs 64 %.2e;" % rec["desc_err32_vs_64"],
          "float32-oracle gradient error vs float64: worst max-rel %.2e, worst L2-rel %.2e" % (worst, worst_l2), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, nargs="+", default=[1, 2, 3, 4, 5], choices=[1, 2, 3, 4, 5])
    args = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count(), 16))
    for c in args.config:
        make(c)


if __name__ == "__main__":
    main()
