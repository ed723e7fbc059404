#!/usr/bin/env python3
"""Golden fixture for BASELINE config 1 (B=1, 640x480, D=3, Resnet34_8s, 1000/500/500 pairs) from the ORACLE
(oracle/: CPU fp32 restatement; the backbone itself is 'parity unpinned', see oracle/resnet_dilated_oracle.py).
Stores a 1/16-subsampled descriptor map of both images, the loss terms and per-parameter gradient norms plus a
few sampled gradient entries, so the GPU test can check a full-size step without /root/reference or a long CPU run.

    python tests/golden/make_backbone_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import resnet_dilated_oracle, step as ostep, synth  # noqa: E402


def main():
    import copy
    torch.set_num_threads(min(os.cpu_count(), 16))
    c = synth.CONFIGS[1]
    model = resnet_dilated_oracle.build(c["backbone"], c["D"], seed=0)
    model.train()
    model64 = copy.deepcopy(model).double()
    img_a, img_b, lists = synth.make_batch(c["B"], c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=1)
    loss, terms, da, db = ostep.forward_loss(model, img_a, img_b, lists, synth.LOSS_CONFIG)
    loss.backward()
    # float64 run of the same oracle: the yard-stick for gradient tolerances (gradients through 36 ReLU/BN layers are
    # ill-conditioned: the float32 oracle itself is only good to a few 1e-2 of max|g| on some tensors)
    loss64, _, da64, _ = ostep.forward_loss(model64, img_a.double(), img_b.double(), lists, synth.LOSS_CONFIG)
    loss64.backward()
    names, norms64, samples64, err32_max, err32_l2, gmax64 = [], [], [], [], [], []
    for (k, p), (_, p6) in zip(model.named_parameters(), model64.named_parameters()):
        names.append(k)
        g6 = p6.grad
        norms64.append(float(g6.norm()))
        gmax64.append(float(g6.abs().max()))
        flat = g6.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 16).long()
        samples64.append(flat[idx].numpy())
        err32_max.append(float((p.grad.double() - g6).abs().max()))
        err32_l2.append(float((p.grad.double() - g6).norm()))
    out = os.path.join(HERE, "config1_oracle.npz")
    np.savez_compressed(out, desc_a=da.detach()[:, :, ::16, ::16].numpy(), desc_b=db.detach()[:, :, ::16, ::16].numpy(),
                        desc_a_absmax=float(da.abs().max()), loss=float(loss.detach()),
                        desc_err32_vs_64=float((da.detach().double() - da64.detach()).abs().max() / da64.abs().max()),
                        terms=np.array([float(t.detach().sum()) for t in terms[0]]), grad_names=np.array(names),
                        grad_norms64=np.array(norms64), grad_max64=np.array(gmax64), grad_samples64=np.stack(samples64),
                        grad_err32_max=np.array(err32_max), grad_err32_l2=np.array(err32_l2),
                        running_mean_bn1=model.resnet34_8s.bn1.running_mean.numpy())
    worst = max(e / m for e, m, n in zip(err32_max, gmax64, names) if not n.endswith("fc.bias"))
    print(out, os.path.getsize(out), "bytes; loss", float(loss.detach()), "float32-oracle gradient error vs float64:",
          "worst max-rel %.2e" % worst)


if __name__ == "__main__":
    main()
