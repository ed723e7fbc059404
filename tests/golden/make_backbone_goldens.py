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
    torch.set_num_threads(os.cpu_count())
    c = synth.CONFIGS[1]
    model = resnet_dilated_oracle.build(c["backbone"], c["D"], seed=0)
    model.train()
    img_a, img_b, lists = synth.make_batch(c["B"], c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=1)
    loss, terms, da, db = ostep.forward_loss(model, img_a, img_b, lists, synth.LOSS_CONFIG)
    loss.backward()
    names, norms, samples = [], [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        flat = p.grad.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 8).long()
        samples.append(flat[idx].numpy())
    out = os.path.join(HERE, "config1_oracle.npz")
    np.savez_compressed(out, desc_a=da.detach()[:, :, ::16, ::16].numpy(), desc_b=db.detach()[:, :, ::16, ::16].numpy(),
                        desc_a_absmax=float(da.abs().max()), loss=float(loss),
                        terms=np.array([float(t.sum()) for t in terms[0]]), grad_names=np.array(names),
                        grad_norms=np.array(norms), grad_samples=np.stack(samples),
                        running_mean_bn1=model.resnet34_8s.bn1.running_mean.numpy())
    print(out, os.path.getsize(out), "bytes; loss", float(loss), [float(t.sum()) for t in terms[0]])


if __name__ == "__main__":
    main()
