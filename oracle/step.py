"""Oracle (TEST INFRASTRUCTURE, not product): one reference training step on CPU, fp32.

Follows dense_correspondence/training/training.py:325-346:
    optimizer.zero_grad()
    image_a_pred = dcn.forward(img_a);  image_a_pred = dcn.process_network_output(image_a_pred, B)
    image_b_pred = dcn.forward(img_b);  image_b_pred = dcn.process_network_output(image_b_pred, B)
    loss, ... = loss_composer.get_loss(pcl, match_type, preds, 8 index tensors)
    loss.backward();  optimizer.step()
with ``process_network_output`` = ``view(N, D, W*H).permute(0, 2, 1)`` (network.py:303-319).

Batched semantics (SURVEY.md section 8a note B; the reference itself is batch-1 only): every one
of the B pairs is treated exactly as one reference iteration (own lists, own hard-negative
counts); ``loss = mean_b loss_b``.  BN statistics are over the B images of one forward call.
"""
import time

import torch

from . import loss_oracle
from .loss_oracle import PixelwiseContrastiveLoss, SpartanDatasetDataType


def process_network_output(image_pred, N):
    D = image_pred.shape[1]
    return image_pred.reshape(N, D, -1).permute(0, 2, 1)


def forward_loss(model, img_a, img_b, lists, loss_config, match_type=SpartanDatasetDataType.SINGLE_OBJECT_WITHIN_SCENE):
    """Returns (loss, per-pair term list, desc_a, desc_b).  ``lists``: one dict per pair (synth.make_index_lists)."""
    B, _, H, W = img_a.shape
    pcl = PixelwiseContrastiveLoss(image_shape=[H, W], config=loss_config)
    desc_a = model(img_a)
    desc_b = model(img_b)
    pa = process_network_output(desc_a, B)
    pb = process_network_output(desc_b, B)
    total = 0.0
    terms = []
    for b in range(B):
        L = lists[b]
        out = loss_oracle.get_loss(pcl, torch.tensor(match_type), pa[b:b + 1], pb[b:b + 1],
                                   L["matches_a"], L["matches_b"],
                                   L["masked_non_matches_a"], L["masked_non_matches_b"],
                                   L["background_non_matches_a"], L["background_non_matches_b"],
                                   L["blind_non_matches_a"], L["blind_non_matches_b"])
        total = total + out[0]
        terms.append(out)
    loss = total / B
    return loss, terms, desc_a, desc_b


def train_step(model, optimizer, img_a, img_b, lists, loss_config):
    optimizer.zero_grad()
    loss, terms, desc_a, desc_b = forward_loss(model, img_a, img_b, lists, loss_config)
    loss.backward()
    optimizer.step()
    return loss.detach(), terms, desc_a.detach(), desc_b.detach()


def time_cpu_step(model, img_a, img_b, lists, loss_config, steps=3, warmup=1, lr=1e-4, weight_decay=1e-4):
    """CPU baseline timer used by bench.py's ``cpu_baseline`` leg.  Returns seconds per step (median)."""
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
    model.train()
    ts = []
    budget_s = 30.0  # bounded sample: stop adding steps once ~30 s of CPU work have been spent
    spent = 0.0
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        train_step(model, opt, img_a, img_b, lists, loss_config)
        t1 = time.perf_counter()
        spent += t1 - t0
        if i >= warmup or spent > budget_s:
            ts.append(t1 - t0)
        if spent > budget_s:
            break
    ts.sort()
    return ts[len(ts) // 2]
