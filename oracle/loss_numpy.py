"""Oracle (TEST INFRASTRUCTURE, not product): independent float64 numpy statement of the within-scene
contrastive loss and its gradient -- a second opinion that shares no code with ``loss_oracle``.

Formulae (reference: pixelwise_contrastive_loss.py:132-213, 271-352; loss_composer.py:70-143):

  match     = 1/P_m * sum_i ||A[a_i] - B[b_i]||^2                                     (pcl.py:165)
  l_j       = max(0, M - ||A[a_j] - B[b_j]||_2)^2     (invert: max(0, ||.|| - M)^2)   (pcl.py:203-208)
  hard_neg  = #{ j : l_j != 0 }                                                      (pcl.py:210-211)
  w_j       = min(||uv(b_match(j)) - uv(b_j)||_2, M_pixel) / M_pixel   (optional)     (pcl.py:307-352)
  S         = sum_j l_j [* w_j]
  loss      = w_m * match + w_nm * (S_masked + S_background) / max(h_masked + h_background, 1)
              (or / (max(P_k,1) + max(P_g,1)) when scale_by_hard_negatives is False)  (loss_composer.py:107-134)
"""
import numpy as np


def _pixel_weights(matches_b, non_b, width, m_pixel):
    per = len(non_b) // len(matches_b)
    gt = np.repeat(np.asarray(matches_b, np.int64), per)
    nb = np.asarray(non_b, np.int64)[: len(gt)]
    du = (gt % width - nb % width).astype(np.float64)
    dv = (gt // width - nb // width).astype(np.float64)
    return np.minimum(np.sqrt(du * du + dv * dv), m_pixel) / m_pixel


def non_match_terms(A, B, ia, ib, margin, invert=False, weights=None):
    """A, B: [HW, D] float64.  Returns (sum, hard_neg_count, gradA, gradB) with grads of ``sum``."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    gA = np.zeros_like(A)
    gB = np.zeros_like(B)
    if len(ia) == 0:
        return 0.0, 0, gA, gB
    diff = A[ia] - B[ib]
    dist = np.sqrt((diff * diff).sum(1))
    h = (dist - margin) if invert else (margin - dist)
    h = np.maximum(h, 0.0)
    l = h * h
    hard = int(np.count_nonzero(l))
    w = np.ones_like(l) if weights is None else weights
    total = float((l * w).sum())
    # d l / d diff = 2 h * (-/+ diff/dist); d||x||/dx at x = 0 is 0 (torch convention)
    safe = np.where(dist > 0, dist, 1.0)
    coef = np.where(dist > 0, 2.0 * h * w / safe, 0.0) * (1.0 if invert else -1.0)
    g = coef[:, None] * diff
    np.add.at(gA, ia, g)
    np.add.at(gB, ib, -g)
    return total, hard, gA, gB


def within_scene(A, B, lists, cfg, width):
    """One image pair.  ``lists`` = dict(matches_a, matches_b, masked_a, masked_b, background_a,
    background_b[, blind_a, blind_b]); empty list == the reference's ``[-1]`` sentinel or length 0.
    Returns dict with loss terms, counts and gradients w.r.t. A and B ([HW, D] float64)."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)

    def clean(x):
        x = np.asarray(x, np.int64)
        return x[:0] if (len(x) == 1 and x[0] == -1) else x

    ma, mb = clean(lists["matches_a"]), clean(lists["matches_b"])
    ka, kb = clean(lists["masked_a"]), clean(lists["masked_b"])
    ga, gb = clean(lists["background_a"]), clean(lists["background_b"])
    Pm = max(len(ma), 1)
    d = A[ma] - B[mb]
    match = float((d * d).sum()) / Pm
    gmA = np.zeros_like(A)
    gmB = np.zeros_like(B)
    np.add.at(gmA, ma, 2.0 * d / Pm)
    np.add.at(gmB, mb, -2.0 * d / Pm)

    wk = _pixel_weights(mb, kb, width, cfg["M_pixel"]) if cfg.get("use_l2_pixel_loss_on_masked_non_matches") else None
    wg = _pixel_weights(mb, gb, width, cfg["M_pixel"]) if cfg.get("use_l2_pixel_loss_on_background_non_matches") else None
    Sk, hk, gkA, gkB = non_match_terms(A, B, ka, kb, cfg["M_masked"], weights=wk)
    Sg, hg, ggA, ggB = non_match_terms(A, B, ga, gb, cfg["M_background"], weights=wg)
    if cfg["scale_by_hard_negatives"]:
        scale = max(hk + hg, 1)
        masked_scaled = Sk / max(hk, 1)
        background_scaled = Sg / max(hg, 1)
    else:
        scale = max(len(ka), 1) + max(len(ga), 1)
        masked_scaled = Sk / max(len(ka), 1)
        background_scaled = Sg / max(len(ga), 1)
    wm, wnm = cfg["match_loss_weight"], cfg["non_match_loss_weight"]
    loss = wm * match + wnm * (Sk + Sg) / scale
    gA = wm * gmA + wnm * (gkA + ggA) / scale
    gB = wm * gmB + wnm * (gkB + ggB) / scale
    return dict(loss=loss, match_loss=match, masked=masked_scaled, background=background_scaled,
                S_masked=Sk, S_background=Sg, h_masked=hk, h_background=hg, gradA=gA, gradB=gB)
