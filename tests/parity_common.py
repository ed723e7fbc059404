"""Shared by tests/test_gpu_configs.py and tools/parity_report.py (not a test module): one full-size training step of a
BASELINE config on the MI355X against its committed oracle fixture (tests/golden/config<N>_oracle.npz, made by
tests/golden/make_backbone_goldens.py).  Returns the measured deviations; the caller asserts / prints."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("matches_a", "matches_b", "masked_non_matches_a", "masked_non_matches_b", "background_non_matches_a",
        "background_non_matches_b", "blind_non_matches_a", "blind_non_matches_b")


NUM_PROBES = 32


def probe_vectors(tensor_index, numel):
    """NUM_PROBES reproducible +-1 vectors (float64) for gradient tensor `tensor_index` (CPU mt19937: platform independent).
    For any two gradients g, g' of the tensor, E_r (<g' - g, r>)^2 = ||g' - g||_2^2: the fixtures store <g64, r_j>, the test
    computes <g_gpu, r_j>, and the root mean square of the differences estimates the L2 error of the WHOLE tensor."""
    g = torch.Generator().manual_seed(7919 + tensor_index)
    return [(torch.randint(0, 2, (numel,), generator=g, dtype=torch.int8).double() * 2 - 1) for _ in range(NUM_PROBES)]


def grad_error_stats(named_gpu, o32_params, o64_params, skip_suffix=("fc.bias",)):
    """Relative L2 error of every gradient tensor against the float64 oracle, for the GPU and for the float32 oracle.
    Two float32 implementations are two independent draws of the same ill-conditioning noise (ReLU kinks, hard-negative
    boundaries): tensor by tensor their ratio is anything between 0.2 and 5 (wherever the oracle happens to be lucky), so the
    meaningful statements are about the DISTRIBUTION: the root mean square over the tensors and the worst tensor."""
    import math
    e_gpu, e_o32, names = [], [], []
    for (k, p), po, p6 in zip(named_gpu, o32_params, o64_params):
        if k.endswith(tuple(skip_suffix)):
            continue   # fc.bias: mathematically zero gradient (the loss only sees descriptor differences): round-off only
        n6 = float(p6.grad.norm().clamp_min(1e-300))
        e_gpu.append(float((p.grad.detach().double().cpu() - p6.grad).norm()) / n6)
        e_o32.append(float((po.grad.double() - p6.grad).norm()) / n6)
        names.append(k)
    rms = lambda v: math.sqrt(sum(x * x for x in v) / len(v))
    i = max(range(len(names)), key=lambda j: e_gpu[j])
    return {"rms_gpu": rms(e_gpu), "rms_o32": rms(e_o32), "max_gpu": max(e_gpu), "max_o32": max(e_o32),
            "worst": (names[i], e_gpu[i], e_o32[i])}


def assert_as_accurate_as_float32(st, factor=1.5, floor=2e-4):
    """GPU gradients vs float64 are no worse than `factor` x what the float32 CPU oracle manages, in the r.m.s. over the
    tensors and on the worst tensor."""
    assert st["rms_gpu"] <= factor * st["rms_o32"] + floor, st
    assert st["max_gpu"] <= factor * st["max_o32"] + floor, st


def fixture_path(config):
    return os.path.join(GOLDEN_DIR, "config%d_oracle.npz" % config)


def build_dcn(arch, D, H, W):
    """DenseCorrespondenceNetwork on the GPU carrying the oracle's seeded weights (same state-dict keys)."""
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    from oracle import resnet_dilated_oracle
    cfg = {"descriptor_dimension": D, "image_width": W, "image_height": H,
           "backbone": {"model_class": "Resnet", "resnet_name": arch}}
    dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False)
    o = resnet_dilated_oracle.build(arch, D, seed=0)
    dcn.fcn.load_state_dict(o.state_dict())
    return dcn, o


def run_config_against_fixture(config, pair_call=False):
    """-> dict of measured deviations of one full-size step of BASELINE config `config` from its oracle fixture."""
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import synth
    z = np.load(fixture_path(config))
    c = synth.CONFIGS[config]
    B = c["B"]
    dcn, _ = build_dcn(c["backbone"], c["D"], c["H"], c["W"])
    img_a, img_b, lists = synth.make_batch(B, c["H"], c["W"], c["Pm"], c["Pk"], c["Pg"], seed=1, masked=c.get("masked", False))
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    img_a, img_b = img_a.cuda(), img_b.cuda()
    if pair_call:
        ya, yb = dcn.forward_pair(img_a, img_b)
    else:
        ya, yb = dcn.forward(img_a), dcn.forward(img_b)
    pa, pb = dcn.process_network_output(ya, B), dcn.process_network_output(yb, B)
    assert pa.is_contiguous()
    tup = [tuple(Ld[k].cuda() for k in KEYS) for Ld in lists]
    loss, terms, hard = loss_composer.get_loss_batched(pcl, 0, pa, pb, tup)
    loss.backward()
    torch.cuda.synchronize()
    s = int(z["desc_stride"])
    scale = float(z["desc_a_absmax"])
    out = {"config": config, "desc_err32_vs_64": float(z["desc_err32_vs_64"])}
    out["desc_a"] = float((ya.detach().cpu()[:, :, ::s, ::s] - torch.tensor(z["desc_a"])).abs().max()) / scale
    out["desc_b"] = float((yb.detach().cpu()[:, :, ::s, ::s] - torch.tensor(z["desc_b"])).abs().max()) / scale
    # against the FLOAT64 oracle ("truth"): what the 1e-4 bound is really about (at config 5 the float32 CPU path itself is
    # 7.8e-5 away from it, so two float32 implementations may differ by more than 1e-4 while both are within 1e-4 of truth)
    s64 = float(z["desc64_absmax"])
    a64 = torch.tensor(z["desc_a"].astype(np.float64) + z["desc_a64_minus_32"])
    b64 = torch.tensor(z["desc_b"].astype(np.float64) + z["desc_b64_minus_32"])
    out["desc_a_vs_f64"] = float((ya.detach().cpu()[:, :, ::s, ::s].double() - a64).abs().max()) / s64
    out["desc_b_vs_f64"] = float((yb.detach().cpu()[:, :, ::s, ::s].double() - b64).abs().max()) / s64
    out["loss_vs_f64"] = abs(loss.item() - float(z["loss64"])) / abs(float(z["loss64"]))
    out["loss"] = abs(loss.item() - float(z["loss"])) / abs(float(z["loss"]))
    zt = z["terms"].reshape(B, 5)
    t = terms.cpu().numpy().astype(np.float64)
    out["terms"] = float(np.max(np.abs(t - zt) / np.maximum(np.abs(zt), 1e-12)))
    h = hard.cpu().numpy()
    zh = z["hard"]
    out["hard_diff"] = int(np.max(np.abs(h[:, 1:3].astype(np.int64) - zh[:, :2])))
    out["hard_tie_band"] = int(zh[:, 2].max())
    out["hard_match_len_ok"] = bool((h[:, 0] == c["Pm"]).all())
    # gradients against the float64 run of the oracle, in units of the float32 oracle's own deviation from it
    names = [str(n) for n in z["grad_names"]]
    params = dict(dcn.fcn.named_parameters())
    worst_l2, worst_max, per = 0.0, 0.0, []
    for i, k in enumerate(names):
        if k.endswith("fc.bias"):
            continue   # mathematically zero (the loss only sees descriptor differences): round-off only
        gq = params[k].grad.detach().cpu().double()
        n64, m64 = float(z["grad_norms64"][i]), float(z["grad_max64"][i])
        e_nrm = abs(float(gq.norm()) - n64)
        flat = gq.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 16).long()
        e_smp = float((flat[idx] - torch.tensor(z["grad_samples64"][i])).abs().max())
        # unbiased estimate of ||g_gpu - g64||_2 from the random-sign probes
        pr = torch.tensor([float((flat * r).sum()) for r in probe_vectors(i, flat.numel())], dtype=torch.float64)
        e_l2 = float(((pr - torch.tensor(z["grad_probes64"][i])) ** 2).mean().sqrt())
        err32_l2, err32_mx = float(z["grad_err32_l2"][i]), float(z["grad_err32_max"][i])
        per.append((k, e_l2 / n64, e_l2 / (err32_l2 + 1e-30), e_smp / m64, e_smp / (err32_mx + 1e-30), e_nrm / n64))
        # excess over a 2e-4 relative floor (tensors the float32 oracle happens to get almost exactly), in yard-sticks
        worst_l2 = max(worst_l2, (e_l2 - 2e-4 * n64) / (err32_l2 + 1e-30))
        worst_max = max(worst_max, (e_smp - 2e-4 * m64) / (err32_mx + 1e-30))
    out["grad_l2_ratio"] = worst_l2        # diagnostic only (per-tensor ratios of two independent noise draws): see below
    out["grad_sample_ratio"] = worst_max
    out["per_tensor"] = per   # (name, est. L2 rel err, in yard-sticks, sampled max rel err, in yard-sticks, |norm diff| rel)
    # what is asserted: the distribution over the tensors -- r.m.s. and worst tensor of the relative L2 error against the
    # float64 oracle, GPU vs the float32 oracle's own (grad_error_stats explains why not tensor by tensor)
    keep = [i for i, k in enumerate(names) if not k.endswith("fc.bias")]
    o32_rel = [float(z["grad_err32_l2"][i]) / float(z["grad_norms64"][i]) for i in keep]
    o32_smp = [float(z["grad_err32_max"][i]) / float(z["grad_max64"][i]) for i in keep]
    gpu_rel = [t[1] for t in per]
    gpu_smp = [t[3] for t in per]
    rms = lambda v: float(np.sqrt(np.mean(np.square(v))))
    out["grad_rms_gpu"], out["grad_rms_o32"] = rms(gpu_rel), rms(o32_rel)
    out["grad_max_gpu"], out["grad_max_o32"] = max(gpu_rel), max(o32_rel)
    out["grad_sample_max_gpu"], out["grad_sample_max_o32"] = max(gpu_smp), max(o32_smp)   # (16 samples vs the whole tensor)
    # per-pair loss terms: a non-match pair within round-off of its margin may flip its hard-negative status; its own loss
    # term is ~0 but the normaliser 1 / #hard-negatives of that list moves by 1 / h  (config 3: h ~ 130 of 50 000)
    tie = zh[:, 2:3].astype(np.float64) + np.abs(h[:, 1:3].astype(np.float64) - zh[:, :2])
    hmin = np.maximum(np.minimum(h[:, 1:3], zh[:, :2]).astype(np.float64), 1.0)
    slack = np.zeros_like(zt)
    slack[:, 2:4] = tie / hmin
    slack[:, 0] = (tie / hmin).max(axis=1)
    out["terms_excess"] = float(np.max(np.abs(t - zt) / np.maximum(np.abs(zt), 1e-12) - slack))
    if "running_mean_bn1" in z.files:
        attr = getattr(dcn.fcn, dcn.fcn.attr)
        rm = attr.bn1.running_mean.cpu()
        out["running_mean_bn1"] = float((rm - torch.tensor(z["running_mean_bn1"])).abs().max() /
                                        np.abs(z["running_mean_bn1"]).max())
    return out


_ORACLE_TRAJECTORIES = {}   # the two CPU oracle runs depend on the problem only, not on the product variant: made once per process


def run_trajectory(dcn, oracle, B, H, W, steps, device, pairs=(400, 200, 200), decay_every=5, decay=0.9, lr=1.0e-4,
                   separate_forwards=True, fc_scale=1.0, seed=1):
    """`steps` iterations of the reference's training loop (training.py:325-346: zero_grad, forward x2, loss, backward,
    Adam step, with adjust_learning_rate of :544-558 -- lr *= decay whenever iteration % decay_every == 0) on one fixed
    synthetic batch, from identical weights, by THREE runners: the product (`dcn` on `device`, dcn_hip.optim.Adam), the
    oracle in float32 and the oracle in float64 (CPU, torch.optim.Adam).  Returns a dict:
      loss_p / loss_o32 / loss_o64   the three loss sequences
      dev_p / dev_o32                 per step |loss - loss_o64| / loss_o64 of the product and of the float32 oracle
      desc_p / desc_o32               relative deviation of the eval-mode descriptor map after the last step from the float64
                                      oracle's, up to a constant offset per channel (see below)
      slack                           per step: hard-negative tie slack product vs float32 oracle (see below)
    Why a yard-stick and not a fixed bound: Adam's update is lr * m / (sqrt(v) + eps) -- in the first iterations lr * sign(g)
    for EVERY weight, however small its gradient -- so weights whose gradient is round-off-dominated take full-size steps in
    round-off-determined directions, and every ReLU / hinge / hard-negative count is a discontinuity of the gradient.  The
    float32 oracle's trajectory leaves the float64 one's by 1e-4 after ONE step and by 1e-2 within ten (Resnet34_8s, 96 x 128,
    measured), whichever float32 implementation runs it: what can be asserted is that the product stays as close to the float64
    trajectory as the float32 reference implementation does.
    Tie slack: the loss divides the non-match sums by the number of hard negatives (loss_composer.py:107-119), a count of
    pairs whose hinge is non-zero -- a pair within round-off of its margin is counted by one float32 implementation and not by
    the other (the tie band of SURVEY.md 8c), which moves the loss by (count difference) / count.
    Descriptor offset: the scoring layer's bias has an identically zero gradient in exact arithmetic (loss and matching only see
    descriptor DIFFERENCES), so what Adam normalises into a step of +-lr per iteration is the sign of round-off -- in the oracle
    as much as in the product; a common offset of the map is no observable of the method."""
    import copy
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from dcn_hip.loss import PairLists
    from dcn_hip.optim import Adam
    from oracle import step as ostep, synth
    img_a, img_b, lists = synth.make_batch(B, H, W, pairs[0], pairs[1], pairs[2], seed=seed)
    if fc_scale != 1.0:
        with torch.no_grad():   # (tests on tiny networks scale the scoring layer so that both sides of the hinge occur)
            for net in (dcn.fcn, oracle):
                trunk = getattr(net, [n for n, _ in net.named_children()][0])
                trunk.get_parameter("fc.weight").mul_(fc_scale)
    # ---- the two oracle trajectories (cached: identical for every product variant of the same problem)
    key = (tuple(sorted((k, tuple(v.shape), float(v.double().sum())) for k, v in oracle.state_dict().items()))[:6], B, H, W, steps,
           pairs, decay_every, decay, lr, fc_scale, seed)
    if key not in _ORACLE_TRAJECTORIES:
        o32, o64 = copy.deepcopy(oracle), copy.deepcopy(oracle).double()
        o32.train(); o64.train()
        opt_o = torch.optim.Adam(o32.parameters(), lr=lr, weight_decay=1e-4)
        opt_6 = torch.optim.Adam(o64.parameters(), lr=lr, weight_decay=1e-4)
        lo, l6, h_o = [], [], []
        for it in range(1, steps + 1):
            for opt in (opt_o, opt_6):
                if it % decay_every == 0:
                    for g in opt.param_groups:
                        g["lr"] = g["lr"] * decay
            loss_o, _, da_o, db_o = ostep.train_step(o32, opt_o, img_a, img_b, lists, synth.LOSS_CONFIG)
            l6.append(float(ostep.train_step(o64, opt_6, img_a.double(), img_b.double(), lists, synth.LOSS_CONFIG)[0].item()))
            lo.append(float(loss_o.item()))
            row = []
            for b in range(B):   # the float32 oracle's own hard-negative counts, from the descriptor maps its step produced
                pa, pb = ostep.process_network_output(da_o[b:b + 1], 1), ostep.process_network_output(db_o[b:b + 1], 1)
                row.append(sum(ostep.PixelwiseContrastiveLoss.non_match_descriptor_loss(
                    pa, pb, lists[b][n + "_non_matches_a"], lists[b][n + "_non_matches_b"], M=synth.LOSS_CONFIG["M_" + n])[1]
                    for n in ("masked", "background")))
            h_o.append(row)
        with torch.no_grad():
            o32.eval(); o64.eval()
            _ORACLE_TRAJECTORIES[key] = (lo, l6, h_o, o32(img_a).double(), o64(img_a.double()))
    lo, l6, h_o, y3, y6 = _ORACLE_TRAJECTORIES[key]
    # ---- the product
    dcn.train()
    opt_p = Adam(dcn.parameters(), lr=lr, weight_decay=1e-4)
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    xa, xb = img_a.to(device), img_b.to(device)
    pl = PairLists.from_lists([tuple(L[k] for k in KEYS) for L in lists], device, hw=H * W)
    lp, slack = [], []
    for it in range(1, steps + 1):
        if it % decay_every == 0:
            for g in opt_p.param_groups:
                g["lr"] = g["lr"] * decay
        opt_p.zero_grad()
        if separate_forwards:
            ya, yb = dcn.forward(xa), dcn.forward(xb)
        else:
            ya, yb = dcn.forward_pair(xa, xb)
        loss_p, _, hard_p = loss_composer.get_loss_batched(pcl, 0, dcn.process_network_output(ya, B), dcn.process_network_output(yb, B), pl)
        loss_p.backward()
        opt_p.step()
        lp.append(float(loss_p.item()))
        hp = hard_p.cpu()
        s_it = 0.0
        for b in range(B):
            h_p = int(hp[b, 1]) + int(hp[b, 2])
            s_it = max(s_it, abs(h_p - h_o[it - 1][b]) / max(min(h_p, h_o[it - 1][b]), 1))
        slack.append(s_it)
    with torch.no_grad():
        dcn.eval()
        yp = dcn.forward(xa).cpu().double()
        dcn.train()
    centre = lambda y: y - y.mean(dim=(0, 2, 3), keepdim=True)
    yp, y3, y6 = centre(yp), centre(y3), centre(y6)
    rel = lambda a, b: [abs(x - y) / abs(y) for x, y in zip(a, b)]
    return {"loss_p": lp, "loss_o32": lo, "loss_o64": l6, "dev_p": rel(lp, l6), "dev_o32": rel(lo, l6), "slack": slack,
            "desc_p": float((yp - y6).abs().max() / y6.abs().max()), "desc_o32": float((y3 - y6).abs().max() / y6.abs().max())}


def assert_trajectory_as_close_as_float32(r, factor=3.0, floor=1e-4, first=1e-4):
    """Every step's loss deviation from the float64 trajectory within `factor` x the largest deviation the float32 oracle has
    shown up to that step (+ floor + tie slack); the first step -- no optimizer step behind it -- within the north star's 1e-4."""
    assert r["dev_p"][0] < first + r["slack"][0], r
    env = 0.0
    for t, (dp, do, sl) in enumerate(zip(r["dev_p"], r["dev_o32"], r["slack"])):
        env = max(env, do)
        assert dp <= factor * env + floor + 1.5 * sl, (t, dp, env, sl, r)
    assert r["desc_p"] <= factor * r["desc_o32"] + 10 * floor, r
