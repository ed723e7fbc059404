"""The oracle against (1) golden vectors produced by the reference's own source text and
(2) an independent float64 numpy restatement.  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import loss_numpy, loss_oracle, resnet_dilated_oracle, synth


def _load(path):
    z = np.load(path, allow_pickle=False)
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        k = str(k)
        cfg[k] = bool(v) if k.startswith(("use_", "scale_")) else float(v)
    return z, cfg


GOLDENS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "loss_ref_*.npz")))


def test_goldens_present():
    assert len(GOLDENS) >= 10


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[9:-4] for p in GOLDENS])
def test_loss_oracle_matches_reference_goldens(path):
    z, cfg = _load(path)
    A = torch.tensor(z["A"], requires_grad=True)
    B = torch.tensor(z["B"], requires_grad=True)
    t = lambda k: torch.tensor(z[k])
    pcl = loss_oracle.PixelwiseContrastiveLoss([int(z["H"]), int(z["W"])], cfg)
    out = loss_oracle.get_loss(pcl, torch.tensor([int(z["match_type"])]), A, B, t("matches_a"), t("matches_b"),
                               t("masked_a"), t("masked_b"), t("background_a"), t("background_b"),
                               t("blind_a"), t("blind_b"))
    got = np.array([float(o.sum().item()) for o in out])
    np.testing.assert_allclose(got, z["out"], rtol=1e-6, atol=1e-9)
    if out[0].requires_grad:
        out[0].sum().backward()
        np.testing.assert_allclose(A.grad.numpy(), z["gradA"], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(B.grad.numpy(), z["gradB"], rtol=1e-5, atol=1e-8)
    if "f7_vec" in z.files:
        PCL = loss_oracle.PixelwiseContrastiveLoss
        with torch.no_grad():
            ml, _, _ = PCL.match_loss(A, B, t("matches_a"), t("matches_b"))
            vec, hn, _, _ = PCL.non_match_descriptor_loss(A, B, t("masked_a"), t("masked_b"), M=cfg["M_masked"])
            veci, hni, _, _ = PCL.non_match_descriptor_loss(A, B, t("masked_a"), t("masked_b"), M=cfg["M_masked"],
                                                            invert=True)
        np.testing.assert_allclose(ml.numpy(), z["f6_match_loss"], rtol=1e-6)
        np.testing.assert_allclose(vec.numpy(), z["f7_vec"], rtol=1e-6, atol=1e-9)
        assert hn == int(z["f7_hard"]) and hni == int(z["f7_hard_invert"])
        np.testing.assert_allclose(veci.numpy(), z["f7_vec_invert"], rtol=1e-6, atol=1e-9)
    if "triplet" in z.files:
        with torch.no_grad():
            trip = loss_oracle.PixelwiseContrastiveLoss.get_triplet_loss(
                A, B, t("matches_a"), t("matches_b"), t("triplet_non_matches_a"), t("masked_b"), cfg["alpha_triplet"])
        np.testing.assert_allclose(trip.numpy(), z["triplet"], rtol=1e-6)


@pytest.mark.parametrize("path", [p for p in GOLDENS if "within" in p or "multi" in p],
                         ids=lambda p: os.path.basename(p)[9:-4])
def test_numpy_second_opinion_matches_reference_goldens(path):
    z, cfg = _load(path)
    if len(z["blind_a"]) > 1:
        pytest.skip("blind term is not part of `loss`; covered by the torch oracle")
    lists = {k: z[k] for k in ("matches_a", "matches_b", "masked_a", "masked_b", "background_a", "background_b")}
    r = loss_numpy.within_scene(z["A"][0], z["B"][0], lists, cfg, int(z["W"]))
    np.testing.assert_allclose([r["loss"], r["match_loss"], r["masked"], r["background"]], z["out"][:4],
                               rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(r["gradA"], z["gradA"][0], rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(r["gradB"], z["gradB"][0], rtol=2e-4, atol=2e-7)


def test_backbone_oracle_param_count_and_shapes():
    # SURVEY.md 8a F1 / BASELINE.md section 2: 21 286 211 params at D=3; 23 573 600 for Resnet50_8s D=32
    m = resnet_dilated_oracle.build("Resnet34_8s", 3)
    assert sum(p.numel() for p in m.parameters()) == 21286211
    keys = list(m.state_dict().keys())
    assert keys[0] == "resnet34_8s.conv1.weight" and "resnet34_8s.layer3.0.downsample.0.weight" in keys
    assert "resnet34_8s.fc.bias" in keys
    # dilation pattern: layer3 d=2 everywhere incl. first block, layer4 d=4, downsample stride 1
    r = m.resnet34_8s
    assert r.layer3[0].conv1.dilation == (2, 2) and r.layer3[0].conv1.stride == (1, 1)
    assert r.layer4[0].conv1.dilation == (4, 4) and r.layer4[2].conv2.padding == (4, 4)
    assert r.layer3[0].downsample[0].stride == (1, 1) and r.layer2[0].downsample[0].stride == (2, 2)
    m50 = resnet_dilated_oracle.build("Resnet50_8s", 32)
    assert sum(p.numel() for p in m50.parameters()) == 23573600
    x = torch.randn(2, 3, 32, 48)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (2, 3, 32, 48)


def test_backbone_oracle_is_seed_deterministic():
    a = resnet_dilated_oracle.build("Resnet18_8s", 3, seed=0, base_width=8)
    b = resnet_dilated_oracle.build("Resnet18_8s", 3, seed=0, base_width=8)
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k


def test_synth_shapes_and_sentinels():
    a, b, lists = synth.make_batch(2, 16, 24, 10, 5, 0, seed=1)
    assert a.shape == (2, 3, 16, 24) and len(lists) == 2
    assert lists[0]["matches_a"].dtype == torch.int64 and lists[0]["matches_a"].max() < 16 * 24
    assert loss_oracle.is_empty(lists[0]["background_non_matches_a"])
    assert loss_oracle.is_empty(lists[0]["blind_non_matches_a"])


# ---------------------------------------------------------------------------------------------- pair generation (8f-2)
CORR_GOLDENS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "corr_ref_*.npz")))


@pytest.mark.parametrize("path", CORR_GOLDENS, ids=[os.path.basename(p)[:-4] for p in CORR_GOLDENS])
def test_correspondence_oracle_matches_reference_goldens(path):
    """oracle/correspondence_oracle.py against the outputs of the reference's own correspondence_finder source
    (tests/golden/make_correspondence_goldens_from_reference.py)."""
    from oracle import correspondence_oracle as co
    z = np.load(path)
    uv_a, uv_b = co.find_correspondences_for_candidates(z["depth_a"], z["pose_a"], z["depth_b"], z["pose_b"],
                                                        torch.tensor(z["cand_u"]), torch.tensor(z["cand_v"]))
    assert np.array_equal(uv_a[0].numpy(), z["uv_a_u"]) and np.array_equal(uv_a[1].numpy(), z["uv_a_v"])
    np.testing.assert_allclose(uv_b[0].numpy(), z["uv_b_u"], rtol=1e-6)
    np.testing.assert_allclose(uv_b[1].numpy(), z["uv_b_v"], rtol=1e-6)
    mask = torch.tensor(z["mask"]) if z["mask"].size else None
    H, W = z["depth_a"].shape
    nu, nv = co.create_non_correspondences(len(z["uv_a_u"]), (H, W), int(z["per_match"]), mask, torch.tensor(z["rand"]))
    assert np.array_equal(nu.numpy(), z["non_u"]) and np.array_equal(nv.numpy(), z["non_v"])


def test_evaluation_oracle_matches_reference_golden():
    """oracle/evaluation_oracle.py against the outputs of the reference's own find_best_match + statistics block
    (tests/golden/make_eval_goldens_from_reference.py)."""
    from oracle import evaluation_oracle as eo
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_ref.npz"))
    for q, (u, v) in enumerate(z["uv"]):
        s = eo.match_statistics((int(u), int(v)), (int(u), int(v)), z["res_a"], z["res_b"], z["mask_b"])
        assert tuple(s["uv_b_pred"]) == tuple(int(x) for x in z["uv_b_pred"][q])
        assert tuple(s["uv_b_pred_masked"]) == tuple(int(x) for x in z["uv_b_pred_masked"][q])
        for k_o, k_g in (("norm_diff_pred", "best_match_diff"), ("norm_diff_pred_masked", "best_match_diff_masked"),
                         ("pixel_match_error_l2", "pixel_match_error_l2"), ("pixel_match_error_l1", "pixel_match_error_l1"),
                         ("pixel_match_error_l2_masked", "pixel_match_error_l2_masked"),
                         ("norm_diff_descriptor_ground_truth", "norm_diff_descriptor_ground_truth"),
                         ("num_pixels_closer_than_ground_truth", "num_pixels_closer_than_ground_truth"),
                         ("num_pixels_closer_than_ground_truth_masked", "num_pixels_closer_than_ground_truth_masked"),
                         ("fraction_pixels_closer_than_ground_truth", "fraction_pixels_closer_than_ground_truth"),
                         ("fraction_pixels_closer_than_ground_truth_masked", "fraction_pixels_closer_than_ground_truth_masked"),
                         ("average_l2_distance_for_false_positives", "average_l2_distance_for_false_positives"),
                         ("average_l2_distance_for_false_positives_masked", "average_l2_distance_for_false_positives_masked")):
            np.testing.assert_allclose(float(s[k_o]), float(z[k_g][q]), rtol=1e-12, err_msg=k_o)


# ---------------------------------------------------------------------------------------------------------------------
# The backbone is un-vendored: no reference-held vector can pin it.  What CAN be done is to remove the
# single-derivation risk: a second statement of the architecture, derived the other way round (stock ResNet + FCN surgery
# as a pass over a flat record list, functional interpreter -- tests/backbone_second_statement.py), must agree with the
# oracle bit for bit, and the checkpoint layout of oracle, product and second statement must equal a committed fixture.
def _layout_fixture(name):
    rows = []
    for line in open(os.path.join(os.path.dirname(__file__), "golden", name)):
        if line.startswith("#") or not line.strip():
            continue
        k, shp = line.split()
        rows.append((k, () if shp == "scalar" else tuple(int(v) for v in shp.split("x"))))
    return rows


@pytest.mark.parametrize("arch,D,fixture", [("Resnet34_8s", 3, "resnet34_8s_d3_state_dict_layout.txt"),
                                            ("Resnet50_8s", 32, "resnet50_8s_d32_state_dict_layout.txt")])
def test_state_dict_layout_fixture(arch, D, fixture):
    """Ordered key / shape list of the checkpoint (`_fcn.` + these keys under DenseCorrespondenceNetwork, network.py:43):
    oracle == committed fixture == the layout derived from the stock-ResNet record list; the trunk part (everything but fc)
    is key for key the layout of a stock torchvision checkpoint, which is what `pretrained=True` of the original loads."""
    import backbone_second_statement as second
    want = _layout_fixture(fixture)
    o = resnet_dilated_oracle.build(arch, D)
    got = [(k, tuple(v.shape)) for k, v in o.state_dict().items()]
    assert got == want
    assert second.expected_state_dict_layout(arch, D) == want
    pref = arch.lower() + "."
    assert all(k.startswith(pref) for k, _ in want)
    trunk = [k[len(pref):] for k, _ in want if not k.startswith(pref + "fc.")]
    assert trunk[:6] == ["conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var", "bn1.num_batches_tracked"]
    assert "layer2.0.downsample.0.weight" in trunk and "layer1.0.downsample.0.weight" not in trunk or arch == "Resnet50_8s"
    assert sum(int(np.prod(s)) for k, s in want if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))) == \
        sum(p.numel() for p in o.parameters())


@pytest.mark.parametrize("arch,bw,shape,D", [("Resnet18_8s", 8, (2, 40, 56), 3), ("Resnet34_8s", 16, (2, 32, 48), 3),
                                             ("Resnet50_8s", 8, (1, 48, 40), 5)])
def test_second_statement_of_the_backbone_equals_the_oracle_bitwise(arch, bw, shape, D):
    import backbone_second_statement as second
    N, H, W = shape
    o = resnet_dilated_oracle.build(arch, D, seed=2, base_width=bw)
    o.train()
    sd = {k: v.clone() for k, v in o.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    x = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(5))
    gy = torch.randn(N, D, H, W, generator=torch.Generator().manual_seed(6))
    y1 = o(x)
    y2 = second.run(arch, sd, x, base_width=bw)
    assert y1.shape == (N, D, H, W) and torch.equal(y1, y2)
    (y1 * gy).sum().backward()
    (y2 * gy).sum().backward()
    for k, p in o.named_parameters():
        assert torch.equal(p.grad, sd[k].grad), k
    for k, b in o.named_buffers():          # running statistics and counters follow nn.BatchNorm2d in both
        assert torch.equal(b, sd[k].detach()), k
    # the surgery really did what the fork does: the FIRST block of layer3 / layer4 is dilated, strides 1, padding = dilation
    net = second.to_output_stride(second.stock_spec(arch, bw))
    first3 = [b for b in net["blocks"] if b["convs"][0]["name"].startswith("layer3.0.")][0]
    first4 = [b for b in net["blocks"] if b["convs"][0]["name"].startswith("layer4.0.")][0]
    assert {c["dil"] for c in first3["convs"] if c["k"] == 3} == {2} and {c["dil"] for c in first4["convs"] if c["k"] == 3} == {4}
    assert all(c["stride"] == 1 for b in (first3, first4) for c in b["convs"] + [b["proj"]])
    trunk = getattr(o, arch.lower())
    assert trunk.layer3[0].conv2.dilation == (2, 2) and trunk.layer4[0].conv2.padding == (4, 4) and trunk.layer4[0].conv2.stride == (1, 1)


def test_torchvision_surgery_equals_the_oracle_bitwise():
    """The same check against a network made from the installed torchvision by the fork's surgery (skipped where torchvision
    is not installed -- the authoring image and the GPU image have none)."""
    pytest.importorskip("torchvision")
    import backbone_second_statement as second
    o = resnet_dilated_oracle.build("Resnet34_8s", 3, seed=1)
    tv = second.torchvision_surgery("Resnet34_8s", 3)
    sd = {k[len("resnet34_8s."):]: v for k, v in o.state_dict().items()}
    tv.tv.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc.")}, strict=False)
    tv.fc.load_state_dict({"weight": sd["fc.weight"], "bias": sd["fc.bias"]})
    o.train(); tv.train()
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(2))
    assert torch.equal(o(x), tv(x))
