"""One full reference training step (training.py:325-346) through the drop-in API --
DenseCorrespondenceNetwork.from_config -> forward x2 -> process_network_output -> loss_composer.get_loss ->
backward -> Adam -- against the oracle's step, on a narrow network (kernels host-emulated; CPU only)."""
import pytest
import torch

from helpers import rel_err, use_emulation_library


@pytest.fixture(scope="module", autouse=True)
def _lib():
    return use_emulation_library()


FC_SCALE = 4.0


def _make(arch="Resnet18_8s", D=3, H=32, W=48, bw=8):
    import pytorch_segmentation_detection.models.resnet_dilated as rd
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    from oracle import resnet_dilated_oracle as orc

    class Narrow(getattr(rd, arch)):   # from_config passes only num_classes (network.py:375)
        def __init__(self, num_classes):
            super(Narrow, self).__init__(num_classes=num_classes, base_width=bw)
    rd.NarrowTestNet = Narrow
    Narrow.arch, Narrow.attr = getattr(rd, arch).arch, getattr(rd, arch).attr
    cfg = {"descriptor_dimension": D, "image_width": W, "image_height": H, "normalize": False,
           "backbone": {"model_class": "Resnet", "resnet_name": "NarrowTestNet"}}
    dcn = DenseCorrespondenceNetwork.from_config(cfg, load_stored_params=False)
    o = orc.build(arch, D, seed=0, base_width=bw)
    dcn.fcn.load_state_dict(o.state_dict())
    return dcn, o


def test_api_surface_and_state_dict_keys():
    dcn, o = _make()
    assert dcn.training and dcn.descriptor_dimension == 3 and dcn.image_shape == [32, 48]
    keys = list(dcn.state_dict().keys())
    assert keys[0] == "_fcn.resnet18_8s.conv1.weight" and "_fcn.resnet18_8s.fc.bias" in keys
    assert [k[5:] for k in keys] == list(o.state_dict().keys())
    x = torch.randn(1, 3, 32, 48)
    y = dcn.forward(x)
    p = dcn.process_network_output(y, 1)
    assert p.shape == (1, 32 * 48, 3) and p.is_contiguous()
    # same values as the reference's view/permute on an NCHW tensor
    ref = y.contiguous().view(1, 3, 48 * 32).permute(0, 2, 1)
    assert torch.equal(p, ref)
    s = dcn.forward_single_image_tensor(x[0])
    assert s.shape == (32, 48, 3)
    with pytest.raises(ValueError):
        dcn.path_to_network_params_folder
    uv, diff, norm = dcn.find_best_match((3, 4), s.detach().numpy(), s.detach().numpy())
    assert uv == (3, 4) and diff == 0.0 and norm.shape == (32, 48)


@pytest.mark.parametrize("B", [1, 2])
def test_training_step_matches_oracle(B):
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import step as ostep, synth
    H, W, D = 32, 48, 3
    dcn, o = _make(D=D, H=H, W=W)
    img_a, img_b, lists = synth.make_batch(B, H, W, 60, 30, 30, seed=1)
    # untrained descriptors are tiny -> every non-match is a hard negative; scale the fc layer so both hinge sides occur
    with torch.no_grad():
        for net in (dcn.fcn.resnet18_8s, o.resnet18_8s):
            net.fc.weight.mul_(FC_SCALE)
    o.train(); dcn.train()
    opt_o = torch.optim.Adam(o.parameters(), lr=1e-4, weight_decay=1e-4)
    from dcn_hip.optim import Adam
    opt_m = Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4)
    pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
    for it in range(2):
        loss_o, terms_o, da_o, db_o = ostep.train_step(o, opt_o, img_a, img_b, lists, synth.LOSS_CONFIG)
        opt_m.zero_grad()
        pa = dcn.process_network_output(dcn.forward(img_a), B)
        pb = dcn.process_network_output(dcn.forward(img_b), B)
        if B == 1:
            L = lists[0]
            loss, ml, mk, bg, bl = loss_composer.get_loss(
                pcl, torch.tensor([0]), pa, pb, L["matches_a"], L["matches_b"], L["masked_non_matches_a"],
                L["masked_non_matches_b"], L["background_non_matches_a"], L["background_non_matches_b"],
                L["blind_non_matches_a"], L["blind_non_matches_b"])
            assert abs(ml.item() - terms_o[0][1].item()) <= 1e-4 * abs(terms_o[0][1].item())
            assert abs(mk.item() - terms_o[0][2].item()) <= 1e-4 * abs(terms_o[0][2].item()) + 1e-9
            assert abs(bg.item() - terms_o[0][3].item()) <= 1e-4 * abs(terms_o[0][3].item()) + 1e-9
            assert bl.item() == 0.0
            if it == 0:   # the hinge must be exercised on both sides for this test to mean anything
                hn = loss_composer.get_loss_batched(pcl, 0, pa.detach(), pb.detach(), [tuple(
                    L[k] for k in ("matches_a", "matches_b", "masked_non_matches_a", "masked_non_matches_b",
                                   "background_non_matches_a", "background_non_matches_b",
                                   "blind_non_matches_a", "blind_non_matches_b"))])[2]
                assert 0 < int(hn[0, 1]) < 30, hn
        else:
            tup = [(L["matches_a"], L["matches_b"], L["masked_non_matches_a"], L["masked_non_matches_b"],
                    L["background_non_matches_a"], L["background_non_matches_b"], L["blind_non_matches_a"],
                    L["blind_non_matches_b"]) for L in lists]
            loss, terms, hard = loss_composer.get_loss_batched(pcl, 0, pa, pb, tup)
        assert rel_err(pa.detach().reshape(B, H, W, D).permute(0, 3, 1, 2), da_o) < 1e-4
        assert abs(loss.item() - loss_o.item()) <= 1e-4 * abs(loss_o.item()), (it, loss.item(), loss_o.item())
        loss.backward()
        opt_m.step()
        for (k, p), (_, po) in zip(dcn.fcn.named_parameters(), o.named_parameters()):
            # fc.bias has a mathematically zero gradient (the loss only sees descriptor differences): pure round-off
            tol = 2e-3 * float(po.grad.abs().max()) + 1e-6
            assert float((p.grad - po.grad).abs().max()) < tol, (it, k, rel_err(p.grad, po.grad))
            # Adam's first updates are lr * sign(g): a gradient that is ~0 may pick the other sign, 2 * lr apart
            assert float((p - po).abs().max()) < 2.5e-4, (it, k)
        # continue from IDENTICAL parameters, otherwise those sign flips (not the kernels) are what iteration 2 compares
        dcn.fcn.load_state_dict(o.state_dict())


def test_checkpoint_round_trip_through_model_folder(tmp_path):
    """training.py:501-521 saves `dcn.state_dict()` as %06d.pth next to training.yaml; network.py:441-485 loads it back.
    Keys must be the reference's (`_fcn.<backbone>.…`), including BN buffers; the legacy fallback (state dict of the
    bare fcn, network.py:429-433) must load too."""
    import yaml
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    dcn, o = _make()
    x = torch.randn(1, 3, 32, 48)
    dcn.train()
    with torch.no_grad():
        dcn.forward(x)                       # moves the BN running statistics
    dcn.eval()
    with torch.no_grad():
        y_ref = dcn.forward(x).clone()
    sd = dcn.state_dict()
    assert all(k.startswith("_fcn.resnet18_8s.") for k in sd) and "_fcn.resnet18_8s.bn1.num_batches_tracked" in sd
    torch.save(sd, str(tmp_path / "003500.pth"))
    torch.save({"dummy": 0}, str(tmp_path / "003500.pth.opt"))
    cfg = {"dense_correspondence_network": {"descriptor_dimension": 3, "image_width": 48, "image_height": 32,
                                            "backbone": {"model_class": "Resnet", "resnet_name": "NarrowTestNet"}}}
    (tmp_path / "training.yaml").write_text(yaml.safe_dump(cfg))
    loaded = DenseCorrespondenceNetwork.from_model_folder(str(tmp_path))
    assert loaded.constructed_from_model_folder and loaded.config["model_param_filename_tail"] == "003500.pth"
    loaded.eval()
    with torch.no_grad():
        assert torch.equal(loaded.forward(x), y_ref)
    # conv weights keep the kernels' channels_last storage after load_state_dict
    w = loaded.fcn.resnet18_8s.get_parameter("layer1.0.conv1.weight")
    assert w.is_contiguous(memory_format=torch.channels_last)
    # legacy checkpoints hold the bare backbone's state dict
    torch.save(dcn.fcn.state_dict(), str(tmp_path / "legacy.pth"))
    legacy = DenseCorrespondenceNetwork.from_config(dict(cfg["dense_correspondence_network"]), load_stored_params=True,
                                                    model_param_file=str(tmp_path / "legacy.pth"))
    legacy.eval()
    with torch.no_grad():
        assert torch.equal(legacy.forward(x), y_ref)


def test_batched_descriptor_image_export(tmp_path):
    """evaluation/utils.py:109-160 and compute_descriptor_images.py:38-72: same files, names and contents as the reference's
    one-image-at-a-time loops, computed ``batch_size`` images per engine call."""
    import os
    import numpy as np
    from dense_correspondence.evaluation import utils as eval_utils
    dcn, _ = _make()
    H, W = 32, 48
    g = torch.Generator().manual_seed(9)
    images = {idx: torch.randn(3, H, W, generator=g) for idx in (3, 0, 12, 7, 1)}

    class Scene(object):   # the slice of SpartanDataset the export touches
        def get_pose_data(self, scene_name):
            assert scene_name == "scene-a"
            return {k: None for k in images}

        def get_rgb_image_from_scene_name_and_idx(self, scene_name, idx):
            return ("rgb", idx)

        def get_rgbd_mask_pose(self, scene_name, idx):
            return ("rgb", idx), None, None, None

        def rgb_image_to_tensor(self, rgb):
            return images[rgb[1]]
    dcn.eval()
    ref = {idx: dcn.forward_single_image_tensor(t).detach().clone() for idx, t in images.items()}   # the reference's loop
    dcn.train()
    out = str(tmp_path / "desc")
    eval_utils.extract_descriptor_images_for_scene(dcn, Scene(), "scene-a", out, batch_size=2)
    assert dcn.training                                           # mode restored
    assert sorted(os.listdir(out)) == ["%06d_descriptor.npy" % i for i in sorted(images)]
    for idx in images:
        arr = np.load(os.path.join(out, "%06d_descriptor.npy" % idx))
        assert arr.shape == (H, W, 3) and arr.dtype == np.float32
        assert rel_err(torch.from_numpy(arr), ref[idx]) < 1e-5   # (batch of 2 vs batch of 1: eval mode, no batch statistics)
    with pytest.raises(ValueError):
        eval_utils.extract_descriptor_images_for_scene(dcn, Scene(), "scene-a", out)      # exists, overwrite=False
    eval_utils.extract_descriptor_images_for_scene(dcn, Scene(), "scene-a", out, overwrite=True, batch_size=8)
    out2 = str(tmp_path / "desc2")
    n = eval_utils.compute_descriptor_images_for_single_scene(Scene(), "scene-a", dcn, out2, batch_size=3)
    assert n == 5 and sorted(os.listdir(out2)) == ["%06d_descriptor_image.npy" % i for i in sorted(images)]
    assert rel_err(torch.from_numpy(np.load(os.path.join(out2, "000012_descriptor_image.npy"))), ref[12]) < 1e-5


@pytest.mark.parametrize("mode,separate", [("f16x3", True), ("f16x3", False), ("fp32", True)])
def test_training_trajectory_tracks_the_oracle(mode, separate):
    """Six iterations of the reference's loop (training.py:325-346 with the learning-rate decay of :544-558) on a narrow
    network, kernels host-emulated, against the float64 oracle with the float32 oracle as yard-stick
    (parity_common.run_trajectory; the GPU suite runs 12 steps of the real Resnet34_8s: tests/test_gpu_round4.py).  The floor
    is 1e-3 here: on the 8 x 8 maps of this network ONE pre-activation within round-off of zero that a different summation
    order puts on the other side of the ReLU moves a channel's gradient by a per cent -- Adam then walks that weight the other
    way (seen once in six steps at these sizes, in both arithmetics)."""
    import warnings
    import parity_common as pc
    from dcn_hip import backbone as bb
    bb.set_conv_mode(mode)
    try:
        H, W = (32, 48) if separate else (64, 64)   # (the grouped launch needs whole 64-row tiles per image batch at 1/8 resolution)
        dcn, o = _make(D=3, H=H, W=W)
        with warnings.catch_warnings():
            warnings.simplefilter("error")          # forward_pair silently falling back to two calls would not test the pair path
            r = pc.run_trajectory(dcn, o, 2, H, W, 6, torch.device("cpu"), pairs=(60, 30, 30), decay_every=2,
                                  separate_forwards=separate, fc_scale=FC_SCALE)
    finally:
        bb.set_conv_mode(None)
    assert r["loss_o64"][-1] < r["loss_o64"][0]
    pc.assert_trajectory_as_close_as_float32(r, floor=1e-3)
