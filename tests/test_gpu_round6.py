"""-m gpu tests of round 6: the order-independent loss backward (64-bit fixed-point accumulation with integer atomics,
csrc/loss_kernels.hip) at BASELINE configs[1]'s sizes, and what it buys -- a training step whose parameters are bit-identical
run to run (the fp32 atomics of the loss scatter were the one non-reproducible kernel of the step)."""
import pytest
import torch

from helpers import rel_err, use_gfx950_library

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = use_gfx950_library()
    assert torch.cuda.is_available()
    return lib


def _loss_backward(exact, da, db, lists, pcl):
    from dcn_hip import loss as K
    from dense_correspondence.loss_functions import loss_composer
    old = K.EXACT_BACKWARD
    K.EXACT_BACKWARD = exact
    try:
        a, b = da.clone().requires_grad_(True), db.clone().requires_grad_(True)
        loss = loss_composer.get_loss_batched(pcl, 0, a, b, lists)[0]
        ga, gb = torch.autograd.grad(loss, [a, b])
        return loss.detach(), ga, gb
    finally:
        K.EXACT_BACKWARD = old


@pytest.mark.parametrize("B,D,pairs", [(4, 3, (5000, 2500, 2500)), (2, 16, (10000, 50000, 50000))])
def test_exact_loss_backward_full_size(L, B, D, pairs):
    from dcn_hip.loss import PairLists
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import synth
    H, W = 480, 640
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    lists = []
    for _ in range(B):
        t8 = []
        for n in pairs:
            t8 += [torch.randint(0, H * W, (n,), generator=g, dtype=torch.int64) for _ in range(2)]
        # duplicates on purpose: the first 64 matches of image a all land on one pixel
        t8[0][:64] = 12345
        t8 += [torch.tensor([-1])] * 2
        lists.append(tuple(t8))
    pl = PairLists.from_lists(lists, dev, hw=H * W)
    gg = torch.Generator(device=dev).manual_seed(4)
    da = (torch.rand(B, H * W, D, device=dev, generator=gg) * 2 - 1) * 0.3
    db = (torch.rand(B, H * W, D, device=dev, generator=gg) * 2 - 1) * 0.3
    pcl = PixelwiseContrastiveLoss(image_shape=[H, W], config=synth.LOSS_CONFIG)
    l0, ga0, gb0 = _loss_backward(True, da, db, pl, pcl)
    for _ in range(3):                               # bit for bit, launch after launch
        l1, ga1, gb1 = _loss_backward(True, da, db, pl, pcl)
        assert torch.equal(ga0, ga1) and torch.equal(gb0, gb1) and torch.equal(l0, l1)
    # another order of the same pairs: other lanes / workgroups / arrival order of the atomics
    shuffled = []
    for t8 in lists:
        s8 = []
        for t in range(3):
            perm = torch.randperm(t8[2 * t].numel(), generator=g)
            s8 += [t8[2 * t][perm], t8[2 * t + 1][perm]]
        shuffled.append(tuple(s8) + t8[6:])
    _, ga2, gb2 = _loss_backward(True, da, db, PairLists.from_lists(shuffled, dev, hw=H * W), pcl)
    assert torch.equal(ga0, ga2) and torch.equal(gb0, gb2)
    # against the fp32-atomics path (order-dependent in the last bits) and float64 on the host for a slice
    _, fa, fb = _loss_backward(False, da, db, pl, pcl)
    assert rel_err(ga0.cpu(), fa.cpu()) < 2e-6 and rel_err(gb0.cpu(), fb.cpu()) < 2e-6
    from oracle import loss_oracle
    a64 = da[:1].double().cpu().requires_grad_(True)
    b64 = db[:1].double().cpu().requires_grad_(True)
    po = loss_oracle.PixelwiseContrastiveLoss([H, W], synth.LOSS_CONFIG)
    (loss_oracle.get_loss(po, torch.tensor([0]), a64, b64, *lists[0])[0] / B).backward()
    assert rel_err(ga0[:1].cpu(), a64.grad) < 2e-6 and rel_err(gb0[:1].cpu(), b64.grad) < 2e-6


def test_training_steps_are_bit_reproducible(L):
    """Two runs of the same three iterations (reference call sequence, Adam) from the same state: every parameter, BN buffer
    and the logged loss identical to the bit.  Resnet34_8s at 240 x 320, two image pairs per step."""
    from dcn_hip.loss import PairLists
    from dcn_hip.optim import Adam
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from dense_correspondence.network.dense_correspondence_network import DenseCorrespondenceNetwork
    from oracle import synth
    H, W, D, B = 240, 320, 3, 2
    dev = torch.device("cuda:0")
    cfg = {"descriptor_dimension": D, "image_width": W, "image_height": H}
    img_a, img_b, lists = synth.make_batch(B, H, W, 3000, 1500, 1500, seed=3)
    keys = ("matches_a", "matches_b", "masked_non_matches_a", "masked_non_matches_b", "background_non_matches_a",
            "background_non_matches_b", "blind_non_matches_a", "blind_non_matches_b")
    pl = PairLists.from_lists([tuple(Ld[k] for k in keys) for Ld in lists], dev, hw=H * W)
    img_a, img_b = img_a.to(dev), img_b.to(dev)
    torch.manual_seed(0)
    first = DenseCorrespondenceNetwork.from_config(dict(cfg), load_stored_params=False).to(dev)
    state0 = {k: v.clone() for k, v in first.state_dict().items()}

    def run(dcn, pair_call):
        dcn.load_state_dict(state0)
        dcn.train()
        opt = Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4)
        pcl = PixelwiseContrastiveLoss(image_shape=dcn.image_shape, config=synth.LOSS_CONFIG)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            if pair_call:
                ya, yb = dcn.forward_pair(img_a, img_b)
            else:
                ya, yb = dcn.forward(img_a), dcn.forward(img_b)
            loss = loss_composer.get_loss_batched(pcl, 0, dcn.process_network_output(ya, B), dcn.process_network_output(yb, B), pl)[0]
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return losses, {k: v.clone() for k, v in dcn.state_dict().items()}

    for pair_call in (True, False):
        l1, s1 = run(first, pair_call)
        second = DenseCorrespondenceNetwork.from_config(dict(cfg), load_stored_params=False).to(dev)
        l2, s2 = run(second, pair_call)
        assert all(torch.equal(a, b) for a, b in zip(l1, l2))
        bad = [k for k in s1 if not torch.equal(s1[k], s2[k])]
        assert not bad, (pair_call, bad[:5])
