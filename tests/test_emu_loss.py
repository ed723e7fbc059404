"""Loss kernels (K9) through the C ABI -- kernels compiled for the host (tests/hostemu) -- against
(1) the golden vectors produced by the reference's own source and (2) the oracle on seeded inputs.
CPU only; the same checks run on the real gfx950 build in test_gpu_parity.py."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import lists_from_golden, load_golden_loss, rel_err, use_emulation_library

GOLDENS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "loss_ref_*.npz")))


@pytest.fixture(scope="module", autouse=True)
def _lib():
    return use_emulation_library()


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[9:-4] for p in GOLDENS])
def test_composer_matches_reference_goldens(path):
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    z, cfg = load_golden_loss(path)
    A = torch.tensor(z["A"], requires_grad=True)
    B = torch.tensor(z["B"], requires_grad=True)
    pcl = PixelwiseContrastiveLoss([int(z["H"]), int(z["W"])], cfg)
    out = loss_composer.get_loss(pcl, torch.tensor([int(z["match_type"])]), A, B, *lists_from_golden(z))
    got = np.array([float(o.sum().item()) for o in out])
    np.testing.assert_allclose(got, z["out"], rtol=2e-6, atol=1e-9)   # tolerance 1e-4 required; we hold 2e-6
    out[0].backward()
    assert rel_err(A.grad, z["gradA"]) < 1e-5 and rel_err(B.grad, z["gradB"]) < 1e-5


def test_building_blocks_match_reference_goldens():
    """F6 / F7 / F8 called one by one, as pcl.py exposes them."""
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss as PCL
    z, cfg = load_golden_loss([p for p in GOLDENS if p.endswith("within_d16.npz")][0])
    A = torch.tensor(z["A"], requires_grad=True)
    B = torch.tensor(z["B"])
    ma, mb, ka, kb = [torch.tensor(z[k]) for k in ("matches_a", "matches_b", "masked_a", "masked_b")]
    ml, da, db = PCL.match_loss(A, B, ma, mb)
    np.testing.assert_allclose(ml.item(), z["f6_match_loss"], rtol=2e-6)
    assert da.shape == (1, len(ma), 16)
    vec, hn, _, _ = PCL.non_match_descriptor_loss(A, B, ka, kb, M=cfg["M_masked"])
    np.testing.assert_allclose(vec.detach().numpy(), z["f7_vec"], rtol=1e-5, atol=1e-8)  # (M - d)^2 near the margin amplifies 1-ulp differences of d
    assert hn == int(z["f7_hard"])
    veci, hni, _, _ = PCL.non_match_descriptor_loss(A, B, ka, kb, M=cfg["M_masked"], invert=True)
    np.testing.assert_allclose(veci.detach().numpy(), z["f7_vec_invert"], rtol=1e-5, atol=1e-8)
    assert hni == int(z["f7_hard_invert"])
    # the vector is differentiable, like the reference's
    w = torch.linspace(0.5, 1.5, len(ka))
    (vec * w).sum().backward()
    A2 = torch.tensor(z["A"], requires_grad=True)
    from oracle import loss_oracle
    v2, _, _, _ = loss_oracle.PixelwiseContrastiveLoss.non_match_descriptor_loss(A2, B, ka, kb, M=cfg["M_masked"])
    (v2 * w).sum().backward()
    assert rel_err(A.grad, A2.grad) < 1e-5
    pcl = PCL([int(z["H"]), int(z["W"])], cfg)
    s, h = pcl.non_match_loss_descriptor_only(A, B, ka, kb, M_descriptor=cfg["M_masked"])
    np.testing.assert_allclose(s.item(), z["f7_vec"].sum(), rtol=2e-6)
    assert h == int(z["f7_hard"])
    # triplet loss kernel: value against the REFERENCE's golden, gradients against the oracle's autograd
    tna = torch.tensor(z["triplet_non_matches_a"])
    A3 = torch.tensor(z["A"], requires_grad=True); B3 = torch.tensor(z["B"], requires_grad=True)
    trip = PCL.get_triplet_loss(A3, B3, ma, mb, tna, kb, cfg["alpha_triplet"])
    np.testing.assert_allclose(trip.item(), z["triplet"], rtol=1e-6)
    (trip * 1.7).backward()
    A4 = torch.tensor(z["A"], requires_grad=True); B4 = torch.tensor(z["B"], requires_grad=True)
    (loss_oracle.PixelwiseContrastiveLoss.get_triplet_loss(A4, B4, ma, mb, tna, kb, cfg["alpha_triplet"]) * 1.7).backward()
    assert rel_err(A3.grad, A4.grad) < 1e-5 and rel_err(B3.grad, B4.grad) < 1e-5
    with pytest.raises(ValueError):   # the reference's index_select shapes only agree for whole multiples
        PCL.get_triplet_loss(A3, B3, ma, mb, tna[:-1], kb[:-1], cfg["alpha_triplet"])
    orig = pcl.get_loss_original(A, B, ma, mb, ka, kb)
    np.testing.assert_allclose([o.item() for o in orig], z["original_loss"], rtol=1e-6)
    # ... and its gradient (hinge on the squared distance: kernel hinge mode 2) against the oracle's autograd
    A5 = torch.tensor(z["A"], requires_grad=True); B5 = torch.tensor(z["B"], requires_grad=True)
    A6 = torch.tensor(z["A"], requires_grad=True); B6 = torch.tensor(z["B"], requires_grad=True)
    pcl.get_loss_original(A5, B5, ma, mb, ka, kb, M_margin=0.3, non_match_loss_weight=0.7)[0].backward()
    opcl = loss_oracle.PixelwiseContrastiveLoss([int(z["H"]), int(z["W"])], cfg)
    lo = opcl.get_loss_original(A6, B6, ma, mb, ka, kb, M_margin=0.3, non_match_loss_weight=0.7)
    lo[0].backward()
    assert float(lo[2]) > 0 and rel_err(A5.grad, A6.grad) < 1e-5 and rel_err(B5.grad, B6.grad) < 1e-5


@pytest.mark.parametrize("D", [2, 3, 5, 8, 16, 32, 40])   # lane groups of 4 / 8 / 16 / 32 per pair; 40: two components per lane
def test_batched_ragged_lists_vs_oracle(D):
    """B = 3 pairs with different list lengths, an empty background list, duplicates, a device-side sentinel."""
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import loss_oracle, synth
    H, W, B = 24, 32, 3
    g = torch.Generator().manual_seed(7)
    A = ((torch.rand(B, H * W, D, generator=g) * 2 - 1) * 0.6 / D ** 0.5).requires_grad_(True)
    Bt = ((torch.rand(B, H * W, D, generator=g) * 2 - 1) * 0.6 / D ** 0.5).requires_grad_(True)
    sizes = [(50, 1100, 70), (1, 3, 2), (2049, 5, 1025)]   # crosses the pairs-per-workgroup chunk boundaries (64 .. 512)
    lists = []
    for pm, pk, pg in sizes:
        L = synth.make_index_lists(1, H * W, pm, pk, pg, g)[0]
        lists.append((L["matches_a"], L["matches_b"], L["masked_non_matches_a"], L["masked_non_matches_b"],
                      L["background_non_matches_a"], L["background_non_matches_b"],
                      L["blind_non_matches_a"], L["blind_non_matches_b"]))
    pcl = PixelwiseContrastiveLoss([H, W], synth.LOSS_CONFIG)
    loss, terms, hard = loss_composer.get_loss_batched(pcl, 0, A, Bt, lists)
    loss.backward()
    A2 = A.detach().clone().requires_grad_(True)
    B2 = Bt.detach().clone().requires_grad_(True)
    opcl = loss_oracle.PixelwiseContrastiveLoss([H, W], synth.LOSS_CONFIG)
    tot = 0
    for b in range(B):
        out = loss_oracle.get_loss(opcl, torch.tensor([0]), A2[b:b + 1], B2[b:b + 1], *lists[b])
        tot = tot + out[0]
        np.testing.assert_allclose(terms[b].numpy(), [float(o.sum()) for o in out], rtol=3e-6, atol=1e-9)
    (tot / B).backward()
    assert abs(loss.item() - (tot / B).item()) <= 3e-6 * abs(tot.item() / B)
    assert rel_err(A.grad, A2.grad) < 1e-5 and rel_err(Bt.grad, B2.grad) < 1e-5


def test_empty_lists_vs_numpy_oracle():
    """Empty masked / background lists (the reference itself would crash in index_select on the `[-1]` sentinel,
    so the independent numpy statement is the checker) -- also with the sentinel left on the 'device'."""
    from dcn_hip import loss as K
    from oracle import loss_numpy, synth
    H, W, D = 16, 16, 3
    g = torch.Generator().manual_seed(3)
    A = ((torch.rand(1, H * W, D, generator=g) * 2 - 1) * 0.3).requires_grad_(True)
    B = ((torch.rand(1, H * W, D, generator=g) * 2 - 1) * 0.3).requires_grad_(True)
    ma = torch.randint(0, H * W, (40,), generator=g)
    mb = torch.randint(0, H * W, (40,), generator=g)
    ka = torch.randint(0, H * W, (30,), generator=g)
    kb = torch.randint(0, H * W, (30,), generator=g)
    sentinel = torch.tensor([-1])
    lists = K.PairLists.from_lists([(ma, mb, ka, kb, sentinel, sentinel, None, None)], "cpu")
    assert lists.length(0, K.LIST_BACKGROUND) == 0
    c = synth.LOSS_CONFIG
    cfg = K.make_config([0, c["M_masked"], c["M_background"], c["M_masked"]], W)
    loss = K.contrastive_loss(A, B, lists, cfg)[0]
    loss.backward()
    ref = loss_numpy.within_scene(A.detach()[0].numpy(), B.detach()[0].numpy(),
                                  dict(matches_a=ma, matches_b=mb, masked_a=ka, masked_b=kb,
                                       background_a=[-1], background_b=[-1]), c, W)
    np.testing.assert_allclose(loss.item(), ref["loss"], rtol=3e-6)
    assert rel_err(A.grad[0], ref["gradA"]) < 1e-5
    # a sentinel that was NOT stripped on the host (as if it lived on the GPU): the kernel skips negative indices
    raw = K.PairLists(torch.cat([ma, ka, sentinel]), torch.cat([mb, kb, sentinel]), [0, 40, 70, 70, 71])
    loss2, terms2 = K.contrastive_loss(A.detach(), B.detach(), raw, cfg)[:2]
    np.testing.assert_allclose(loss2.item(), loss.item(), rtol=1e-7)
    assert terms2[0, 4].item() == 0.0


def test_index_out_of_range_sets_status_and_is_skipped():
    from dcn_hip import loss as K
    A = torch.rand(1, 100, 3)
    B = torch.rand(1, 100, 3)
    good = torch.tensor([1, 2, 3])
    bad = torch.tensor([1, 100, 3])   # 100 == HW is out of range
    lists = K.PairLists.from_lists([(good, bad, None, None, None, None, None, None)], "cpu")
    cfg = K.make_config([0, .5, .5, .5], 10)
    out = K.contrastive_loss(A, B, lists, cfg)
    assert int(out[4]) == 1
    ref = ((A[0, [1, 3]] - B[0, [1, 3]]) ** 2).sum() / 3
    np.testing.assert_allclose(out[0].item(), ref.item(), rtol=1e-6)


@pytest.mark.parametrize("P", [8, 9])   # 8: the single-launch finalize of the training configs; 9: one workgroup per pair + mean
def test_both_finalize_paths_agree_with_per_pair_calls_and_carry_the_status(P):
    """The per-pair 5-tuples, the mean and the out-of-range flag (raised by the LAST pair only, in its background list) through
    the few-pair finalize kernel and through the finalize + mean pair of kernels: each pair must equal a call of its own."""
    from dcn_hip import loss as K
    g = torch.Generator().manual_seed(11)
    HW, D = 400, 3
    A = (torch.rand(P, HW, D, generator=g) - 0.5) * 0.8
    B = (torch.rand(P, HW, D, generator=g) - 0.5) * 0.8
    cfg = K.make_config([1.0, 0.5, 0.5, 0.5], 20)
    lists = []
    for b in range(P):
        n = [5 + 37 * b, 3 + 11 * b, 2 + 29 * (P - b)]
        t = [torch.randint(0, HW, (k,), generator=g) for k in n for _ in range(2)]
        lists.append((t[0], t[1], t[2], t[3], t[4], t[5], None, None))
    ok = K.contrastive_loss(A, B, K.PairLists.from_lists(lists, "cpu"), cfg)
    assert int(ok[4]) == 0
    singles = [K.contrastive_loss(A[b:b + 1], B[b:b + 1], K.PairLists.from_lists([lists[b]], "cpu"), cfg) for b in range(P)]
    for b in range(P):
        np.testing.assert_allclose(ok[1][b].numpy(), singles[b][1][0].numpy(), rtol=1e-6, atol=1e-12)
        assert torch.equal(ok[3][b], singles[b][3][0])
    np.testing.assert_allclose(ok[0].item(), np.mean([s_[0].item() for s_ in singles]), rtol=1e-6)
    bad = list(lists)
    last = list(bad[-1])
    last[5] = last[5].clone()
    last[5][-1] = HW                       # out of range: skipped, flagged
    bad[-1] = tuple(last)
    flagged = K.contrastive_loss(A, B, K.PairLists.from_lists(bad, "cpu"), cfg)
    assert int(flagged[4]) == 1
    for b in range(P - 1):
        np.testing.assert_allclose(flagged[1][b].numpy(), ok[1][b].numpy(), rtol=0, atol=0)


def test_host_lists_are_range_checked_and_debug_surfaces_the_device_status():
    """an index >= H*W (lists built for another image size) must not silently train on a partial loss.  Lists
    that arrive on the host raise IndexError like the reference's index_select; for device-resident lists the kernel's
    status word is kept on the loss object and raised when `debug` is on."""
    from dcn_hip import loss as K
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    from oracle import synth
    A = torch.rand(1, 100, 3)
    B = torch.rand(1, 100, 3)
    good, bad = torch.tensor([1, 2, 3]), torch.tensor([1, 100, 3])
    e = torch.tensor([-1])
    pcl = PixelwiseContrastiveLoss([10, 10], synth.LOSS_CONFIG)
    with pytest.raises(IndexError):
        loss_composer.get_loss(pcl, 0, A, B, good, bad, e, e, e, e, e, e)
    with pytest.raises(IndexError):
        K.PairLists.from_lists([(good, torch.tensor([1, -7, 3]), None, None, None, None, None, None)], "cpu", hw=100)
    raw = K.PairLists.from_lists([(good, bad, None, None, None, None, None, None)], "cpu")   # unchecked (as if on the GPU)
    loss_composer.get_loss_batched(pcl, 0, A, B, raw)
    assert int(pcl.last_status) == 1
    pcl.debug = True
    with pytest.raises(IndexError):
        loss_composer.get_loss_batched(pcl, 0, A, B, raw)
    with pytest.raises(TypeError):
        loss_composer.get_loss_batched(pcl, 0, A.double(), B.double(), raw)


def test_mismatched_lists_raise():
    from dcn_hip import loss as K
    with pytest.raises(ValueError):
        K.PairLists.from_lists([(torch.tensor([1, 2]), torch.tensor([1]), None, None, None, None, None, None)], "cpu")


@pytest.mark.parametrize("D,mult", [(1, 1), (5, 3), (32, 2)])
def test_triplet_kernel_shapes_vs_oracle(D, mult):
    """Descriptor widths without a specialised kernel instance, multiplier 1, gradient accumulation onto repeated pixels."""
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss as PCL
    from oracle import loss_oracle
    g = torch.Generator().manual_seed(D)
    HW, Pm = 200, 17
    A0, B0 = torch.randn(1, HW, D, generator=g), torch.randn(1, HW, D, generator=g)
    ma = torch.randint(0, HW, (Pm,), generator=g)
    mb = torch.randint(0, 8, (Pm,), generator=g)                 # few distinct pixels: many atomic adds per address
    na = ma.repeat_interleave(mult)
    nb = torch.randint(0, HW, (Pm * mult,), generator=g)
    res = []
    for fn in (PCL.get_triplet_loss, loss_oracle.PixelwiseContrastiveLoss.get_triplet_loss):
        A, B = A0.clone().requires_grad_(True), B0.clone().requires_grad_(True)
        l = fn(A, B, ma, mb, na, nb, 0.1)
        l.backward()
        res.append((l.item(), A.grad, B.grad))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0])
    assert rel_err(res[0][1], res[1][1]) < 1e-5 and rel_err(res[0][2], res[1][2]) < 1e-5


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[9:-4] for p in GOLDENS])
def test_record_based_backward_equals_the_gathering_backward(path):
    """Round 4: the forward pass keeps per-pair (difference, factor) records and the backward pass reads them instead of
    gathering both descriptors again; the gradient maps are zero-filled ahead of time.  Every scattered value is the one the
    gathering backward pass computes (same expressions); the maps agree to the order of the atomic accumulation, on every
    reference golden (pixel-distance weights, inverted / legacy hinges, empty lists, D = 3 ... 16) and after a second backward
    pass through the same graph."""
    from dcn_hip import loss as K
    from dense_correspondence.loss_functions import loss_composer
    from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss
    z, cfg = load_golden_loss(path)
    grads = {}
    for mode in (False, True):
        K.SAVE_PAIR_RECORDS = K.PREFILL_GRADIENTS = mode
        try:
            A = torch.tensor(z["A"], requires_grad=True)
            B = torch.tensor(z["B"], requires_grad=True)
            pcl = PixelwiseContrastiveLoss([int(z["H"]), int(z["W"])], cfg)
            out = loss_composer.get_loss(pcl, torch.tensor([int(z["match_type"])]), A, B, *lists_from_golden(z))
            out[0].backward(retain_graph=True)
            first = (A.grad.clone(), B.grad.clone())
            A.grad = None; B.grad = None
            out[0].backward()
            # (duplicate pixel indices accumulate with atomics: the order -- and the last bit -- may differ from pass to pass)
            assert rel_err(A.grad, first[0]) < 1e-6 and rel_err(B.grad, first[1]) < 1e-6
            grads[mode] = first + (out[0].detach().clone(),)
        finally:
            K.SAVE_PAIR_RECORDS, K.PREFILL_GRADIENTS = True, False   # (the module's defaults)
    assert torch.equal(grads[False][2], grads[True][2])
    assert rel_err(grads[True][0], grads[False][0]) < 1e-6 and rel_err(grads[True][1], grads[False][1]) < 1e-6


def test_record_based_backward_large_descriptors_and_no_grad_path():
    """D > 32 (lane groups loop over the components) through the record path; a forward pass without gradient keeps no records."""
    from dcn_hip import loss as K
    g = torch.Generator().manual_seed(8)
    P, HW, D = 2, 60, 40
    lists = []
    for _ in range(P):
        lists.append(tuple(torch.randint(0, HW, (n,), generator=g) for n in (17, 17, 9, 9, 30, 30)) + (torch.tensor([-1]), torch.tensor([-1])))
    pl = K.PairLists.from_lists(lists, "cpu", hw=HW)
    cfg = K.make_config([0.0, 0.5, 0.7, 0.5], 10)
    res = {}
    for mode in (False, True):
        K.SAVE_PAIR_RECORDS = K.PREFILL_GRADIENTS = mode
        try:
            A = (torch.rand(P, HW, D, generator=torch.Generator().manual_seed(1)) * 0.2).requires_grad_(True)
            B = (torch.rand(P, HW, D, generator=torch.Generator().manual_seed(2)) * 0.2).requires_grad_(True)
            loss = K.contrastive_loss(A, B, pl, cfg)[0]
            loss.backward()
            res[mode] = (loss.detach().clone(), A.grad.clone(), B.grad.clone())
        finally:
            K.SAVE_PAIR_RECORDS, K.PREFILL_GRADIENTS = True, False   # (the module's defaults)
    assert torch.equal(res[False][0], res[True][0])
    assert rel_err(res[True][1], res[False][1]) < 1e-6 and rel_err(res[True][2], res[False][2]) < 1e-6
    with torch.no_grad():
        out = K.contrastive_loss(A.detach(), B.detach(), pl, cfg)
    assert torch.equal(out[0], res[True][0]) and not out[0].requires_grad
