"""Pair-generation kernels (host-emulated, through the C ABI) against the reference's golden outputs and the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import use_emulation_library

CORR_GOLDENS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "corr_ref_*.npz")))


@pytest.fixture(scope="module", autouse=True)
def _lib():
    return use_emulation_library()


def _depth(a):
    return torch.from_numpy(a.astype(np.uint16).view(np.int16))


@pytest.mark.parametrize("path", CORR_GOLDENS, ids=[os.path.basename(p)[:-4] for p in CORR_GOLDENS])
def test_find_correspondences_matches_reference_golden(path):
    from dcn_hip import pairgen
    from oracle import correspondence_oracle as co
    z = np.load(path)
    ua, va, ub, vb = pairgen.find_correspondences(_depth(z["depth_a"]), _depth(z["depth_b"]), co.get_default_K_matrix(),
                                                  z["pose_a"], z["pose_b"], torch.tensor(z["cand_u"]), torch.tensor(z["cand_v"]))
    # the surviving candidates, in order: exact; projected sub-pixel coordinates: fp32 round-off of a 4-matrix chain on
    # values up to 640 (ulp 6e-5) -- 3e-4 pixel absolute
    assert np.array_equal(ua.numpy(), z["uv_a_u"]) and np.array_equal(va.numpy(), z["uv_a_v"])
    np.testing.assert_allclose(ub.numpy(), z["uv_b_u"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(vb.numpy(), z["uv_b_v"], rtol=0, atol=3e-4)


@pytest.mark.parametrize("path", CORR_GOLDENS, ids=[os.path.basename(p)[:-4] for p in CORR_GOLDENS])
def test_non_correspondence_sampling_matches_reference_golden(path):
    from dcn_hip import pairgen
    z = np.load(path)
    H, W = z["depth_a"].shape
    n = z["non_u"].size
    rand = torch.tensor(z["rand"])
    if z["mask"].size:
        lst, cnt = pairgen.mask_nonzero(torch.tensor(z["mask"]))
        ref_list = np.flatnonzero(z["mask"].reshape(-1))
        assert int(cnt) == ref_list.size and np.array_equal(lst[:int(cnt)].numpy(), ref_list)
        u, v = pairgen.sample_pixels(rand, n, W, H, lst, cnt)
    else:
        u, v = pairgen.sample_pixels(rand, n, W, H)
    assert np.array_equal(u.view(z["non_u"].shape).numpy(), z["non_u"])
    assert np.array_equal(v.view(z["non_v"].shape).numpy(), z["non_v"])


def test_edge_cases():
    from dcn_hip import pairgen
    from oracle import correspondence_oracle as co
    z = np.load(CORR_GOLDENS[0])
    # no candidate survives: zero depth everywhere
    zero = torch.zeros(480, 640, dtype=torch.int16)
    out = pairgen.find_correspondences(zero, zero, co.get_default_K_matrix(), z["pose_a"], z["pose_b"],
                                       torch.tensor(z["cand_u"][:100]), torch.tensor(z["cand_v"][:100]))
    assert all(t.numel() == 0 for t in out)
    # empty mask / full mask / a single candidate
    lst, cnt = pairgen.mask_nonzero(torch.zeros(7, 9))
    assert int(cnt) == 0
    lst, cnt = pairgen.mask_nonzero(torch.ones(33, 65))
    assert int(cnt) == 33 * 65 and np.array_equal(lst.numpy(), np.arange(33 * 65))
    one = pairgen.find_correspondences(_depth(z["depth_a"]), _depth(z["depth_b"]), co.get_default_K_matrix(), z["pose_a"],
                                       z["pose_b"], torch.tensor(z["uv_a_u"][:1]), torch.tensor(z["uv_a_v"][:1]))
    assert one[0].numel() == 1 and int(one[0]) == int(z["uv_a_u"][0])


def test_out_of_image_candidates_and_mirrored_api_shapes():
    """Candidates outside the image are dropped (the reference would index out of bounds); the mirrored
    correspondence_finder API returns the reference's tuple-of-tensors shapes."""
    from dcn_hip import pairgen
    from oracle import correspondence_oracle as co
    z = np.load(CORR_GOLDENS[0])
    H, W = z["depth_a"].shape
    good_u, good_v = torch.tensor(z["uv_a_u"][:5]), torch.tensor(z["uv_a_v"][:5])
    cu = torch.cat([torch.tensor([-1, W, 3, 7]), good_u])
    cv = torch.cat([torch.tensor([5, 5, -2, H]), good_v])
    ua, va, ub, vb = pairgen.find_correspondences(_depth(z["depth_a"]), _depth(z["depth_b"]), co.get_default_K_matrix(),
                                                  z["pose_a"], z["pose_b"], cu, cv)
    assert np.array_equal(ua.numpy(), good_u.numpy()) and np.array_equal(va.numpy(), good_v.numpy())
    np.testing.assert_allclose(ub.numpy(), z["uv_b_u"][:5], rtol=0, atol=3e-4)
    # uniform sampling stays inside the image for random numbers arbitrarily close to 1
    r = torch.tensor([[0.0, 0.5, 0.99999994], [0.99999994, 0.0, 0.5]])
    u, v = pairgen.sample_pixels(r, 3, W, H)
    assert u.tolist() == [0.0, W // 2, W - 1] and v.tolist() == [H - 1, 0.0, H // 2]
