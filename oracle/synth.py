"""Oracle-side (TEST INFRASTRUCTURE) seeded synthetic inputs, SURVEY.md section 8d.

* images: ``rand(B,3,H,W)`` in [0,1), then ``(x - mean) / std`` with the reference's
  ``DEFAULT_IMAGE_MEAN / DEFAULT_IMAGE_STD_DEV``
  (modules/dense_correspondence_manipulation/utils/constants.py:18-19) -- what
  ``spartan_dataset_masked.py:297-304`` hands to the network.
* index lists: int64 ``u + W*v`` (spartan_dataset_masked.py:1256-1264), drawn uniformly with
  replacement (duplicates are present, as in ``correspondence_finder.py:326-328``); the non-match
  list is split 50/50 into masked / background (training.yaml:20-21); blind lists are the
  ``[-1]`` sentinel (dense_correspondence_dataset_masked.py:209-223).
* loss config = training.yaml:51-61 verbatim.

bench.py keeps its own copy of these few constants so that the timed MI355X path never imports
``oracle``; tests/test_abi.py::test_synth_constants_consistent_with_bench checks the two agree.
"""
import torch

DEFAULT_IMAGE_MEAN = [0.5573105812072754, 0.37420374155044556, 0.37020164728164673]
DEFAULT_IMAGE_STD_DEV = [0.24336038529872894, 0.2987397611141205, 0.31875079870224]

LOSS_CONFIG = {  # config/dense_correspondence/training/training.yaml:51-61
    "M_masked": 0.5,
    "M_background": 0.5,
    "M_pixel": 50,
    "match_loss_weight": 1.0,
    "non_match_loss_weight": 1.0,
    "use_l2_pixel_loss_on_masked_non_matches": False,
    "use_l2_pixel_loss_on_background_non_matches": False,
    "scale_by_hard_negatives": True,
    "scale_by_hard_negatives_DIFFERENT_OBJECT": True,
    "alpha_triplet": 0.1,
}

# BASELINE.json configs -> (B pairs, H, W, D, P_match, P_masked, P_background, backbone)
CONFIGS = {
    1: dict(B=1, H=480, W=640, D=3, Pm=1000, Pk=500, Pg=500, backbone="Resnet34_8s"),
    2: dict(B=4, H=480, W=640, D=3, Pm=5000, Pk=2500, Pg=2500, backbone="Resnet34_8s"),
    3: dict(B=32, H=480, W=640, D=16, Pm=10000, Pk=50000, Pg=50000, backbone="Resnet34_8s"),
    4: dict(B=8, H=480, W=640, D=3, Pm=5000, Pk=2500, Pg=2500, backbone="Resnet34_8s"),   # per GPU, x8 GPUs
    5: dict(B=2, H=960, W=1280, D=32, Pm=2500, Pk=5000, Pg=5000, backbone="Resnet50_8s", masked=True),  # per GPU, x8
}


def make_images(B, H, W, gen):
    mean = torch.tensor(DEFAULT_IMAGE_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(DEFAULT_IMAGE_STD_DEV).view(1, 3, 1, 1)
    a = (torch.rand(B, 3, H, W, generator=gen) - mean) / std
    b = (torch.rand(B, 3, H, W, generator=gen) - mean) / std
    return a, b


def make_index_lists(B, HW, Pm, Pk, Pg, gen):
    """Returns a list (one per image pair) of dicts of int64 tensors."""
    out = []
    for _ in range(B):
        d = {}
        for name, n in (("matches", Pm), ("masked_non_matches", Pk), ("background_non_matches", Pg)):
            for side in ("a", "b"):
                if n > 0:
                    d[name + "_" + side] = torch.randint(0, HW, (n,), generator=gen, dtype=torch.int64)
                else:
                    d[name + "_" + side] = torch.tensor([-1], dtype=torch.int64)
        d["blind_non_matches_a"] = torch.tensor([-1], dtype=torch.int64)
        d["blind_non_matches_b"] = torch.tensor([-1], dtype=torch.int64)
        out.append(d)
    return out




def make_masked_index_lists(B, H, W, Pm, per_match, gen):
    """BASELINE config 5 "masked-background non-match sampling" (SURVEY.md section 8d): a random elliptic object mask
    (~15 % of the image) per image; matches lie on the mask in both images; for every match ``per_match`` masked
    non-matches (b index ON the mask) and ``per_match`` background non-matches (b index OFF the mask), with the a index
    repeated per match -- the grouped layout of spartan_dataset_masked.py:841-858."""
    out = []
    ys = torch.arange(H).view(H, 1).float()
    xs = torch.arange(W).view(1, W).float()
    for _ in range(B):
        masks = []
        for _side in range(2):
            cy = (0.3 + 0.4 * torch.rand(1, generator=gen)) * H
            cx = (0.3 + 0.4 * torch.rand(1, generator=gen)) * W
            ry, rx = 0.22 * H, 0.22 * W
            masks.append((((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 <= 1.0).reshape(-1))
        on_a, on_b = masks[0].nonzero().reshape(-1), masks[1].nonzero().reshape(-1)
        off_b = (~masks[1]).nonzero().reshape(-1)
        pick = lambda pool, n: pool[torch.randint(0, pool.numel(), (n,), generator=gen)]
        ma, mb = pick(on_a, Pm), pick(on_b, Pm)
        rep = ma.repeat_interleave(per_match)
        d = {"matches_a": ma, "matches_b": mb,
             "masked_non_matches_a": rep.clone(), "masked_non_matches_b": pick(on_b, Pm * per_match),
             "background_non_matches_a": rep.clone(), "background_non_matches_b": pick(off_b, Pm * per_match),
             "blind_non_matches_a": torch.tensor([-1], dtype=torch.int64),
             "blind_non_matches_b": torch.tensor([-1], dtype=torch.int64)}
        out.append(d)
    return out


def make_batch(B, H, W, Pm, Pk, Pg, seed=1, masked=False):
    gen = torch.Generator().manual_seed(seed)
    img_a, img_b = make_images(B, H, W, gen)
    if masked:
        lists = make_masked_index_lists(B, H, W, Pm, Pk // Pm, gen)
    else:
        lists = make_index_lists(B, H * W, Pm, Pk, Pg, gen)
    return img_a, img_b, lists


def make_descriptor_pair(HW, D, seed=1, scale=1.0):
    """Loss-only vectors with descriptors ~ U(-scale, scale) so the hinge is active on both sides."""
    gen = torch.Generator().manual_seed(seed)
    A = (torch.rand(1, HW, D, generator=gen) * 2 - 1) * scale
    B = (torch.rand(1, HW, D, generator=gen) * 2 - 1) * scale
    return A, B
