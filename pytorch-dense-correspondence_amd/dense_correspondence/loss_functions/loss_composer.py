"""``loss_composer`` with the reference's API (dense_correspondence/loss_functions/loss_composer.py; ``:N`` below
cites that file).  ``get_loss`` takes the same 12 positional arguments as training.py:336-342 passes and returns
the same 5-tuple of tensors -- but the whole composition (match term, masked / background / blind non-match
hinges, hard-negative counting and scaling) is ONE fused gfx950 kernel pass (dcn_hip.loss); the hard-negative
counts stay on the device, so unlike :107-117 nothing here forces a device->host sync.

``get_loss_batched`` is the B > 1 extension (SURVEY.md section 8a note B): every image pair keeps its own lists
and hard-negative normaliser, ``loss = mean_b loss_b``; for B == 1 it is exactly ``get_loss``.
"""
import torch

from dense_correspondence.dataset.spartan_dataset_masked import SpartanDataset, SpartanDatasetDataType
from dense_correspondence.loss_functions.pixelwise_contrastive_loss import PixelwiseContrastiveLoss  # noqa: F401
from dcn_hip import loss as _k


def _match_type_code(match_type):
    """``(match_type == X).all()`` of :24-58 without a device round trip when the type arrives as python / CPU data."""
    if torch.is_tensor(match_type):
        vals = match_type.reshape(-1).tolist()  # the DataLoader hands a tiny CPU tensor
        if len(set(vals)) != 1:
            raise ValueError("Should only have above scenes?")
        return int(vals[0])
    return int(match_type)


def _kernel_config(pcl, code):
    c = pcl._config
    T = SpartanDatasetDataType
    if code in (T.SINGLE_OBJECT_WITHIN_SCENE, T.MULTI_OBJECT, T.SYNTHETIC_MULTI_OBJECT):
        # :70-143 -- masked uses M_masked, background M_background, blind (logged only) M_masked
        return _k.make_config(
            [0.0, c["M_masked"], c["M_background"], c["M_masked"]], pcl.image_width,
            match_loss_weight=c["match_loss_weight"], non_match_loss_weight=c["non_match_loss_weight"],
            scale_by_hard_negatives=c["scale_by_hard_negatives"], compose=_k.COMPOSE_WITHIN_SCENE,
            pixel_weight=(0, c["use_l2_pixel_loss_on_masked_non_matches"],
                          c["use_l2_pixel_loss_on_background_non_matches"], 0),
            m_pixel=c["M_pixel"])
    if code == T.DIFFERENT_OBJECT:
        # :168-191
        return _k.make_config([0.0, 0.0, 0.0, c["M_background"]], pcl.image_width,
                              scale_by_hard_negatives=c["scale_by_hard_negatives_DIFFERENT_OBJECT"],
                              compose=_k.COMPOSE_DIFFERENT_OBJECT)
    if code == T.SINGLE_OBJECT_ACROSS_SCENE:
        # :193-212 (in-tree version references an undefined `pcl`; the evident intent is implemented)
        return _k.make_config([0.0, 0.0, 0.0, c["M_masked"]], pcl.image_width, invert=(0, 0, 0, 1),
                              scale_by_hard_negatives=c["scale_by_hard_negatives"],
                              compose=_k.COMPOSE_ACROSS_SCENE)
    raise ValueError("Should only have above scenes?")


def _check_pixel_layout(pcl, lists):
    c = pcl._config
    for flag, t in (("use_l2_pixel_loss_on_masked_non_matches", _k.LIST_MASKED),
                    ("use_l2_pixel_loss_on_background_non_matches", _k.LIST_BACKGROUND)):
        if c.get(flag):
            for p in range(lists.num_pairs):
                pm = lists.length(p, _k.LIST_MATCH)
                if pm == 0 or lists.length(p, t) % pm != 0:
                    raise RuntimeError("pixel-distance weighting needs a whole number of non-matches per match "
                                       "(pixelwise_contrastive_loss.py:321-325)")


def get_loss_batched(pixelwise_contrastive_loss, match_type, image_a_pred, image_b_pred, pair_lists):
    """image_*_pred: [B, W*H, D]; pair_lists: ``dcn_hip.loss.PairLists`` for the B pairs (or a sequence of 8-tuples).
    Returns (loss, terms [B,5], hard_negatives int32 [B,4]) -- all device tensors, no sync."""
    if not isinstance(pair_lists, _k.PairLists):
        pair_lists = _k.PairLists.from_lists(pair_lists, image_a_pred.device, hw=int(image_a_pred.shape[1]))
    cfg = _kernel_config(pixelwise_contrastive_loss, _match_type_code(match_type))
    if cfg.compose == _k.COMPOSE_WITHIN_SCENE:
        _check_pixel_layout(pixelwise_contrastive_loss, pair_lists)
    loss, terms, sums, hard, status, _ = _k.contrastive_loss(image_a_pred, image_b_pred, pair_lists, cfg)
    # status (device int32): 1 if a device-resident list held an index outside [0, H*W) -- the kernel skips such pairs where
    # the reference's index_select raises.  Kept on the loss object (no sync on the hot path); with `debug` on it is read
    # and raised immediately.
    pixelwise_contrastive_loss.last_status = status
    if getattr(pixelwise_contrastive_loss, "debug", False) and int(status.item()) != 0:
        raise IndexError("pixel index outside [0, %d) in a pair list (pixelwise_contrastive_loss.debug check)"
                         % int(image_a_pred.shape[1]))
    return loss, terms, hard


def get_loss(pixelwise_contrastive_loss, match_type, image_a_pred, image_b_pred, matches_a, matches_b,
             masked_non_matches_a, masked_non_matches_b, background_non_matches_a, background_non_matches_b,
             blind_non_matches_a, blind_non_matches_b):
    """:7-67.  -> (loss, match_loss, masked_non_match_loss, background_non_match_loss, blind_non_match_loss)"""
    lists = _k.PairLists.from_lists([(matches_a, matches_b, masked_non_matches_a, masked_non_matches_b,
                                      background_non_matches_a, background_non_matches_b,
                                      blind_non_matches_a, blind_non_matches_b)], image_a_pred.device,
                                    hw=int(image_a_pred.shape[1]))
    loss, terms, _hard = get_loss_batched(pixelwise_contrastive_loss, match_type, image_a_pred, image_b_pred, lists)
    t = terms[0]
    return loss, t[1], t[2], t[3], t[4]


def get_within_scene_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred, matches_a, matches_b,
                          masked_non_matches_a, masked_non_matches_b, background_non_matches_a,
                          background_non_matches_b, blind_non_matches_a, blind_non_matches_b):
    """:70-143"""
    return get_loss(pixelwise_contrastive_loss, SpartanDatasetDataType.SINGLE_OBJECT_WITHIN_SCENE, image_a_pred,
                    image_b_pred, matches_a, matches_b, masked_non_matches_a, masked_non_matches_b,
                    background_non_matches_a, background_non_matches_b, blind_non_matches_a, blind_non_matches_b)


def get_within_scene_loss_triplet(pixelwise_contrastive_loss, image_a_pred, image_b_pred, matches_a, matches_b,
                                  masked_non_matches_a, masked_non_matches_b, background_non_matches_a,
                                  background_non_matches_b, blind_non_matches_a, blind_non_matches_b):
    """:145-166 (no caller in the reference)"""
    pcl = pixelwise_contrastive_loss
    masked = pcl.get_triplet_loss(image_a_pred, image_b_pred, matches_a, matches_b, masked_non_matches_a,
                                  masked_non_matches_b, pcl._config["alpha_triplet"])
    background = pcl.get_triplet_loss(image_a_pred, image_b_pred, matches_a, matches_b, background_non_matches_a,
                                      background_non_matches_b, pcl._config["alpha_triplet"])
    z = zero_loss(image_a_pred)
    return masked + background, z, z, z, z


def get_different_object_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred, blind_non_matches_a,
                              blind_non_matches_b):
    """:168-191"""
    e = SpartanDataset.empty_tensor()
    return get_loss(pixelwise_contrastive_loss, SpartanDatasetDataType.DIFFERENT_OBJECT, image_a_pred, image_b_pred,
                    e, e, e, e, e, e, blind_non_matches_a, blind_non_matches_b)


def get_same_object_across_scene_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred, blind_non_matches_a,
                                      blind_non_matches_b):
    """:193-212"""
    e = SpartanDataset.empty_tensor()
    return get_loss(pixelwise_contrastive_loss, SpartanDatasetDataType.SINGLE_OBJECT_ACROSS_SCENE, image_a_pred,
                    image_b_pred, e, e, e, e, e, e, blind_non_matches_a, blind_non_matches_b)


def zero_loss(like=None):
    """:214-215 -- a [1] float zero on the GPU (on the device of ``like`` when given)."""
    if like is not None:
        return torch.zeros(1, dtype=torch.float32, device=like.device)
    return torch.zeros(1, dtype=torch.float32, device="cuda")


def is_zero_loss(loss):
    """:217-218"""
    return loss.item() < 1e-20
