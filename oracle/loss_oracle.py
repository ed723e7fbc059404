"""Oracle (TEST INFRASTRUCTURE, not product): py3 / plain-torch CPU restatement of the reference's
pixelwise contrastive loss and its composer.

Follows, function for function:
  dense_correspondence/loss_functions/pixelwise_contrastive_loss.py  (cited per method below)
  dense_correspondence/loss_functions/loss_composer.py               (cited per function below)

Semantics-preserving changes only (the reference is Python 2 / torch 1.1):
  * integer ``/`` on python ints / LongTensors becomes ``//`` (pcl.py:113, :321, :351)
  * ``.cuda()`` in ``zero_loss`` (loss_composer.py:215) becomes "same device as the prediction"
  * ``long(...)`` -> ``int(...)`` (pcl.py:295)
Pinned: tests/golden/loss_ref_*.npz hold outputs of the reference's OWN source text executed in
the authoring container (tests/golden/make_loss_goldens_from_reference.py); tests/test_oracle.py
checks this restatement against them.
"""
import torch


class SpartanDatasetDataType:
    """dense_correspondence/dataset/spartan_dataset_masked.py:31-36"""
    SINGLE_OBJECT_WITHIN_SCENE = 0
    SINGLE_OBJECT_ACROSS_SCENE = 1
    DIFFERENT_OBJECT = 2
    MULTI_OBJECT = 3
    SYNTHETIC_MULTI_OBJECT = 4


def is_empty(tensor):
    """dense_correspondence/dataset/dense_correspondence_dataset_masked.py:218-223"""
    return (len(tensor) == 1) and (tensor[0] == -1)


class PixelwiseContrastiveLoss(object):
    def __init__(self, image_shape, config=None):
        # pcl.py:7-17
        self.type = "pixelwise_contrastive"
        self.image_width = image_shape[1]
        self.image_height = image_shape[0]
        assert config is not None
        self._config = config
        self._debug_data = dict()
        self._debug = False

    @property
    def debug(self):
        return self._debug

    @debug.setter
    def debug(self, value):
        self._debug = value

    @property
    def config(self):
        return self._config

    @property
    def debug_data(self):
        return self._debug_data

    def get_loss_matched_and_non_matched_with_l2(self, image_a_pred, image_b_pred, matches_a, matches_b,
                                                 non_matches_a, non_matches_b, M_descriptor=None, M_pixel=None,
                                                 non_match_loss_weight=1.0, use_l2_pixel_loss=None):
        # pcl.py:35-101
        PCL = PixelwiseContrastiveLoss
        if M_descriptor is None:
            M_descriptor = self._config["M_descriptor"]
        if M_pixel is None:
            M_pixel = self._config["M_pixel"]
        if use_l2_pixel_loss is None:
            use_l2_pixel_loss = self._config['use_l2_pixel_loss_on_masked_non_matches']
        match_loss, _, _ = PCL.match_loss(image_a_pred, image_b_pred, matches_a, matches_b)
        if use_l2_pixel_loss:
            non_match_loss, num_hard_negatives = self.non_match_loss_with_l2_pixel_norm(
                image_a_pred, image_b_pred, matches_b, non_matches_a, non_matches_b,
                M_descriptor=M_descriptor, M_pixel=M_pixel)
        else:
            non_match_loss, num_hard_negatives = self.non_match_loss_descriptor_only(
                image_a_pred, image_b_pred, non_matches_a, non_matches_b, M_descriptor=M_descriptor)
        return match_loss, non_match_loss, num_hard_negatives

    @staticmethod
    def get_triplet_loss(image_a_pred, image_b_pred, matches_a, matches_b, non_matches_a, non_matches_b, alpha):
        # pcl.py:104-129  (py2 integer division at :113)
        num_matches = matches_a.size()[0]
        num_non_matches = non_matches_a.size()[0]
        multiplier = num_non_matches // num_matches
        matches_b_long = torch.t(matches_b.repeat(multiplier, 1)).contiguous().view(-1)
        matches_a_descriptors = torch.index_select(image_a_pred, 1, non_matches_a)
        matches_b_descriptors = torch.index_select(image_b_pred, 1, matches_b_long)
        non_matches_b_descriptors = torch.index_select(image_b_pred, 1, non_matches_b)
        triplet_losses = (matches_a_descriptors - matches_b_descriptors).pow(2) \
            - (matches_a_descriptors - non_matches_b_descriptors).pow(2) + alpha
        return 1.0 / num_non_matches * torch.clamp(triplet_losses, min=0).sum()

    @staticmethod
    def match_loss(image_a_pred, image_b_pred, matches_a, matches_b):
        # pcl.py:132-167
        num_matches = matches_a.size()[0]
        matches_a_descriptors = torch.index_select(image_a_pred, 1, matches_a)
        matches_b_descriptors = torch.index_select(image_b_pred, 1, matches_b)
        if len(matches_a) == 1:
            matches_a_descriptors = matches_a_descriptors.unsqueeze(0)
            matches_b_descriptors = matches_b_descriptors.unsqueeze(0)
        match_loss = 1.0 / num_matches * (matches_a_descriptors - matches_b_descriptors).pow(2).sum()
        return match_loss, matches_a_descriptors, matches_b_descriptors

    @staticmethod
    def non_match_descriptor_loss(image_a_pred, image_b_pred, non_matches_a, non_matches_b, M=0.5, invert=False):
        # pcl.py:171-213
        non_matches_a_descriptors = torch.index_select(image_a_pred, 1, non_matches_a).squeeze()
        non_matches_b_descriptors = torch.index_select(image_b_pred, 1, non_matches_b).squeeze()
        if len(non_matches_a) == 1:
            non_matches_a_descriptors = non_matches_a_descriptors.unsqueeze(0)
            non_matches_b_descriptors = non_matches_b_descriptors.unsqueeze(0)
        norm_degree = 2
        non_match_loss = (non_matches_a_descriptors - non_matches_b_descriptors).norm(norm_degree, 1)
        if not invert:
            non_match_loss = torch.clamp(M - non_match_loss, min=0).pow(2)
        else:
            non_match_loss = torch.clamp(non_match_loss - M, min=0).pow(2)
        hard_negative_idxs = torch.nonzero(non_match_loss)
        num_hard_negatives = len(hard_negative_idxs)
        return non_match_loss, num_hard_negatives, non_matches_a_descriptors, non_matches_b_descriptors

    def non_match_loss_with_l2_pixel_norm(self, image_a_pred, image_b_pred, matches_b, non_matches_a,
                                          non_matches_b, M_descriptor=0.5, M_pixel=None):
        # pcl.py:215-269
        if M_descriptor is None:
            M_descriptor = self._config["M_descriptor"]
        if M_pixel is None:
            M_pixel = self._config["M_pixel"]
        PCL = PixelwiseContrastiveLoss
        num_non_matches = non_matches_a.size()[0]
        non_match_descriptor_loss, num_hard_negatives, _, _ = PCL.non_match_descriptor_loss(
            image_a_pred, image_b_pred, non_matches_a, non_matches_b, M=M_descriptor)
        non_match_pixel_l2_loss, _, _ = self.l2_pixel_loss(matches_b, non_matches_b, M_pixel=M_pixel)
        non_match_loss = (non_match_descriptor_loss * non_match_pixel_l2_loss).sum()
        if self.debug:
            self._debug_data['num_hard_negatives'] = num_hard_negatives
            self._debug_data['fraction_hard_negatives'] = num_hard_negatives * 1.0 / num_non_matches
        return non_match_loss, num_hard_negatives

    def non_match_loss_descriptor_only(self, image_a_pred, image_b_pred, non_matches_a, non_matches_b,
                                       M_descriptor=0.5, invert=False):
        # pcl.py:271-304
        PCL = PixelwiseContrastiveLoss
        if M_descriptor is None:
            M_descriptor = self._config["M_descriptor"]
        non_match_loss_vec, num_hard_negatives, _, _ = PCL.non_match_descriptor_loss(
            image_a_pred, image_b_pred, non_matches_a, non_matches_b, M=M_descriptor, invert=invert)
        num_non_matches = int(non_match_loss_vec.size()[0])
        non_match_loss = non_match_loss_vec.sum()
        if self._debug:
            self._debug_data['num_hard_negatives'] = num_hard_negatives
            self._debug_data['fraction_hard_negatives'] = num_hard_negatives * 1.0 / num_non_matches
        return non_match_loss, num_hard_negatives

    def l2_pixel_loss(self, matches_b, non_matches_b, M_pixel=None):
        # pcl.py:307-334  (py2 integer division at :321)
        if M_pixel is None:
            M_pixel = self._config['M_pixel']
        num_non_matches_per_match = len(non_matches_b) // len(matches_b)
        ground_truth_pixels_for_non_matches_b = torch.t(
            matches_b.repeat(num_non_matches_per_match, 1)).contiguous().view(-1, 1)
        ground_truth_u_v_b = self.flattened_pixel_locations_to_u_v(ground_truth_pixels_for_non_matches_b)
        sampled_u_v_b = self.flattened_pixel_locations_to_u_v(non_matches_b.unsqueeze(1))
        norm_degree = 2
        squared_l2_pixel_loss = 1.0 / M_pixel * torch.clamp(
            (ground_truth_u_v_b - sampled_u_v_b).float().norm(norm_degree, 1), max=M_pixel)
        return squared_l2_pixel_loss, ground_truth_u_v_b, sampled_u_v_b

    def flattened_pixel_locations_to_u_v(self, flat_pixel_locations):
        # pcl.py:338-352  (torch-1.1 LongTensor "/" is integer division, :351)
        u_v_pixel_locations = flat_pixel_locations.repeat(1, 2)
        u_v_pixel_locations[:, 0] = u_v_pixel_locations[:, 0] % self.image_width
        u_v_pixel_locations[:, 1] = u_v_pixel_locations[:, 1] // self.image_width
        return u_v_pixel_locations

    def get_loss_original(self, image_a_pred, image_b_pred, matches_a, matches_b, non_matches_a, non_matches_b,
                          M_margin=0.5, non_match_loss_weight=1.0):
        # pcl.py:357-411
        num_matches = matches_a.size()[0]
        num_non_matches = non_matches_a.size()[0]
        matches_a_descriptors = torch.index_select(image_a_pred, 1, matches_a)
        matches_b_descriptors = torch.index_select(image_b_pred, 1, matches_b)
        match_loss = 1.0 / num_matches * (matches_a_descriptors - matches_b_descriptors).pow(2).sum()
        non_matches_a_descriptors = torch.index_select(image_a_pred, 1, non_matches_a)
        non_matches_b_descriptors = torch.index_select(image_b_pred, 1, non_matches_b)
        pixel_wise_loss = (non_matches_a_descriptors - non_matches_b_descriptors).pow(2).sum(dim=2)
        pixel_wise_loss = torch.add(torch.neg(pixel_wise_loss), M_margin)
        zeros_vec = torch.zeros_like(pixel_wise_loss)
        non_match_loss = non_match_loss_weight * 1.0 / num_non_matches * torch.max(zeros_vec, pixel_wise_loss).sum()
        loss = match_loss + non_match_loss
        return loss, match_loss, non_match_loss


# ----------------------------------------------------------------------------- loss_composer.py

def zero_loss(like=None):
    # loss_composer.py:214-215 (device-agnostic)
    dev = like.device if like is not None else "cpu"
    return torch.zeros(1, dtype=torch.float32, device=dev)


def is_zero_loss(loss):
    # loss_composer.py:217-218
    return loss.item() < 1e-20


def get_loss(pixelwise_contrastive_loss, match_type, image_a_pred, image_b_pred, matches_a, matches_b,
             masked_non_matches_a, masked_non_matches_b, background_non_matches_a, background_non_matches_b,
             blind_non_matches_a, blind_non_matches_b):
    # loss_composer.py:7-67
    T = SpartanDatasetDataType
    match_type = torch.as_tensor(match_type)
    if (match_type == T.SINGLE_OBJECT_WITHIN_SCENE).all():
        return get_within_scene_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred, matches_a, matches_b,
                                     masked_non_matches_a, masked_non_matches_b, background_non_matches_a,
                                     background_non_matches_b, blind_non_matches_a, blind_non_matches_b)
    if (match_type == T.SINGLE_OBJECT_ACROSS_SCENE).all():
        return get_same_object_across_scene_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred,
                                                 blind_non_matches_a, blind_non_matches_b)
    if (match_type == T.DIFFERENT_OBJECT).all():
        return get_different_object_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred,
                                         blind_non_matches_a, blind_non_matches_b)
    if (match_type == T.MULTI_OBJECT).all() or (match_type == T.SYNTHETIC_MULTI_OBJECT).all():
        return get_within_scene_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred, matches_a, matches_b,
                                     masked_non_matches_a, masked_non_matches_b, background_non_matches_a,
                                     background_non_matches_b, blind_non_matches_a, blind_non_matches_b)
    raise ValueError("Should only have above scenes?")


def get_within_scene_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred, matches_a, matches_b,
                          masked_non_matches_a, masked_non_matches_b, background_non_matches_a,
                          background_non_matches_b, blind_non_matches_a, blind_non_matches_b):
    # loss_composer.py:70-143
    pcl = pixelwise_contrastive_loss
    match_loss, masked_non_match_loss, num_masked_hard_negatives = \
        pcl.get_loss_matched_and_non_matched_with_l2(image_a_pred, image_b_pred, matches_a, matches_b,
                                                     masked_non_matches_a, masked_non_matches_b,
                                                     M_descriptor=pcl._config["M_masked"])
    if pcl._config["use_l2_pixel_loss_on_background_non_matches"]:
        background_non_match_loss, num_background_hard_negatives = pcl.non_match_loss_with_l2_pixel_norm(
            image_a_pred, image_b_pred, matches_b, background_non_matches_a, background_non_matches_b,
            M_descriptor=pcl._config["M_background"])
    else:
        background_non_match_loss, num_background_hard_negatives = pcl.non_match_loss_descriptor_only(
            image_a_pred, image_b_pred, background_non_matches_a, background_non_matches_b,
            M_descriptor=pcl._config["M_background"])

    blind_non_match_loss = zero_loss(image_a_pred)
    num_blind_hard_negatives = 1
    if not is_empty(blind_non_matches_a.data):
        blind_non_match_loss, num_blind_hard_negatives = pcl.non_match_loss_descriptor_only(
            image_a_pred, image_b_pred, blind_non_matches_a, blind_non_matches_b,
            M_descriptor=pcl._config["M_masked"])

    total_num_hard_negatives = num_masked_hard_negatives + num_background_hard_negatives
    total_num_hard_negatives = max(total_num_hard_negatives, 1)

    if pcl._config["scale_by_hard_negatives"]:
        scale_factor = total_num_hard_negatives
        masked_non_match_loss_scaled = masked_non_match_loss * 1.0 / max(num_masked_hard_negatives, 1)
        background_non_match_loss_scaled = background_non_match_loss * 1.0 / max(num_background_hard_negatives, 1)
        blind_non_match_loss_scaled = blind_non_match_loss * 1.0 / max(num_blind_hard_negatives, 1)
    else:
        num_masked_non_matches = max(len(masked_non_matches_a), 1)
        num_background_non_matches = max(len(background_non_matches_a), 1)
        num_blind_non_matches = max(len(blind_non_matches_a), 1)
        scale_factor = num_masked_non_matches + num_background_non_matches
        masked_non_match_loss_scaled = masked_non_match_loss * 1.0 / num_masked_non_matches
        background_non_match_loss_scaled = background_non_match_loss * 1.0 / num_background_non_matches
        blind_non_match_loss_scaled = blind_non_match_loss * 1.0 / num_blind_non_matches

    non_match_loss = 1.0 / scale_factor * (masked_non_match_loss + background_non_match_loss)
    loss = pcl._config["match_loss_weight"] * match_loss + pcl._config["non_match_loss_weight"] * non_match_loss
    return loss, match_loss, masked_non_match_loss_scaled, background_non_match_loss_scaled, \
        blind_non_match_loss_scaled


def get_within_scene_loss_triplet(pixelwise_contrastive_loss, image_a_pred, image_b_pred, matches_a, matches_b,
                                  masked_non_matches_a, masked_non_matches_b, background_non_matches_a,
                                  background_non_matches_b, blind_non_matches_a, blind_non_matches_b):
    # loss_composer.py:145-166
    pcl = pixelwise_contrastive_loss
    masked = pcl.get_triplet_loss(image_a_pred, image_b_pred, matches_a, matches_b, masked_non_matches_a,
                                  masked_non_matches_b, pcl._config["alpha_triplet"])
    background = pcl.get_triplet_loss(image_a_pred, image_b_pred, matches_a, matches_b, background_non_matches_a,
                                      background_non_matches_b, pcl._config["alpha_triplet"])
    z = zero_loss(image_a_pred)
    return masked + background, z, z, z, z


def get_different_object_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred,
                              blind_non_matches_a, blind_non_matches_b):
    # loss_composer.py:168-191
    pcl = pixelwise_contrastive_loss
    scale_by_hard_negatives = pcl.config["scale_by_hard_negatives_DIFFERENT_OBJECT"]
    blind_non_match_loss = zero_loss(image_a_pred)
    if not is_empty(blind_non_matches_a.data):
        M_descriptor = pcl.config["M_background"]
        blind_non_match_loss, num_hard_negatives = pcl.non_match_loss_descriptor_only(
            image_a_pred, image_b_pred, blind_non_matches_a, blind_non_matches_b, M_descriptor=M_descriptor)
        if scale_by_hard_negatives:
            scale_factor = max(num_hard_negatives, 1)
        else:
            scale_factor = max(len(blind_non_matches_a), 1)
        blind_non_match_loss = 1.0 / scale_factor * blind_non_match_loss
    loss = blind_non_match_loss
    z = zero_loss(image_a_pred)
    return loss, z, z, z, blind_non_match_loss


def get_same_object_across_scene_loss(pixelwise_contrastive_loss, image_a_pred, image_b_pred,
                                      blind_non_matches_a, blind_non_matches_b):
    # loss_composer.py:193-212.  The in-tree function is broken (``pcl`` undefined at :203,
    # ``num_hard_negatives`` unbound when the list is empty at :205-206); restated with the
    # evident intent: pcl == the first argument, empty list -> zero loss with scale 1.
    pcl = pixelwise_contrastive_loss
    blind_non_match_loss = zero_loss(image_a_pred)
    num_hard_negatives = 1
    if not is_empty(blind_non_matches_a.data):
        blind_non_match_loss, num_hard_negatives = pcl.non_match_loss_descriptor_only(
            image_a_pred, image_b_pred, blind_non_matches_a, blind_non_matches_b,
            M_descriptor=pcl._config["M_masked"], invert=True)
    if pcl._config["scale_by_hard_negatives"]:
        scale_factor = max(num_hard_negatives, 1)
    else:
        scale_factor = max(len(blind_non_matches_a), 1)
    loss = 1.0 / scale_factor * blind_non_match_loss
    z = zero_loss(image_a_pred)
    return loss, z, z, z, blind_non_match_loss
