"""``PixelwiseContrastiveLoss`` with the reference's API
(dense_correspondence/loss_functions/pixelwise_contrastive_loss.py; ``pcl.py:N`` below cites that file), computed
by the fused gfx950 gather-L2-hinge kernel (dcn_hip.loss / csrc/loss_kernels.hip).

Every public method of the reference exists with the same arguments and return tuples.  The training loop does
not call them one by one any more -- ``loss_composer.get_loss`` makes ONE fused kernel call for all lists -- but
they remain usable on their own (each is one kernel launch over a single list).
``num_hard_negatives`` is returned as a python int where the reference does (that costs one device->host sync,
exactly like ``len(torch.nonzero(...))`` at pcl.py:210-211); the fused composer path never syncs.
"""
import torch

from dcn_hip import loss as _k


def _single_list(a, b, slot):
    lists = [None] * 8
    lists[2 * slot], lists[2 * slot + 1] = a, b
    return _k.PairLists.from_lists([tuple(lists)], a.device)


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

    # ------------------------------------------------------------------ building blocks (one launch each)
    @staticmethod
    def match_loss(image_a_pred, image_b_pred, matches_a, matches_b):
        """pcl.py:132-167 -> (match_loss, matches_a_descriptors, matches_b_descriptors)"""
        lists = _single_list(matches_a, matches_b, _k.LIST_MATCH)
        cfg = _k.make_config([0, 0, 0, 0], 1, match_loss_weight=1.0, non_match_loss_weight=0.0)
        loss = _k.contrastive_loss(image_a_pred, image_b_pred, lists, cfg)[0]
        # the gathered descriptors are a convenience return value nobody on the training path reads
        da = torch.index_select(image_a_pred, 1, matches_a)
        db = torch.index_select(image_b_pred, 1, matches_b)
        if len(matches_a) == 1:
            da, db = da.unsqueeze(0), db.unsqueeze(0)
        return loss, da, db

    @staticmethod
    def non_match_descriptor_loss(image_a_pred, image_b_pred, non_matches_a, non_matches_b, M=0.5, invert=False):
        """pcl.py:171-213 -> (loss vector [P], num_hard_negatives (int), descriptors a, descriptors b)"""
        lists = _single_list(non_matches_a, non_matches_b, _k.LIST_MASKED)
        cfg = _k.make_config([0, M, 0, 0], 1, invert=(0, int(invert), 0, 0))
        vec, hard = _k.per_term_losses(image_a_pred, image_b_pred, lists, cfg)
        num_hard_negatives = int(hard[0, _k.LIST_MASKED].item())
        da = torch.index_select(image_a_pred, 1, non_matches_a).squeeze()
        db = torch.index_select(image_b_pred, 1, non_matches_b).squeeze()
        if len(non_matches_a) == 1:
            da, db = da.unsqueeze(0), db.unsqueeze(0)
        return vec, num_hard_negatives, da, db

    def non_match_loss_descriptor_only(self, image_a_pred, image_b_pred, non_matches_a, non_matches_b,
                                       M_descriptor=0.5, invert=False):
        """pcl.py:271-304 -> (sum of the loss vector, num_hard_negatives)"""
        if M_descriptor is None:
            M_descriptor = self._config["M_descriptor"]
        lists = _single_list(non_matches_a, non_matches_b, _k.LIST_BLIND)
        cfg = _k.make_config([0, 0, 0, M_descriptor], 1, invert=(0, 0, 0, int(invert)), compose=_k.COMPOSE_RAW_SUMS)
        out = _k.contrastive_loss(image_a_pred, image_b_pred, lists, cfg)
        non_match_loss, hard = out[0], out[3]
        num_non_matches = int(non_matches_a.numel())
        num_hard_negatives = int(hard[0, _k.LIST_BLIND].item())
        if self._debug:
            self._debug_data['num_hard_negatives'] = num_hard_negatives
            self._debug_data['fraction_hard_negatives'] = num_hard_negatives * 1.0 / num_non_matches
        return non_match_loss, num_hard_negatives

    def l2_pixel_loss(self, matches_b, non_matches_b, M_pixel=None):
        """pcl.py:307-334: index arithmetic only (int64 -> fp32), no descriptor traffic; plain tensor ops."""
        if M_pixel is None:
            M_pixel = self._config['M_pixel']
        num_non_matches_per_match = len(non_matches_b) // len(matches_b)
        ground_truth_pixels_for_non_matches_b = torch.t(
            matches_b.repeat(num_non_matches_per_match, 1)).contiguous().view(-1, 1)
        ground_truth_u_v_b = self.flattened_pixel_locations_to_u_v(ground_truth_pixels_for_non_matches_b)
        sampled_u_v_b = self.flattened_pixel_locations_to_u_v(non_matches_b.unsqueeze(1))
        squared_l2_pixel_loss = 1.0 / M_pixel * torch.clamp(
            (ground_truth_u_v_b - sampled_u_v_b).float().norm(2, 1), max=M_pixel)
        return squared_l2_pixel_loss, ground_truth_u_v_b, sampled_u_v_b

    def flattened_pixel_locations_to_u_v(self, flat_pixel_locations):
        # pcl.py:338-352 (integer division)
        u_v_pixel_locations = flat_pixel_locations.repeat(1, 2)
        u_v_pixel_locations[:, 0] = u_v_pixel_locations[:, 0] % self.image_width
        u_v_pixel_locations[:, 1] = u_v_pixel_locations[:, 1] // self.image_width
        return u_v_pixel_locations

    def non_match_loss_with_l2_pixel_norm(self, image_a_pred, image_b_pred, matches_b, non_matches_a, non_matches_b,
                                          M_descriptor=0.5, M_pixel=None):
        """pcl.py:215-269 -> (sum_j l_j * w_j, num_hard_negatives); the pixel weight is computed in the kernel."""
        if M_descriptor is None:
            M_descriptor = self._config["M_descriptor"]
        if M_pixel is None:
            M_pixel = self._config["M_pixel"]
        if non_matches_b.numel() % max(matches_b.numel(), 1) != 0:
            raise RuntimeError("non-matches must hold a whole number of entries per match (pcl.py:321-325)")
        lists = _k.PairLists.from_lists([(matches_b, matches_b, None, None, None, None, non_matches_a, non_matches_b)],
                                        non_matches_a.device)
        cfg = _k.make_config([0, 0, 0, M_descriptor], self.image_width, match_loss_weight=0.0,
                             compose=_k.COMPOSE_RAW_SUMS, pixel_weight=(0, 0, 0, 1), m_pixel=M_pixel)
        out = _k.contrastive_loss(image_a_pred, image_b_pred, lists, cfg)
        num_non_matches = int(non_matches_a.numel())
        non_match_loss = out[0]
        num_hard_negatives = int(out[3][0, _k.LIST_BLIND].item())
        if self.debug:
            self._debug_data['num_hard_negatives'] = num_hard_negatives
            self._debug_data['fraction_hard_negatives'] = num_hard_negatives * 1.0 / num_non_matches
        return non_match_loss, num_hard_negatives

    def get_loss_matched_and_non_matched_with_l2(self, image_a_pred, image_b_pred, matches_a, matches_b,
                                                 non_matches_a, non_matches_b, M_descriptor=None, M_pixel=None,
                                                 non_match_loss_weight=1.0, use_l2_pixel_loss=None):
        """pcl.py:35-101 -> (match_loss, non_match_loss (sum), num_hard_negatives)"""
        if M_descriptor is None:
            M_descriptor = self._config["M_descriptor"]
        if M_pixel is None:
            M_pixel = self._config["M_pixel"]
        if use_l2_pixel_loss is None:
            use_l2_pixel_loss = self._config['use_l2_pixel_loss_on_masked_non_matches']
        match_loss, _, _ = PixelwiseContrastiveLoss.match_loss(image_a_pred, image_b_pred, matches_a, matches_b)
        if use_l2_pixel_loss:
            non_match_loss, num_hard_negatives = self.non_match_loss_with_l2_pixel_norm(
                image_a_pred, image_b_pred, matches_b, non_matches_a, non_matches_b, M_descriptor=M_descriptor,
                M_pixel=M_pixel)
        else:
            non_match_loss, num_hard_negatives = self.non_match_loss_descriptor_only(
                image_a_pred, image_b_pred, non_matches_a, non_matches_b, M_descriptor=M_descriptor)
        return match_loss, non_match_loss, num_hard_negatives

    # ------------------------------------------------------------------ variants off the default path
    @staticmethod
    def get_triplet_loss(image_a_pred, image_b_pred, matches_a, matches_b, non_matches_a, non_matches_b, alpha):
        """pcl.py:104-129: ``1/P * sum max(0, (a - b_match)^2 - (a - b_nonmatch)^2 + alpha)`` with the hinge per descriptor
        component and the match list expanded ``P / P_match`` times, as ONE fused gather kernel
        (``dcn_triplet_loss_forward`` / ``_backward``).  ``matches_a`` only contributes its length, as in the reference
        (``non_matches_a`` already is the replicated ``matches_a``)."""
        if matches_a.size()[0] != matches_b.size()[0]:
            raise ValueError("matches_a / matches_b differ in length")
        return _k.triplet_loss(image_a_pred, image_b_pred, non_matches_a, matches_b, non_matches_b, alpha)

    def get_loss_original(self, image_a_pred, image_b_pred, matches_a, matches_b, non_matches_a, non_matches_b,
                          M_margin=0.5, non_match_loss_weight=1.0):
        """pcl.py:357-411, the legacy loss (no caller in the reference): ``1/P_m sum ||a - b||^2`` +
        ``w/P sum max(0, M - ||a - b||^2)`` -- the hinge acts on the SQUARED distance and is not squared again.  One fused
        kernel pass (hinge mode 2 of dcn_loss_config.invert, raw-sum composition); -> (loss, match_loss, non_match_loss)."""
        lists = _k.PairLists.from_lists([(matches_a, matches_b, non_matches_a, non_matches_b, None, None, None, None)],
                                        image_a_pred.device, hw=int(image_a_pred.shape[1]))
        pm, pn = max(lists.length(0, _k.LIST_MATCH), 1), max(lists.length(0, _k.LIST_MASKED), 1)
        cfg = _k.make_config([0.0, M_margin, 0.0, 0.0], self.image_width, match_loss_weight=1.0 / pm,
                             non_match_loss_weight=float(non_match_loss_weight) / pn, compose=_k.COMPOSE_RAW_SUMS,
                             invert=(0, 2, 0, 0))
        loss, terms, _sums, _hard, _status, _ = _k.contrastive_loss(image_a_pred, image_b_pred, lists, cfg)
        return loss, terms[0, 1] / pm, terms[0, 2] * (float(non_match_loss_weight) / pn)
