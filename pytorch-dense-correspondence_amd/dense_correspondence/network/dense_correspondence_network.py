"""``DenseCorrespondenceNetwork`` with the reference's API surface
(dense_correspondence/network/dense_correspondence_network.py; line numbers below cite that file),
running on the MI355X engine.

What is kept: constructor / ``from_config`` / ``from_model_folder`` signatures, the properties the training and
evaluation code reads, ``forward`` -> ``[N,D,H,W]``, ``process_network_output`` -> ``[N,W*H,D]``,
``forward_single_image_tensor`` -> ``[H,W,D]``, the numpy best-match helpers, ``nn.Module`` behaviour
(``parameters()``, ``state_dict()`` with ``_fcn.<backbone>.`` keys, ``.cuda()/.train()/.eval()``).

What differs by design: the descriptor map comes back in ``torch.channels_last`` memory, so the reference's own
``view(N, D, W*H).permute(0, 2, 1)`` (:317-318) is a *contiguous* [N, HW, D] tensor and each descriptor is one
4*D-byte read for the loss kernel.  Device handling: like the reference (:435) the network lives on the GPU;
CPU tensors are rejected with an error instead of silently running somewhere else.
"""
import logging
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

import dense_correspondence_manipulation.utils.utils as utils
import pytorch_segmentation_detection.models.resnet_dilated as resnet_dilated
from dcn_hip import _lib as _dcn_lib


def _device():
    """The reference hard-codes the GPU (:286, :435).  Only the test-only host-emulation build of the kernel
    library (tests/hostemu) runs on CPU tensors."""
    return torch.device("cpu") if _dcn_lib.is_hostemu() else torch.device("cuda")


def _image_to_tensor(img):
    """transforms.ToTensor() for an HxWx3 uint8 array (the only use of torchvision at :23)."""
    arr = np.asarray(img)
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


class DenseCorrespondenceNetwork(nn.Module):

    IMAGE_TO_TENSOR = valid_transform = staticmethod(_image_to_tensor)

    def __init__(self, fcn, descriptor_dimension, image_width=640, image_height=480, normalize=False):
        # :25-58
        super(DenseCorrespondenceNetwork, self).__init__()
        self._fcn = fcn
        self._descriptor_dimension = descriptor_dimension
        self._image_width = image_width
        self._image_height = image_height
        self._image_mean = np.zeros(3)
        self._image_std_dev = np.ones(3)
        self.config = dict()
        self._descriptor_image_stats = None
        self._normalize = normalize
        self._constructed_from_model_folder = False

    # ---- the attribute-style accessors the training / evaluation code reads (:61-154), generated from one table:
    # name -> (backing field, writable, mirrored into self.config and followed by a refresh of the normalisation transform)
    _ACCESSORS = {
        "fcn": ("_fcn", False, False),
        "config": ("_config", True, False),
        "descriptor_dimension": ("_descriptor_dimension", False, False),
        "image_mean": ("_image_mean", True, True),
        "image_std_dev": ("_image_std_dev", True, True),
        "image_to_tensor": ("_image_to_tensor", True, False),
        "normalize_tensor_transform": ("_normalize_tensor_transform", False, False),
        "constructed_from_model_folder": ("_constructed_from_model_folder", True, False),
    }

    @property
    def image_shape(self):
        return [self._image_height, self._image_width]

    @property
    def path_to_network_params_folder(self):
        if 'path_to_network_params_folder' not in self.config:
            raise ValueError("DenseCorrespondenceNetwork: Config doesn't have a `path_to_network_params_folder`"
                             "entry")
        return self.config['path_to_network_params_folder']

    @property
    def descriptor_image_stats(self):
        if self._descriptor_image_stats is None:
            path_to_params = utils.convert_to_absolute_path(self.path_to_network_params_folder)
            descriptor_stats_file = os.path.join(path_to_params, "descriptor_statistics.yaml")
            self._descriptor_image_stats = utils.getDictFromYamlFilename(descriptor_stats_file)
        return self._descriptor_image_stats

    @property
    def unique_identifier(self):
        # :169-193
        try:
            path_to_network_params_folder = self.path_to_network_params_folder
        except ValueError:
            return None
        identifier_file = os.path.join(path_to_network_params_folder, 'identifier.yaml')
        if not os.path.exists(identifier_file):
            return None
        if not self.constructed_from_model_folder:
            return None
        d = utils.getDictFromYamlFilename(identifier_file)
        return d['id'] + "+" + self.config['model_param_filename_tail']

    def _update_normalize_tensor_transform(self):
        mean = torch.as_tensor(np.asarray(self.image_mean, dtype=np.float32)).view(3, 1, 1)
        std = torch.as_tensor(np.asarray(self.image_std_dev, dtype=np.float32)).view(3, 1, 1)
        self._normalize_tensor_transform = lambda t: (t - mean.to(t.device)) / std.to(t.device)

    # ---- forward paths
    def forward_on_img(self, img, cuda=True):
        # :206-217 (the reference forgets to keep the .cuda() result and to add the batch dimension; the evident
        # intent -- run the network on one HxWx3 image -- is implemented)
        img_tensor = DenseCorrespondenceNetwork.IMAGE_TO_TENSOR(img).unsqueeze(0)
        if cuda:
            img_tensor = img_tensor.to(_device())
        return self.forward(img_tensor)

    def forward_on_img_tensor(self, img):
        # :220-236
        warnings.warn("use forward method instead", DeprecationWarning)
        res = self.forward_single_image_tensor(img)
        return res.detach().cpu().numpy().squeeze()

    def forward(self, img_tensor):
        """[N,3,H,W] (already mean/std normalised by the dataset) -> [N,D,H,W]   (:239-263).
        With ``normalize`` the per-pixel L2 normalisation of :256-259 is fused into the upsample kernel."""
        if self._normalize:
            return self.fcn(img_tensor, normalize=True)
        return self.fcn(img_tensor)

    def forward_pair(self, img_a, img_b):
        """``(forward(img_a), forward(img_b))`` -- the two network calls of a training step (training.py:329-333) -- as ONE
        grouped engine call: identical values (batch-norm statistics, running statistics and gradients are per image
        batch, in call order), but every kernel runs over 2N images, which fills the MI355X better at small batch.
        Not part of the reference API; ``forward`` twice remains valid."""
        return self.fcn.forward_pair(img_a, img_b, normalize=self._normalize)

    def forward_single_image_tensor(self, img_tensor):
        # :265-299
        assert len(img_tensor.shape) == 3
        img_tensor = img_tensor.unsqueeze(0)
        img_tensor = img_tensor.detach().to(device=_device())
        res = self.forward(img_tensor)  # [1,D,H,W]
        res = res.squeeze(0)
        res = res.permute(1, 2, 0)  # [H,W,D]; contiguous, because the map is channels_last
        return res

    def forward_image_tensors(self, img_tensors):
        """Batched ``forward_single_image_tensor`` (:265-299): ``[n, 3, H, W]`` normalised images -> ``[n, H, W, D]``
        descriptor images (contiguous: the map is channels_last) with ONE engine call -- what the descriptor export of
        evaluation/utils.py and compute_descriptor_images.py loops over image by image."""
        assert len(img_tensors.shape) == 4
        img_tensors = img_tensors.detach().to(device=_device())
        with torch.no_grad():
            res = self.forward(img_tensors)   # [n, D, H, W]
        return res.permute(0, 2, 3, 1)

    def process_network_output(self, image_pred, N):
        # :303-319 -- identical view/permute; zero-copy AND contiguous for channels_last input
        W = self._image_width
        H = self._image_height
        if image_pred.is_contiguous(memory_format=torch.channels_last) and not image_pred.is_contiguous():
            # same values as view(N, D, W*H).permute(0, 2, 1), expressed on the NHWC storage
            return image_pred.permute(0, 2, 3, 1).reshape(N, W * H, self.descriptor_dimension)
        image_pred = image_pred.reshape(N, self.descriptor_dimension, W * H)
        image_pred = image_pred.permute(0, 2, 1)
        return image_pred

    def clip_pixel_to_image_size_and_round(self, uv):
        # :321-332
        u = min(int(round(uv[0])), self._image_width - 1)
        v = min(int(round(uv[1])), self._image_height - 1)
        return [u, v]

    def load_training_dataset(self):
        # :333-345 -- with the reference's own dataset package importable behind this source root (dcn_hip/_dropin.py) this is
        # the reference's behaviour; the placeholder SpartanDataset of this root cannot load anything and says so
        from dense_correspondence.dataset.spartan_dataset_masked import SpartanDataset
        if not hasattr(SpartanDataset, "get_within_scene_data"):
            raise NotImplementedError("dataset loading (SpartanDataset) is outside the MI355X hot path: put the reference's "
                                      "source roots behind this one on sys.path (INTEGRATION.md) to use its dataset package")
        network_params_folder = utils.convert_to_absolute_path(self.path_to_network_params_folder)
        config = utils.getDictFromYamlFilename(os.path.join(network_params_folder, 'dataset.yaml'))
        return SpartanDataset(config_expanded=config)

    # ---- construction (:360-485)
    @staticmethod
    def get_fcn(config):
        if config["backbone"]["model_class"] == "Resnet":
            resnet_model = config["backbone"]["resnet_name"]
            if not hasattr(resnet_dilated, resnet_model):
                raise ValueError("Can't build backbone network.  Unknown resnet_name %r" % resnet_model)
            fcn = getattr(resnet_dilated, resnet_model)(num_classes=config['descriptor_dimension'])
        elif config["backbone"]["model_class"] == "Unet":
            raise ValueError("Can't build backbone network.  The Unet backbone has no MI355X implementation")
        else:
            raise ValueError("Can't build backbone network.  I don't know this backbone model class!")
        return fcn

    @staticmethod
    def from_config(config, load_stored_params=True, model_param_file=None):
        # :386-438
        if "backbone" not in config:
            config["backbone"] = dict()
            config["backbone"]["model_class"] = "Resnet"
            config["backbone"]["resnet_name"] = "Resnet34_8s"
        fcn = DenseCorrespondenceNetwork.get_fcn(config)
        normalize = config['normalize'] if 'normalize' in config else False
        dcn = DenseCorrespondenceNetwork(fcn, config['descriptor_dimension'], image_width=config['image_width'],
                                         image_height=config['image_height'], normalize=normalize)
        if load_stored_params:
            assert model_param_file is not None
            config['model_param_file'] = model_param_file
            state = torch.load(model_param_file, map_location="cpu")
            try:
                dcn.load_state_dict(state)
            except Exception:
                logging.info("loading params with the new style failed, falling back to dcn.fcn.load_state_dict")
                dcn.fcn.load_state_dict(state)
        dcn.to(_device())
        dcn.train()
        dcn.config = config
        return dcn

    @staticmethod
    def from_model_folder(model_folder, load_stored_params=True, model_param_file=None, iteration=None):
        # :441-485
        from_model_folder = False
        model_folder = utils.convert_to_absolute_path(model_folder)
        if model_param_file is None:
            model_param_file, _, _ = utils.get_model_param_file_from_directory(model_folder, iteration=iteration)
            from_model_folder = True
        model_param_file = utils.convert_to_absolute_path(model_param_file)
        training_config = utils.getDictFromYamlFilename(os.path.join(model_folder, "training.yaml"))
        config = training_config["dense_correspondence_network"]
        config["path_to_network_params_folder"] = model_folder
        config["model_param_filename_tail"] = os.path.split(model_param_file)[1]
        dcn = DenseCorrespondenceNetwork.from_config(config, load_stored_params=load_stored_params,
                                                     model_param_file=model_param_file)
        dcn.constructed_from_model_folder = from_model_folder
        dcn.model_folder = model_folder
        return dcn

    # ---- numpy best-match search (:488-550); evaluation-side helpers, kept for API completeness
    @staticmethod
    def find_best_match(pixel_a, res_a, res_b, debug=False):
        descriptor_at_pixel = res_a[pixel_a[1], pixel_a[0]]
        return DenseCorrespondenceNetwork.find_best_match_for_descriptor(descriptor_at_pixel, res_b)

    @staticmethod
    def find_best_match_for_descriptor(descriptor, res):
        norm_diffs = np.sqrt(np.sum(np.square(res - descriptor), axis=2))
        best_match_flattened_idx = np.argmin(norm_diffs)
        best_match_xy = np.unravel_index(best_match_flattened_idx, norm_diffs.shape)
        best_match_diff = norm_diffs[best_match_xy]
        best_match_uv = (best_match_xy[1], best_match_xy[0])
        return best_match_uv, best_match_diff, norm_diffs

    @staticmethod
    def find_best_matches(pixels_a, res_a, res_b, mask_b=None, return_norm_diffs=False):
        """MI355X extension of ``find_best_match`` (:488-525) for MANY query pixels at once, on device tensors:
        pixels_a [Q,2] (u,v) into res_a [H,W,D]; res_b [H,W,D].  One pass over res_b answers every query
        (the reference scans the image once per query in numpy).
        -> (best_match_uv int64 [Q,2], best_match_diff [Q], norm_diffs [Q,H,W] or None)"""
        from dcn_hip import match as _match
        pixels_a = torch.as_tensor(pixels_a, device=res_a.device).long().reshape(-1, 2)
        queries = res_a[pixels_a[:, 1], pixels_a[:, 0]]
        idx, dist, nd = _match.find_best_matches(res_b, queries, mask_b, return_norm_diffs)
        width = res_b.shape[1]
        uv = torch.stack([idx % width, idx // width], dim=1)
        return uv, dist, nd

    @staticmethod
    def compute_match_statistics(uv_a, uv_b, res_a, res_b, mask_b=None):
        """MI355X extension: the per-match statistics of the quantitative evaluation (``find_best_match`` + the block at
        evaluation.py:1046-1100) for MANY ground-truth matches of one image pair in a single pass over ``res_b``.
        uv_a, uv_b: [Q, 2] (u, v) pixel pairs; res_a, res_b: [H, W, D] device tensors; mask_b: [H, W] (non-zero = object).
        Returns a dict of length-Q device tensors named like the reference's columns."""
        from dcn_hip import match as _match
        dev = res_b.device
        uv_a = torch.as_tensor(uv_a, device=dev).long().reshape(-1, 2)
        uv_b = torch.as_tensor(uv_b, device=dev).long().reshape(-1, 2)
        height, width = int(res_b.shape[0]), int(res_b.shape[1])
        s = _match.match_statistics(res_b, res_a[uv_a[:, 1], uv_a[:, 0]], uv_b[:, 0] + width * uv_b[:, 1], mask_b)
        out = {"norm_diff_descriptor_ground_truth": s["gt_dist"]}
        ub = uv_b.float()
        n_mask = float((mask_b != 0).sum()) if mask_b is not None else float(height * width)
        for k, (name, denom) in enumerate((("", float(height * width)), ("_masked", n_mask))):
            idx = s["best_idx"][k]
            pred = torch.stack([idx % width, idx // width], dim=1)
            cnt = s["count"][k].float()
            out["uv_b_pred" + name] = pred
            out["norm_diff_pred" + name] = s["best_dist"][k]
            out["pixel_match_error_l2" + name] = (ub - pred.float()).norm(dim=1)
            out["num_pixels_closer_than_ground_truth" + name] = s["count"][k]
            out["fraction_pixels_closer_than_ground_truth" + name] = cnt / denom
            out["average_l2_distance_for_false_positives" + name] = torch.where(cnt > 0, s["dist_sum"][k] / cnt.clamp(min=1),
                                                                                torch.zeros_like(cnt))
        out["pixel_match_error_l1"] = (ub - out["uv_b_pred"].float()).abs().sum(dim=1)
        return out

    def evaluate_descriptor_at_keypoints(self, res, keypoint_list):
        raise NotImplementedError("This function is currently broken")  # :565, same as the reference


def _install_accessors(cls):
    def make(field, writable, mirrored, name):
        def getter(self):
            return getattr(self, field)

        def setter(self, value):
            object.__setattr__(self, field, value) if not isinstance(value, nn.Module) else nn.Module.__setattr__(self, field, value)
            if mirrored:
                self.config[name] = value
                self._update_normalize_tensor_transform()
        return property(getter, setter if writable else None)
    for name, (field, writable, mirrored) in cls._ACCESSORS.items():
        setattr(cls, name, make(field, writable, mirrored, name))


_install_accessors(DenseCorrespondenceNetwork)
