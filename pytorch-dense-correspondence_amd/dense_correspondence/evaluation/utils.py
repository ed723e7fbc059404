"""Descriptor-image export of the reference's evaluation tooling, batched for the MI355X:

    extract_descriptor_images_for_scene          dense_correspondence/evaluation/utils.py:109-160
    compute_descriptor_images_for_single_scene   modules/dense_correspondence_manipulation/scripts/compute_descriptor_images.py:38-72

Same names, arguments, file names and file contents (one ``[H, W, D]`` float32 ``.npy`` per image, what
``dcn.forward_single_image_tensor`` returns) -- but the images go through the network ``batch_size`` at a time
(``DenseCorrespondenceNetwork.forward_image_tensors``: ONE eval-mode engine call per batch, conv + folded batch norm + ReLU fused,
see csrc/backbone_engine.hip) instead of one forward per image, and the device -> host copy of a batch overlaps the next
batch's forward.  ``dataset`` is duck-typed: ``get_pose_data(scene_name)``, ``rgb_image_to_tensor(rgb)`` and either
``get_rgb_image_from_scene_name_and_idx(scene_name, idx)`` or ``get_rgbd_mask_pose(scene_name, idx)`` -- the reference's
``SpartanDataset`` (dataset loading itself is out of this package's scope).  Every other name of the reference's module
(``PandaDataFrameWrapper`` ..., evaluation.py:34) is handed on to it when it is importable behind this source root
(dcn_hip/_dropin.py)."""
import os
import shutil
import time

import numpy as np
import torch

import dense_correspondence_manipulation.utils.utils as utils
from dcn_hip._dropin import reference_sibling as _reference_sibling

_ref = _reference_sibling(__name__, __file__)


def __getattr__(name):
    return _ref.attr(name)

PADDED_STRING_WIDTH = 6   # SpartanDataset.PADDED_STRING_WIDTH (spartan_dataset_masked.py:41)


def descriptor_image_filename(img_idx):
    """SceneStructure.descriptor_image_filename (dense_correspondence/dataset/scene_structure.py:122-124)"""
    return utils.getPaddedString(img_idx) + "_descriptor_image.npy"


def _batches(seq, n):
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def _export(dcn, image_idxs, load_tensor, filename_of, save_dir, batch_size, log_every=50, verbose=True):
    was_training = dcn.training
    dcn.eval()
    pending = None          # (host tensor [n, H, W, D] in flight, event, indices)
    done = 0
    try:
        for chunk in _batches(image_idxs, max(1, int(batch_size))):
            batch = torch.stack([load_tensor(i) for i in chunk])          # [n, 3, H, W], already normalised
            res = dcn.forward_image_tensors(batch)                        # [n, H, W, D] on the device
            host = torch.empty(res.shape, dtype=res.dtype, pin_memory=res.is_cuda)
            host.copy_(res, non_blocking=True)
            ev = None
            if res.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
            if pending is not None:
                done = _flush(pending, filename_of, save_dir, done, len(image_idxs), log_every, verbose)
            pending = (host, ev, chunk)
        if pending is not None:
            done = _flush(pending, filename_of, save_dir, done, len(image_idxs), log_every, verbose)
    finally:
        dcn.train(was_training)
    return done


def _flush(pending, filename_of, save_dir, done, total, log_every, verbose):
    host, ev, chunk = pending
    if ev is not None:
        ev.synchronize()
    for k, idx in enumerate(chunk):
        np.save(os.path.join(save_dir, filename_of(idx)), host[k].numpy())
        if verbose and (done % log_every) == 0:
            print("processing image %d of %d" % (done, total))
        done += 1
    return done


def extract_descriptor_images_for_scene(dcn, dataset, scene_name, save_dir, overwrite=False, batch_size=8):
    """evaluation/utils.py:109-160: ``<idx, 6 digits>_descriptor.npy`` for every image of the scene, in index order."""
    pose_data = dataset.get_pose_data(scene_name)
    image_idxs = sorted(pose_data.keys())
    start_time = time.time()
    if os.path.exists(save_dir):
        if not overwrite:
            raise ValueError("save_dir %s already exists and overwrite is False" % (save_dir))
        shutil.rmtree(save_dir)
    os.makedirs(save_dir)

    def load(idx):
        return dataset.rgb_image_to_tensor(dataset.get_rgb_image_from_scene_name_and_idx(scene_name, idx))
    _export(dcn, image_idxs, load, lambda i: utils.getPaddedString(i, width=PADDED_STRING_WIDTH) + "_descriptor.npy", save_dir,
            batch_size)
    print("computing descriptor images took %d seconds" % (time.time() - start_time))


def compute_descriptor_images_for_single_scene(dataset, scene_name, dcn, save_dir, batch_size=8):
    """compute_descriptor_images.py:38-72: ``<idx>_descriptor_image.npy`` (SceneStructure naming) for every image of the scene."""
    pose_data = dataset.get_pose_data(scene_name)
    if not os.path.isdir(save_dir):
        os.makedirs(save_dir)

    def load(idx):
        rgb = dataset.get_rgbd_mask_pose(scene_name, idx)[0]
        return dataset.rgb_image_to_tensor(rgb)
    return _export(dcn, list(pose_data.keys()), load, descriptor_image_filename, save_dir, batch_size, log_every=1)
