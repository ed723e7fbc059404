"""MI355X mirror of the two sampling functions of
``dense_correspondence/correspondence_tools/correspondence_finder.py``, same names and
argument meaning, for DEVICE-resident inputs: the reference runs them on the CPU in a 5-worker loader
(``device='CPU'``, :22-27), which cannot feed hundreds of images per second.  There is no CPU path here: a CPU request
(``device='CPU'``, or host tensors handed to ``create_non_correspondences``) and every other name of the reference's module
go to the reference's own ``correspondence_finder`` when it is importable behind this source root (dcn_hip/_dropin.py) --
its ``SpartanDataset`` keeps sampling on the CPU exactly as before -- and raise otherwise.
"""
import numpy as np
import torch

from dcn_hip import pairgen as _pg
from dcn_hip._dropin import reference_sibling as _reference_sibling

_ref = _reference_sibling(__name__, __file__)


def __getattr__(name):
    return _ref.attr(name)


def get_default_K_matrix():
    """:36-43"""
    K = np.zeros((3, 3))
    K[0, 0] = 533.6422696034836  # focal x
    K[1, 1] = 534.7824445233571  # focal y
    K[0, 2] = 319.4091030774892  # principal point x
    K[1, 2] = 236.4374299691866  # principal point y
    K[2, 2] = 1.0
    return K


def _device_depth(d, dev):
    if isinstance(d, np.ndarray):
        d = torch.from_numpy(d.astype(np.uint16).view(np.int16))   # same bits; torch has no general uint16 support
    if d.element_size() != 2:
        raise ValueError("depth images are 16-bit millimetre maps")
    return d.to(dev)


def batch_find_pixel_correspondences(img_a_depth, img_a_pose, img_b_depth, img_b_pose, uv_a=None, num_attempts=20,
                                     device='GPU', img_a_mask=None, K=None):
    """:409-619.  Returns ``(uv_a, uv_b)``: ``uv_a`` a tuple of int64 device tensors, ``uv_b`` of float32 device tensors,
    or ``(None, None)`` when nothing survives.  ``uv_a``: optional ``(u, v)`` tensors of candidate pixels (the reference
    only accepts one pixel there; here any number)."""
    if device != 'GPU':
        ref = _ref.get()
        if ref is None:
            raise ValueError("this module only implements device='GPU'; the CPU sampler is the reference's own (%s)"
                             % _ref.why_not())
        return ref.batch_find_pixel_correspondences(img_a_depth, img_a_pose, img_b_depth, img_b_pose, uv_a=uv_a,
                                                    num_attempts=num_attempts, device=device, img_a_mask=img_a_mask, K=K)
    dev = torch.device("cuda")
    da, db = _device_depth(img_a_depth, dev), _device_depth(img_b_depth, dev)
    h, w = int(da.shape[0]), int(da.shape[1])
    if uv_a is not None:
        cu = torch.as_tensor(uv_a[0], device=dev).long().reshape(-1)
        cv = torch.as_tensor(uv_a[1], device=dev).long().reshape(-1)
    elif img_a_mask is None:
        u, v = _pg.sample_pixels(torch.rand(2, num_attempts, device=dev), num_attempts, w, h)       # :459-460
        cu, cv = u.long(), v.long()
    else:
        mask = img_a_mask.to(dev) if torch.is_tensor(img_a_mask) else torch.as_tensor(np.asarray(img_a_mask), device=dev)
        lst, cnt = _pg.mask_nonzero(mask)
        if int(cnt.item()) == 0:
            return (None, None)                                                                      # :472-473
        u, v = _pg.sample_pixels(torch.rand(num_attempts, device=dev), num_attempts, w, h, lst, cnt)  # :92-121
        cu, cv = u.long(), v.long()
    if K is None:
        K = get_default_K_matrix()
    ua, va, ub, vb = _pg.find_correspondences(da, db, K, img_a_pose, img_b_pose, cu, cv)
    if ua.numel() == 0:
        return (None, None)
    return (ua, va), (ub, vb)


def create_non_correspondences(uv_b_matches, img_b_shape, num_non_matches_per_match=100, img_b_mask=None):
    """:276-405.  ``(u, v)`` float32 device tensors of shape [num_matches, per_match]: uniform samples of image b, or of
    the non-zero pixels of ``img_b_mask``.  (The reference then means to push samples that fall within one pixel of the
    match away, but builds the indicator from ``zeros_like`` (:343), so nothing is ever moved; the wrap-around that
    follows is a no-op for in-range samples.  The observable result -- the plain samples -- is what is returned.)"""
    if uv_b_matches is None:
        return None
    if torch.is_tensor(uv_b_matches[0]) and not uv_b_matches[0].is_cuda and _ref.get() is not None:
        # host tensors: the reference's CPU loader calling its own sampler (spartan_dataset_masked.py:672-695)
        return _ref.get().create_non_correspondences(uv_b_matches, img_b_shape,
                                                     num_non_matches_per_match=num_non_matches_per_match, img_b_mask=img_b_mask)
    h, w = int(img_b_shape[0]), int(img_b_shape[1])
    num_matches = len(uv_b_matches[0])
    n = num_matches * num_non_matches_per_match
    dev = torch.device("cuda")
    u = v = None
    if img_b_mask is not None:
        lst, cnt = _pg.mask_nonzero(torch.as_tensor(img_b_mask, device=dev))
        if int(cnt.item()) > 0:
            u, v = _pg.sample_pixels(torch.rand(n, device=dev), n, w, h, lst, cnt)
    if u is None:                                                           # no mask, or an empty one (:313-316, :325)
        u, v = _pg.sample_pixels(torch.rand(2, n, device=dev), n, w, h)
    return u.view(num_matches, num_non_matches_per_match), v.view(num_matches, num_non_matches_per_match)
