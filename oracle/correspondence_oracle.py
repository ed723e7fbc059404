"""Oracle (TEST INFRASTRUCTURE ONLY) for the pair-generation path, SURVEY.md section 8f rank 2: a py3 / current-torch
restatement of ``dense_correspondence/correspondence_tools/correspondence_finder.py``

* ``batch_find_pixel_correspondences`` (:409-619), from the point where the candidate pixels ``uv_a`` are known
  (:486 onwards; the random choice of candidates, :459-483, is the caller's -- it is passed in, so the function is a
  deterministic geometric filter: unproject with the depth of image a, move through both camera poses, project into
  image b, prune zero depth / out of view / occluded).
* ``create_non_correspondences`` (:276-405) with the uniform random numbers passed in.  The reference builds its
  "too close to a match" indicator from ``ones = torch.zeros_like(...)`` (:343), so the indicator is identically zero
  and the perturbation (:353-363) and the wrap-around (:369-401) never change anything; the restatement keeps the
  observable behaviour: the non-matches ARE the random samples, as float (u, v) of shape [matches, per_match].

Pinned: tests/golden/make_correspondence_goldens_from_reference.py executes the reference's own source text on seeded
inputs; tests/test_oracle.py checks this file against those outputs.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of a benchmark may import this module.
"""
import numpy as np
import torch

DEPTH_IM_SCALE = 1000.0   # modules/dense_correspondence_manipulation/utils/constants.py:10


def get_default_K_matrix():   # :36-43
    K = np.zeros((3, 3))
    K[0, 0] = 533.6422696034836
    K[1, 1] = 534.7824445233571
    K[0, 2] = 319.4091030774892
    K[1, 2] = 236.4374299691866
    K[2, 2] = 1.0
    return K


def invert_transform(transform4):   # :52-60 (rigid inverse: R^T, -R^T t; the reference gets there through aliased views)
    t4 = np.copy(transform4)
    Rt = np.transpose(transform4[0:3, 0:3]).copy()
    t4[0:3, 0:3] = Rt
    t4[0:3, 3] = -1.0 * Rt.dot(transform4[0:3, 3])
    return t4


def _apply(vec3, transform4):   # apply_transform_torch, :62-66
    vec4 = torch.cat((vec3, torch.ones_like(vec3[0, :]).unsqueeze(0)), 0)
    return transform4.mm(vec4)[0:3]


def find_correspondences_for_candidates(img_a_depth, img_a_pose, img_b_depth, img_b_pose, u_a, v_a, K=None):
    """:486-619.  img_*_depth: HxW uint16 (numpy), poses 4x4 float64, u_a / v_a: int64 tensors of candidate pixels.
    -> ((u_a, v_a) int64, (u_b, v_b) float32), or (None, None)."""
    H, W = img_a_depth.shape
    if K is None:
        K = get_default_K_matrix()
    K_inv = np.linalg.inv(K)
    flat = v_a * W + u_a
    depth_a = torch.from_numpy(img_a_depth.astype(np.float32)).view(-1, 1)
    depth_vec = (torch.index_select(depth_a, 0, flat) * 1.0 / DEPTH_IM_SCALE).squeeze(1)
    nz = torch.nonzero(depth_vec).squeeze(1)                     # case 1: no depth return
    if nz.numel() == 0:
        return None, None
    depth_vec = depth_vec[nz]
    u_p, v_p = u_a[nz], v_a[nz]
    full = torch.stack((u_p.float() * depth_vec, v_p.float() * depth_vec, depth_vec))
    cam = torch.from_numpy(K_inv).float().mm(full)
    world = _apply(cam, torch.from_numpy(img_a_pose).float())
    cam2 = _apply(world, torch.from_numpy(invert_transform(img_b_pose)).float())
    vec2 = torch.from_numpy(K).float().mm(cam2)
    u2, v2, z2 = vec2[0] / vec2[2], vec2[1] / vec2[2], vec2[2]
    eps = 1e-3
    for axis, bound in ((0, W * 1.0 - eps), (1, H * 1.0 - eps)):   # case 2: outside the field of view (u, then v)
        t = u2 if axis == 0 else v2
        t = torch.where(t < 0.0, torch.zeros_like(t), t)
        t = torch.where(t > bound, torch.zeros_like(t), t)
        keep = torch.nonzero(t).squeeze(1)                       # (an exact 0.0 coordinate is dropped too, as in the reference)
        if keep.numel() == 0:
            return None, None
        u2, v2, z2, u_p, v_p = u2[keep], v2[keep], z2[keep], u_p[keep], v_p[keep]
    depth_b = torch.from_numpy(img_b_depth.astype(np.float32)).view(-1, 1)
    flat_b = v2.long() * W + u2.long()
    depth2 = (torch.index_select(depth_b, 0, flat_b) * 1.0 / 1000).squeeze(1)
    z2 = z2 - 0.003                                              # occlusion margin
    depth2 = torch.where(depth2 < 0.0, torch.zeros_like(depth2), depth2)
    depth2 = torch.where(depth2 < z2, torch.zeros_like(depth2), depth2)   # case 3: occluded or no return in b
    keep = torch.nonzero(depth2).squeeze(1)
    if keep.numel() == 0:
        return None, None
    return (u_p[keep], v_p[keep]), (u2[keep], v2[keep])


def sample_pixels(width, height, rand2):
    """pytorch_rand_select_pixel, :29-34, with its ``torch.rand(2, n)`` passed in."""
    return torch.floor(rand2[0] * width).long(), torch.floor(rand2[1] * height).long()


def create_non_correspondences(num_matches, img_b_shape, num_non_matches_per_match, img_b_mask, rand):
    """:276-405.  ``rand``: the uniform numbers the reference draws first -- ``torch.rand(n)`` when a non-empty mask is
    given, ``torch.rand(2, n)`` otherwise (n = num_matches * per_match).  -> (u, v) float32 [num_matches, per_match]."""
    H, W = img_b_shape
    n = num_matches * num_non_matches_per_match
    idx = torch.nonzero(img_b_mask.reshape(-1)).squeeze(1) if img_b_mask is not None else None
    if idx is not None and idx.numel() > 0:
        sel = idx[torch.floor(rand * idx.numel()).long()]
        u, v = sel % W, sel // W
    else:
        u, v = sample_pixels(W, H, rand)
    return u.float().view(num_matches, num_non_matches_per_match), v.float().view(num_matches, num_non_matches_per_match)
