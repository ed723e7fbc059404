"""Pins oracle/correspondence_oracle.py: executes the REFERENCE's own source text of
dense_correspondence/correspondence_tools/correspondence_finder.py (read from /root/reference at run time, never copied)
on seeded synthetic scenes and stores inputs + outputs as tests/golden/corr_ref_*.npz.

In-memory patches to run the Python-2 / torch-0.4 source under Python 3 / torch 2 (nothing else is touched):
  * `print "..."` statement -> print(...)
  * the torchvision / PIL imports and `from ...constants import *` are dropped (DEPTH_IM_SCALE = 1000.0 is injected,
    constants.py:10); `utils.flattened_pixel_locations_to_u_v` (utils.py) is injected as (flat % W, flat // W)
  * LongTensor `/ image_width` (integer division in py2 / torch 0.4) -> `//`
  * `diffs_k.view(-1,1)` -> `.reshape(-1,1)` (torch 2 keeps the transposed strides of an elementwise result, torch 0.4
    returned a contiguous tensor; same values)
  * `pytorch_rand_select_pixel` is replaced by a function that returns the seeded candidate pixels (so that the function
    under test is deterministic); `torch.rand` / `torch.randn` are left alone and seeded with torch.manual_seed.

    python tests/golden/make_correspondence_goldens_from_reference.py
"""
import os
import re

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/dense_correspondence/correspondence_tools/correspondence_finder.py"


def load_reference():
    text = open(SRC).read()
    text = re.sub(r'print "([^"]*)"', r'print("\1")', text)
    text = text.replace("from PIL import Image", "").replace("from torchvision import transforms", "")
    text = text.replace("from dense_correspondence_manipulation.utils.constants import *", "DEPTH_IM_SCALE = 1000.0")
    text = text.replace("randomized_mask_b_indices_flat/image_width", "randomized_mask_b_indices_flat//image_width")
    text = text.replace("diffs_0.view(-1,1)", "diffs_0.reshape(-1,1)").replace("diffs_1.view(-1,1)", "diffs_1.reshape(-1,1)")
    ns = {"__name__": "reference_correspondence_finder"}

    class _Utils(object):
        @staticmethod
        def flattened_pixel_locations_to_u_v(flat, image_width):
            return (flat % image_width, flat // image_width)
    ns["utils"] = _Utils
    exec(compile(text, SRC, "exec"), ns)
    return ns


def pose(rx, ry, t):
    cx, sx, cy, sy = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry)
    R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]).dot(np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def scene(seed, H=480, W=640):
    """Two views of a slanted plane with a box on it: depth images by ray casting is overkill for a fixture -- the depth
    maps are smooth random surfaces (plus holes of zero depth), which exercises every pruning branch."""
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    def surf():
        d = 900 + 150 * np.sin(xs / (60 + 40 * rng.rand())) + 120 * np.cos(ys / (50 + 30 * rng.rand())) + 40 * rng.rand()
        d[(rng.rand(H, W) < 0.02)] = 0                      # no-return pixels
        d[100:140, 200:260] = 0
        return d.astype(np.uint16)
    da, db = surf(), surf()
    pa = pose(0.02 * rng.randn(), 0.02 * rng.randn(), 0.05 * rng.randn(3))
    pb = pose(0.05 * rng.randn(), 0.08 * rng.randn(), 0.08 * rng.randn(3))
    mask = np.zeros((H, W), np.float32)
    mask[150:380, 180:470] = 1.0
    return da, pa, db, pb, mask


def main():
    ns = load_reference()
    for case, (seed, n_cand, per_match, masked) in enumerate([(1, 2000, 7, False), (2, 5000, 3, True), (3, 300, 150, True)]):
        da, pa, db, pb, mask = scene(seed)
        H, W = da.shape
        g = torch.Generator().manual_seed(100 + seed)
        cu = torch.randint(0, W, (n_cand,), generator=g)
        cv = torch.randint(0, H, (n_cand,), generator=g)
        if case == 0:   # candidates on the image border: exact-zero coordinates and FOV pruning
            cu[:50] = 0
            cv[50:100] = H - 1
        original = ns["pytorch_rand_select_pixel"]
        ns["pytorch_rand_select_pixel"] = lambda width, height, num_samples=1: (cu.clone(), cv.clone())
        uv_a, uv_b = ns["batch_find_pixel_correspondences"](da, pa, db, pb, num_attempts=n_cand)
        ns["pytorch_rand_select_pixel"] = original
        out = {"depth_a": da, "pose_a": pa, "depth_b": db, "pose_b": pb, "cand_u": cu.numpy(), "cand_v": cv.numpy(),
               "uv_a_u": uv_a[0].numpy(), "uv_a_v": uv_a[1].numpy(), "uv_b_u": uv_b[0].numpy(), "uv_b_v": uv_b[1].numpy(),
               "per_match": per_match, "mask": mask if masked else np.zeros((0, 0), np.float32)}
        # non-correspondences for the matches found above; the reference draws torch.rand(n) (masked) or torch.rand(2, n)
        torch.manual_seed(200 + seed)
        nm = ns["create_non_correspondences"](uv_b, (H, W), num_non_matches_per_match=per_match,
                                              img_b_mask=torch.from_numpy(mask) if masked else None)
        out["non_u"], out["non_v"] = nm[0].numpy(), nm[1].numpy()
        torch.manual_seed(200 + seed)
        n = len(uv_b[0]) * per_match
        out["rand"] = (torch.rand(n) if masked else torch.rand(2, n)).numpy()
        path = os.path.join(HERE, "corr_ref_%d.npz" % case)
        np.savez_compressed(path, **out)
        print(path, "matches", len(uv_a[0]), "of", n_cand, "non-matches", nm[0].shape)


if __name__ == "__main__":
    main()
