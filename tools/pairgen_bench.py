"""Pair-generation throughput at the training configuration's sizes
(training.yaml:9,17-21: 10 000 matching attempts, 150 masked + 150 background non-matches per match) on one MI355X,
next to the oracle (the reference's CPU algorithm, restated) on the host.

    python tools/pairgen_bench.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-dense-correspondence_amd"))
sys.path.insert(0, ROOT)


def main():
    from dense_correspondence.correspondence_tools import correspondence_finder as cf
    z = np.load(os.path.join(ROOT, "tests", "golden", "corr_ref_1.npz"))
    H, W = z["depth_a"].shape
    mask = torch.tensor(z["mask"]).cuda()
    da = torch.from_numpy(z["depth_a"].astype(np.uint16).view(np.int16)).cuda()
    db = torch.from_numpy(z["depth_b"].astype(np.uint16).view(np.int16)).cuda()

    def one_pair():
        uv_a, uv_b = cf.batch_find_pixel_correspondences(da, z["pose_a"], db, z["pose_b"], num_attempts=10000, img_a_mask=mask)
        m = cf.create_non_correspondences(uv_b, (H, W), 150, img_b_mask=mask)
        b = cf.create_non_correspondences(uv_b, (H, W), 150, img_b_mask=1 - mask)
        return uv_a[0].numel(), m, b

    for _ in range(3):
        nmatch, _, _ = one_pair()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        one_pair()
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / reps
    # CPU: the oracle = the reference's algorithm (torch CPU ops), same sizes, this process's threads
    from oracle import correspondence_oracle as co
    mk = torch.tensor(z["mask"])
    t0 = time.perf_counter()
    creps = 5
    for _ in range(creps):
        lst = torch.nonzero(mk.reshape(-1)).squeeze(1)
        sel = lst[torch.floor(torch.rand(10000) * lst.numel()).long()]
        ua, ub = co.find_correspondences_for_candidates(z["depth_a"], z["pose_a"], z["depth_b"], z["pose_b"], sel % W, sel // W)
        n = ua[0].numel() * 150
        co.create_non_correspondences(ua[0].numel(), (H, W), 150, mk, torch.rand(n))
        co.create_non_correspondences(ua[0].numel(), (H, W), 150, 1 - mk, torch.rand(n))
    cpu_ms = 1e3 * (time.perf_counter() - t0) / creps
    samples = nmatch * 300
    print(json.dumps({"image_pairs_per_s_gpu": 1e3 / gpu_ms, "ms_per_pair_gpu": gpu_ms, "matches_per_pair": nmatch,
                      "non_match_samples_per_pair": samples,
                      "algorithmic_GB_per_s": (samples * 2 * 4 + samples * 4 + 2 * H * W * (4 + 8)) / (gpu_ms * 1e-3) / 1e9,
                      "ms_per_pair_cpu_oracle": cpu_ms, "cpu_threads": torch.get_num_threads(),
                      "note": "GPU figure includes two host syncs per call (match count, mask count) as in the mirrored API"}))


if __name__ == "__main__":
    main()
