"""Oracle (TEST INFRASTRUCTURE ONLY) for the per-match statistics of the quantitative evaluation, SURVEY.md section 8f
rank 1: numpy restatement of ``DenseCorrespondenceNetwork.find_best_match`` (network.py:486-525) and of the statistics
block of ``DenseCorrespondenceEvaluation.compute_descriptor_match_statistics`` (evaluation.py:1046-1100).

Pinned: tests/golden/make_eval_goldens_from_reference.py executes those very source lines of the reference on seeded
inputs (tests/golden/eval_ref.npz); tests/test_oracle.py checks this file against them.
"""
import numpy as np


def find_best_match(pixel_a, res_a, res_b):
    """network.py:486-525 -> ((u, v), best_match_diff, norm_diffs [H, W])"""
    descriptor_at_pixel = res_a[pixel_a[1], pixel_a[0]]
    norm_diffs = np.sqrt(np.sum(np.square(res_b - descriptor_at_pixel), axis=2))
    idx = np.argmin(norm_diffs)
    xy = np.unravel_index(idx, norm_diffs.shape)
    return (xy[1], xy[0]), norm_diffs[xy], norm_diffs


def match_statistics(uv_a, uv_b, res_a, res_b, mask_b):
    """evaluation.py:1046-1100 for one match; mask_b: [H, W] of 0 / 1.  Returns the quantities the reference stores."""
    uv_b_pred, best_match_diff, norm_diffs = find_best_match(uv_a, res_a, res_b)
    masked_norm_diffs = norm_diffs + (1 - mask_b) * 1e6
    xy_m = np.unravel_index(np.argmin(masked_norm_diffs), masked_norm_diffs.shape)
    best_match_diff_masked = masked_norm_diffs[xy_m]
    uv_b_pred_masked = (xy_m[1], xy_m[0])
    des_a = res_a[uv_a[1], uv_a[0], :]
    des_b_gt = res_b[uv_b[1], uv_b[0], :]
    gt = np.linalg.norm(des_a - des_b_gt)
    out = {"uv_b_pred": uv_b_pred, "norm_diff_pred": best_match_diff, "uv_b_pred_masked": uv_b_pred_masked,
           "norm_diff_pred_masked": best_match_diff_masked, "norm_diff_descriptor_ground_truth": gt,
           "pixel_match_error_l2": np.linalg.norm(np.array(uv_b) - np.array(uv_b_pred), ord=2),
           "pixel_match_error_l2_masked": np.linalg.norm(np.array(uv_b) - np.array(uv_b_pred_masked), ord=2),
           "pixel_match_error_l1": np.linalg.norm(np.array(uv_b) - np.array(uv_b_pred), ord=1)}
    for name, nd, denom in (("", norm_diffs, res_a.shape[0] * res_a.shape[1]),
                            ("_masked", masked_norm_diffs, len(np.nonzero(mask_b)[0]))):
        v, u = np.where(nd < gt)
        out["num_pixels_closer_than_ground_truth" + name] = len(u)
        out["fraction_pixels_closer_than_ground_truth" + name] = len(u) * 1.0 / denom
        out["average_l2_distance_for_false_positives" + name] = \
            0.0 if len(u) == 0 else np.average(np.sqrt((u - uv_b[0]) ** 2 + (v - uv_b[1]) ** 2))
    return out
