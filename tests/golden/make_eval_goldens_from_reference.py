"""Pins oracle/evaluation_oracle.py: executes the REFERENCE's own source lines
  dense_correspondence/network/dense_correspondence_network.py:486-525   (find_best_match)
  dense_correspondence/evaluation/evaluation.py:1045-1100               (the statistics block of compute_descriptor_match_statistics)
(read from /root/reference at run time, never copied) on seeded inputs and stores inputs + outputs in
tests/golden/eval_ref.npz.  Patches: the py2 `print "..."` debug lines of find_best_match -> print(...); the evaluation
block is dedented and run with `DenseCorrespondenceNetwork.find_best_match` bound to the function above.

    python tests/golden/make_eval_goldens_from_reference.py
"""
import os
import re
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NET = "/root/reference/dense_correspondence/network/dense_correspondence_network.py"
EVAL = "/root/reference/dense_correspondence/evaluation/evaluation.py"


def main():
    net = open(NET).read().split("\n")
    fbm = textwrap.dedent("\n".join(net[486:525]))                     # def find_best_match(...) ... return
    fbm = re.sub(r'print "([^"]*)", (\w[\w.]*)', r'print("\1", \2)', fbm)
    ns = {"np": np}
    exec(compile(fbm, NET, "exec"), ns)

    class DenseCorrespondenceNetwork(object):
        find_best_match = staticmethod(ns["find_best_match"])
    block = textwrap.dedent("\n".join(open(EVAL).read().split("\n")[1044:1100]))
    rng = np.random.RandomState(0)
    H, W, D, Q = 48, 64, 3, 12
    res_a = rng.randn(H, W, D).astype(np.float32)
    res_b = (res_a + 0.35 * rng.randn(H, W, D)).astype(np.float32)      # correlated: ground truth is usually a good match
    mask_b = np.zeros((H, W), np.float32)
    mask_b[10:40, 15:50] = 1
    uv = np.stack([rng.randint(15, 50, Q), rng.randint(10, 40, Q)], 1)   # (u, v), same pixel in a and b
    keys = ["uv_b_pred", "best_match_diff", "uv_b_pred_masked", "best_match_diff_masked", "pixel_match_error_l2",
            "pixel_match_error_l2_masked", "pixel_match_error_l1", "norm_diff_descriptor_ground_truth",
            "num_pixels_closer_than_ground_truth", "fraction_pixels_closer_than_ground_truth",
            "num_pixels_closer_than_ground_truth_masked", "fraction_pixels_closer_than_ground_truth_masked",
            "average_l2_distance_for_false_positives", "average_l2_distance_for_false_positives_masked"]
    out = {k: [] for k in keys}
    for q in range(Q):
        env = {"np": np, "DenseCorrespondenceNetwork": DenseCorrespondenceNetwork, "uv_a": (int(uv[q, 0]), int(uv[q, 1])),
               "uv_b": (int(uv[q, 0]), int(uv[q, 1])), "res_a": res_a, "res_b": res_b, "mask_b": mask_b, "debug": False}
        exec(compile(block, EVAL, "exec"), env)
        for k in keys:
            out[k].append(np.asarray(env[k], dtype=np.float64))
    np.savez_compressed(os.path.join(HERE, "eval_ref.npz"), res_a=res_a, res_b=res_b, mask_b=mask_b, uv=uv,
                        **{k: np.stack(v) for k, v in out.items()})
    print("wrote eval_ref.npz;", "closer-than-gt counts:", [int(x) for x in out["num_pixels_closer_than_ground_truth"]])


if __name__ == "__main__":
    main()
