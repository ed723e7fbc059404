#!/usr/bin/env python3
"""Generate tests/golden/training_loop_ref.npz by executing the REFERENCE'S OWN training driver and modules.

Runs only in the authoring container (needs /root/reference).  ``tests/reference_loop_runner.py --mode reference`` imports
/root/reference/dense_correspondence/training/training.py, its network wrapper, loss and dataset modules from where they lie
(Python-2 text converted in memory, tests/reference_py3.py; nothing is copied), supplies the un-vendored ``resnet_dilated``
with the oracle backbone, and runs ``DenseCorrespondenceTraining.run()`` (7 iterations, one of them the loader's empty
sentinel) followed by ``run_from_pretrained()`` (2 more) on a tiny in-memory dataset.  Only numbers are stored: the logged
loss terms per iteration, the learning-rate schedule, the order the samples were served in, the file names the driver wrote
and per-tensor norms / leading values of the checkpoints it saved.

    python tests/golden/make_training_loop_golden_from_reference.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
KEEP = ("served", "loss_steps", "loss", "match_loss", "masked_non_match_loss", "background_non_match_loss", "learning_rate",
        "files_first", "files_resumed", "param_names", "norms_initial", "norms_first", "norms_resumed", "heads_initial",
        "heads_first", "heads_resumed", "config_hwd", "base_width", "arch", "fc_scale", "pairs", "num_samples")


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "run.npz")
        subprocess.check_call([sys.executable, os.path.join(TESTS, "reference_loop_runner.py"), "--mode", "reference",
                               "--out", out])
        z = np.load(out)
        assert str(z["network_file"]).startswith("/root/reference/") and str(z["loss_file"]).startswith("/root/reference/")
        assert str(z["training_file"]).startswith("/root/reference/")
        path = os.path.join(HERE, "training_loop_ref.npz")
        np.savez_compressed(path, **{k: z[k] for k in KEEP})
        print(path, os.path.getsize(path), "bytes; loss", z["loss"])


if __name__ == "__main__":
    main()
