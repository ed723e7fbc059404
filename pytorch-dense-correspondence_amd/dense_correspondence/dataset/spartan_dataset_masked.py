"""The two names of the reference's dataset module that the loss composer needs
(``from dense_correspondence.dataset.spartan_dataset_masked import SpartanDataset, SpartanDatasetDataType``,
loss_composer.py:1).  Pair generation / image loading stay on the CPU in the reference and are out of scope
here (BASELINE.json north_star); only the output contract is mirrored.

When the reference's own module of this name is importable behind this source root (dcn_hip/_dropin.py) this
placeholder steps aside for it at import time.
"""
from dcn_hip._dropin import step_aside_for_reference as _step_aside

if not _step_aside(__name__, __file__):
    import torch


    class SpartanDatasetDataType:
        """dense_correspondence/dataset/spartan_dataset_masked.py:31-36"""
        SINGLE_OBJECT_WITHIN_SCENE = 0
        SINGLE_OBJECT_ACROSS_SCENE = 1
        DIFFERENT_OBJECT = 2
        MULTI_OBJECT = 3
        SYNTHETIC_MULTI_OBJECT = 4


    class SpartanDataset(object):
        """Only the static helpers of the sample contract (dense_correspondence_dataset_masked.py:202-223)."""

        @staticmethod
        def empty_tensor():
            return torch.LongTensor([-1])

        @staticmethod
        def is_empty(tensor):
            return (len(tensor) == 1) and bool(tensor[0] == -1)

        @staticmethod
        def flatten_uv_tensor(uv_tensor, image_width):
            # spartan_dataset_masked.py:1256-1264
            return uv_tensor[1].long() * image_width + uv_tensor[0].long()
