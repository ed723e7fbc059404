"""Import-time shim for ``pytorch_segmentation_detection.transforms`` (training.py:30-36,
dense_correspondence_dataset_masked.py:19).  The reference imports seven joint image / annotation transforms of the
segmentation toolbox at module load but the dense-correspondence dataset never instantiates them (its own augmentation
lives in ``correspondence_tools/correspondence_augmentation.py``); data loading is out of the MI355X hot path (SURVEY.md
section 8).  The names resolve so that ``training.py`` imports unchanged; constructing one raises with the reason.

When the reference's own module of this name is importable behind this source root (dcn_hip/_dropin.py) this
placeholder steps aside for it at import time.
"""
from dcn_hip._dropin import step_aside_for_reference as _step_aside

if not _step_aside(__name__, __file__):
    _WHY = ("pytorch_segmentation_detection.transforms.%s is an import-time placeholder: the dense-correspondence path does not "
            "use the segmentation toolbox's joint transforms, and they are not re-implemented here")


    class _Placeholder(object):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(_WHY % type(self).__name__)


    class ComposeJoint(_Placeholder):
        pass


    class RandomHorizontalFlipJoint(_Placeholder):
        pass


    class RandomScaleJoint(_Placeholder):
        pass


    class CropOrPad(_Placeholder):
        pass


    class ResizeAspectRatioPreserve(_Placeholder):
        pass


    class RandomCropJoint(_Placeholder):
        pass


    class Split2D(_Placeholder):
        pass
