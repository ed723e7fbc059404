"""``pytorch_segmentation_detection.models``: ``resnet_dilated`` on the MI355X engine; ``fcn`` is a placeholder that steps
aside for the real toolbox when its checkout is on the path."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
from . import resnet_dilated  # noqa: F401,E402
