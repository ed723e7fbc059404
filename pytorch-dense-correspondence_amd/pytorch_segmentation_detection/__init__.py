"""Import-name shim for the reference's un-vendored submodule ``external/pytorch-segmentation-detection``
(imported at dense_correspondence/network/dense_correspondence_network.py:16).  Only the piece on the
training hot path exists here: ``models.resnet_dilated`` backed by the gfx950 kernels."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
