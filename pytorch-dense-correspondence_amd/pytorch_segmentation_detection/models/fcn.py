"""Import-time shim for ``pytorch_segmentation_detection.models.fcn`` (``import ... as fcns``,
dense_correspondence/training/training.py:28).  The reference imports the module but never touches a name of it on the
training hot path (its networks come from ``models.resnet_dilated``, network.py:373-375); the FCN-8s/16s/32s heads of the
original package are not part of the dense-correspondence path and are not provided.  Any attribute access says so.

When the reference's own module of this name is importable behind this source root (dcn_hip/_dropin.py) this
placeholder steps aside for it at import time.
"""
from dcn_hip._dropin import step_aside_for_reference as _step_aside

if not _step_aside(__name__, __file__):
    _WHY = ("pytorch_segmentation_detection.models.fcn.%s is not provided: the dense-correspondence training path only uses "
            "models.resnet_dilated.Resnet{18,34,50,101}_8s (dense_correspondence_network.py:373-375), which this package "
            "implements on the MI355X engine")


    class NotProvided(AttributeError, NotImplementedError):
        """An AttributeError (so that ``hasattr`` / ``getattr(module, name, default)`` / inspect / mock / pickle helpers keep
        working) that also says WHY the name is missing; still catchable as NotImplementedError."""


    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise NotProvided(_WHY % name)
