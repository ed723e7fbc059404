from . import resnet_dilated  # noqa: F401
