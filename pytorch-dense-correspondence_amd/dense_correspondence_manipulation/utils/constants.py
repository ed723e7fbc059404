"""Numeric constants of modules/dense_correspondence_manipulation/utils/constants.py:15-19.

When the reference's own module of this name is importable behind this source root (dcn_hip/_dropin.py) this
placeholder steps aside for it at import time.
"""
from dcn_hip._dropin import step_aside_for_reference as _step_aside

if not _step_aside(__name__, __file__):
    IMAGE_NET_MEAN = [0.485, 0.456, 0.406]
    IMAGE_NET_STD_DEV = [0.229, 0.224, 0.225]
    DEFAULT_IMAGE_MEAN = [0.5573105812072754, 0.37420374155044556, 0.37020164728164673]
    DEFAULT_IMAGE_STD_DEV = [0.24336038529872894, 0.2987397611141205, 0.31875079870224]
