"""dcn_hip -- ctypes binding of libdcn_hip.so (include/dcn_hip.h), the hand-written gfx950 kernels behind the
reference-compatible API in ``dense_correspondence`` / ``pytorch_segmentation_detection``.

There is NO fallback: if the shared library is missing or a tensor is not on the GPU the calls raise."""
from . import _lib  # noqa: F401
from ._lib import library_info, load  # noqa: F401
