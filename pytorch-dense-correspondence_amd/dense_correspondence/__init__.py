"""Host-side mirror of the reference's ``dense_correspondence`` package, restricted to the training hot path
(SURVEY.md section 8): same module paths, class / function names, argument order and return values, with the
arithmetic executed by the gfx950 kernels in ``dcn_hip``."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
