"""``dense_correspondence.evaluation``: the batched descriptor export (``utils``) here, ``evaluation`` / ``plotting`` from the
reference's directory of the same package."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
