"""``dense_correspondence_manipulation`` (the reference keeps it under ``modules/``): only ``utils`` has anything here."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
