"""``dense_correspondence.network``: the MI355X network wrapper (``dense_correspondence_network``)."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
