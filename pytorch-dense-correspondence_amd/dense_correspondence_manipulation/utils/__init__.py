"""``dense_correspondence_manipulation.utils``: ``utils`` / ``constants`` placeholders that step aside for the reference's;
``transformations``, ``visualization`` ... come from the reference's directory of the same package."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
