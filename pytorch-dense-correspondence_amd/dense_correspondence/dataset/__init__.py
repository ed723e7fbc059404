"""``dense_correspondence.dataset``: nothing of the dataset is re-implemented (pair generation and image
loading stay the reference's); ``spartan_dataset_masked`` here is a placeholder that steps aside for the reference's."""
from dcn_hip._dropin import merge_package_path as _merge

__path__ = _merge(__path__, __name__)   # the reference's modules of this package stay importable next to these (dcn_hip/_dropin.py)
