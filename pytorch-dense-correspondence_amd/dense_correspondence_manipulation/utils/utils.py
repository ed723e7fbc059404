"""The handful of helpers of modules/dense_correspondence_manipulation/utils/utils.py that the network wrapper
calls (yaml I/O, path helpers, checkpoint discovery, seeds) -- host plumbing, restated for Python 3.

When the reference's own module of this name is importable behind this source root (dcn_hip/_dropin.py) this
placeholder steps aside for it at import time.
"""
from dcn_hip._dropin import step_aside_for_reference as _step_aside

if not _step_aside(__name__, __file__):
    import fnmatch
    import os
    import random

    import numpy as np
    import torch
    import yaml


    def getDictFromYamlFilename(filename):
        # utils.py:23-27
        with open(filename) as f:
            return yaml.safe_load(f)


    def saveToYaml(data, filename, flush=False):
        # utils.py:29-45
        with open(filename, "w") as outfile:
            yaml.dump(data, outfile, default_flow_style=False)
            if flush:
                outfile.flush()


    def getDenseCorrespondenceSourceDir():
        return os.getenv("DC_SOURCE_DIR")


    def get_data_dir():
        return os.getenv("DC_DATA_DIR")


    def getPdcPath():
        return get_data_dir()


    def add_dense_correspondence_to_python_path():
        """utils.py:133-142 appends the reference's source dirs to sys.path; nothing to add for this package."""
        return None


    def convert_to_absolute_path(path):
        # utils.py:145-159 -- an existing directory is returned as it is, anything else is taken relative to the HOME
        # directory (the relative paths stored in training.yaml files: path_to_network_params_folder, model folders)
        if os.path.isdir(path):
            return path
        return os.path.join(os.path.expanduser("~"), path)


    def getPaddedString(idx, width=6):
        return str(idx).zfill(width)


    def get_model_param_file_from_directory(model_folder, iteration=None):
        # utils.py:279-309
        if not os.path.isdir(model_folder) and getPdcPath() is not None:
            model_folder = os.path.join(getPdcPath(), "trained_models", model_folder)
        if iteration is None:
            files = os.listdir(model_folder)
            model_param_file = sorted(fnmatch.filter(files, "*.pth"))[-1]
            iteration = int(model_param_file.split(".")[0])
            optim = sorted(fnmatch.filter(files, "*.pth.opt"))
            optim_param_file = optim[-1] if optim else model_param_file + ".opt"
        else:
            prefix = getPaddedString(iteration, width=6)
            model_param_file = prefix + ".pth"
            optim_param_file = prefix + ".pth.opt"
        return os.path.join(model_folder, model_param_file), os.path.join(model_folder, optim_param_file), iteration


    def flattened_pixel_locations_to_u_v(flat_pixel_locations, image_width):
        # utils.py:312-323 (integer division, torch-1.1 semantics)
        return (flat_pixel_locations % image_width, flat_pixel_locations // image_width)


    def uv_to_flattened_pixel_locations(uv_tuple, image_width):
        return uv_tuple[1] * image_width + uv_tuple[0]


    def reset_random_seed():
        # utils.py:332-336
        SEED = 1
        random.seed(SEED)
        np.random.seed(SEED)
        torch.manual_seed(SEED)
