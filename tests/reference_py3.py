"""Test infrastructure (not product, not a test module): IMPORT the reference's Python-2 modules under Python 3 -- from where
they lie under /root/reference, nothing is copied -- so that the reference's OWN training driver
(dense_correspondence/training/training.py:46-601) can be executed against this repository's packages.

* ``install()`` puts a path hook in front of ``sys.path_hooks`` that serves every directory under the reference root with a
  source loader which converts the module text IN MEMORY: tabs expanded (pixelwise_contrastive_loss.py mixes tabs and spaces),
  the torch-1.1 / py2 *semantic* patches of ``SEMANTIC_PATCHES`` (each one keeps the original meaning; the same list the
  golden generators use), then the stock ``lib2to3`` fixers (print statements, ``iteritems``, ``long``, implicit relative
  imports ...).  No bytecode is written (the reference tree is read-only for us).
* ``install_third_party_stubs()`` provides import-time stand-ins for modules the reference imports and this image lacks
  (``torchvision.transforms``: Compose / ToTensor / Normalize; ``cv2``): test doubles living in ``sys.modules`` only.

Only ``tests/`` and ``tests/golden/make_*`` use this; the GPU box has no /root/reference and never runs it."""
import importlib.machinery
import io
import os
import re
import sys
import tokenize
import types

REF = os.environ.get("DCN_REFERENCE_ROOT", "/root/reference")

# file (relative to the reference root) -> [(regex, replacement, why)]
SEMANTIC_PATCHES = {
    "dense_correspondence/loss_functions/pixelwise_contrastive_loss.py": [
        (r"num_non_matches / num_matches", "num_non_matches // num_matches", "py2 int division (pcl.py:113)"),
        (r"len\(non_matches_b\)/len\(matches_b\)", "len(non_matches_b)//len(matches_b)", "py2 int division (pcl.py:321)"),
        (r"u_v_pixel_locations\[:,1\]/self\.image_width", "u_v_pixel_locations[:,1]//self.image_width",
         "torch-1.1 LongTensor '/' is integer division (pcl.py:351)"),
    ],
}

_tool = [None]
_cache = {}


def available():
    return os.path.isdir(os.path.join(REF, "dense_correspondence"))


def _refactoring_tool():
    if _tool[0] is None:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # lib2to3 is deprecated (still shipped with 3.10)
            from lib2to3 import refactor
        _tool[0] = refactor.RefactoringTool(sorted(refactor.get_fixers_from_package("lib2to3.fixes")))
    return _tool[0]


def converted_source(path):
    """The Python-3 text of the reference module at ``path`` (in memory)."""
    path = os.path.abspath(path)
    if path in _cache:
        return _cache[path]
    with open(path, "rb") as f:
        raw = f.read()
    enc = tokenize.detect_encoding(io.BytesIO(raw).readline)[0]
    text = raw.decode(enc).expandtabs(8)
    rel = os.path.relpath(path, REF)
    for pat, rep, _why in SEMANTIC_PATCHES.get(rel, []):
        text, n = re.subn(pat, rep, text)
        assert n > 0, (rel, pat, "did not match -- reference changed?")
    if not text.endswith("\n"):
        text += "\n"
    out = str(_refactoring_tool().refactor_string(text, path))
    _cache[path] = out
    return out


class Py2SourceLoader(importlib.machinery.SourceFileLoader):
    def get_code(self, fullname):                     # no bytecode cache: never write under the reference root
        path = self.get_filename(fullname)
        return compile(converted_source(path), path, "exec", dont_inherit=True)

    def get_source(self, fullname):
        return converted_source(self.get_filename(fullname))


def _hook(path):
    ap = os.path.abspath(path)
    if not (ap == REF or ap.startswith(REF + os.sep)) or not os.path.isdir(ap):
        raise ImportError("not under the reference root")
    return importlib.machinery.FileFinder(ap, (Py2SourceLoader, importlib.machinery.SOURCE_SUFFIXES))


def install():
    if _hook not in sys.path_hooks:
        sys.path_hooks.insert(0, _hook)
        sys.path_importer_cache.clear()


def install_third_party_stubs():
    """Stand-ins for ``torchvision.transforms`` (network.py:15,22; the dataset modules) and ``cv2`` (evaluation.py:9,
    plotting.py:3) -- neither is installed in this image."""
    if "torchvision" not in sys.modules:
        import numpy as np
        import torch
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")

        class Compose(object):
            def __init__(self, transforms):
                self.transforms = transforms

            def __call__(self, x):
                for t in self.transforms:
                    x = t(x)
                return x

        class ToTensor(object):
            def __call__(self, pic):
                a = np.asarray(pic)
                if a.ndim == 2:
                    a = a[:, :, None]
                t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
                return t.float().div(255) if t.dtype == torch.uint8 else t

        class Normalize(object):
            def __init__(self, mean, std):
                self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

            def __call__(self, t):
                return (t - self.mean) / self.std

        tr.Compose, tr.ToTensor, tr.Normalize = Compose, ToTensor, Normalize
        tv.transforms = tr
        tv.__path__ = []
        sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")

        def _missing(name):
            raise AttributeError("cv2.%s: OpenCV is not installed in this image (test stand-in)" % name)
        cv2.__getattr__ = _missing
        sys.modules["cv2"] = cv2
