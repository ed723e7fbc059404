"""How this source root shares its package names with the reference's own tree.

The reference's training driver imports, next to the hot-path modules this root replaces, modules this root does not (and
must not) provide: ``dense_correspondence.training.training``, the real ``SpartanDataset``
(``dense_correspondence/dataset/spartan_dataset_masked.py:111``), ``dense_correspondence.evaluation.evaluation``
(``dense_correspondence/training/training.py:38-43``), ``dense_correspondence_manipulation.utils.transformations`` ...
With this root in FRONT of the reference's roots on ``sys.path`` (``$DC_SOURCE_DIR`` and ``$DC_SOURCE_DIR/modules``):

* every package here is a *path-merging* package (``merge_package_path`` = ``pkgutil.extend_path``): a submodule is looked up
  in this root first and in the reference's directory of the same package second, so ``network`` / ``loss_functions`` /
  ``resnet_dilated`` resolve to the MI355X path and ``training`` / ``evaluation.evaluation`` / ``scene_structure`` to the
  reference;
* a *placeholder* module here (a few names, enough for the hot path to import when the reference is absent) steps aside for
  the reference's module of the same name when that one exists and imports (``step_aside_for_reference``);
* a module here that provides only PART of the reference module's names hands the rest on (``reference_sibling``).

Nothing in this file touches the device path."""
import importlib.machinery
import importlib.util
import os
import pkgutil
import sys

_QUIET = os.environ.get("DCN_QUIET_SHIMS") == "1"


class _MergedPath(list):
    """A package ``__path__`` that re-merges whenever ``sys.path`` has changed since the last lookup: the reference extends
    ``sys.path`` at run time (``utils.add_dense_correspondence_to_python_path()``, training.py:27), possibly after a package
    of this root has been imported."""

    def __init__(self, own, name):
        list.__init__(self, own)
        self._own, self._name, self._seen = list(own), name, None
        self._refresh()

    def _refresh(self):
        key = tuple(sys.path)
        if key != self._seen:
            self._seen = key
            self[:] = pkgutil.extend_path(list(self._own), self._name)

    def __iter__(self):
        self._refresh()
        return list.__iter__(self)

    def __len__(self):
        self._refresh()
        return list.__len__(self)

    def __getitem__(self, i):
        self._refresh()
        return list.__getitem__(self, i)


def merge_package_path(path, name):
    """``__path__ = merge_package_path(__path__, __name__)`` in a package's ``__init__``: the directories of the same
    package under every other ``sys.path`` entry are appended (regular and namespace packages alike;
    ``pkgutil.extend_path``), and again whenever ``sys.path`` changes."""
    return _MergedPath(path, name)


def _search_path(fullname, this_file):
    here = os.path.dirname(os.path.abspath(this_file))
    parent = fullname.rpartition(".")[0]
    if parent:
        pkg = sys.modules.get(parent)
        entries = list(getattr(pkg, "__path__", []) or [])
    else:
        entries = list(sys.path)
    return [p for p in entries if os.path.abspath(p or os.getcwd()) != here]


def find_reference_sibling(fullname, this_file):
    """Spec of the module ``fullname`` found anywhere on its package's (merged) path EXCEPT next to ``this_file``; None if
    there is none.  Goes through ``sys.path_hooks`` like a normal import."""
    path = _search_path(fullname, this_file)
    if not path:
        return None
    try:
        spec = importlib.machinery.PathFinder.find_spec(fullname, path)
    except (ImportError, ValueError):
        return None
    if spec is None or spec.loader is None or not spec.origin:
        return None
    if os.path.abspath(spec.origin) == os.path.abspath(this_file):
        return None
    return spec


def _note(msg):
    if not _QUIET:
        sys.stderr.write(msg + "\n")


def step_aside_for_reference(fullname, this_file):
    """Called at the TOP of a placeholder module.  If the reference's module of the same name is on the merged path and
    imports, it takes this module's place in ``sys.modules`` (the import statement that triggered us then returns IT) and
    True is returned: the caller must stop defining things (``if not step_aside...:``).  A reference module that exists but
    does not import (the reference is Python 2: a SyntaxError under this interpreter unless it has been converted) leaves the
    placeholder in charge and says so once on stderr (``DCN_QUIET_SHIMS=1`` silences it)."""
    spec = find_reference_sibling(fullname, this_file)
    if spec is None:
        return False
    mod = importlib.util.module_from_spec(spec)
    shim = sys.modules.get(fullname)
    sys.modules[fullname] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException as e:      # SyntaxError (py2 source), ImportError (cv2, torchvision ...), anything at import time
        if shim is not None:
            sys.modules[fullname] = shim
        else:
            sys.modules.pop(fullname, None)
        if isinstance(e, (KeyboardInterrupt, SystemExit)):
            raise
        _note("%s: the reference's module %s does not import here (%s: %s) -- keeping the hot-path placeholder"
              % (fullname, spec.origin, type(e).__name__, e))
        return False
    return True


class reference_sibling(object):
    """``_ref = reference_sibling(__name__, __file__)`` in a module that implements only part of its reference namesake;
    ``_ref.get()`` is the reference's module (loaded on first use under a private name, NOT replacing this one) or None, and
    ``def __getattr__(name): return _ref.attr(name)`` at module level hands every name this module lacks on to it."""

    def __init__(self, fullname, this_file):
        self._fullname, self._file = fullname, this_file
        self._mod, self._tried, self._why = None, False, None

    def get(self):
        if not self._tried:
            self._tried = True
            spec = find_reference_sibling(self._fullname, self._file)
            if spec is None:
                self._why = "no module %s on the reference's side of the package path" % self._fullname
                return None
            private = self._fullname + "__reference"
            try:   # (the same kind of loader under the private name: a source-file loader or a subclass of one, e.g. a converting one)
                spec2 = importlib.util.spec_from_file_location(private, spec.origin, loader=type(spec.loader)(private, spec.origin))
                mod = importlib.util.module_from_spec(spec2)
            except Exception as e:
                self._why = "%s cannot be loaded under a private name (%s: %s)" % (spec.origin, type(e).__name__, e)
                return None
            mod.__package__ = self._fullname.rpartition(".")[0]      # its relative / sibling imports resolve as usual
            sys.modules[private] = mod
            try:
                spec2.loader.exec_module(mod)
            except BaseException as e:
                sys.modules.pop(private, None)
                if isinstance(e, (KeyboardInterrupt, SystemExit)):
                    raise
                self._why = "%s does not import here (%s: %s)" % (spec.origin, type(e).__name__, e)
                return None
            self._mod = mod
        return self._mod

    def why_not(self):
        self.get()
        return self._why

    def attr(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        mod = self.get()
        if mod is None or not hasattr(mod, name):
            raise AttributeError("module %r has no attribute %r (%s)" % (
                self._fullname, name, self._why or "nor has the reference's module of that name"))
        return getattr(mod, name)
