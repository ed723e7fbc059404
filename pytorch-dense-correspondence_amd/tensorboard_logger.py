"""Import-time shim for ``tensorboard_logger`` (``import tensorboard_logger``, dense_correspondence/training/training.py:20;
``tensorboard_logger.Logger(dir)`` :584, ``.log_value(name, value, step)`` :364-411).  The real package is not installed in
this image (no network).  ``Logger`` keeps the reference's two-call surface and appends ``step<TAB>name<TAB>value`` lines to
``<logdir>/scalars.tsv`` -- enough for the training loop to run and for its curves to be read back; it is host-side
bookkeeping, not part of the MI355X path."""
import os
import sys
import threading

__all__ = ["Logger", "configure", "log_value"]


def _real_package():
    """The real ``tensorboard_logger`` if one is installed: this shim sits on the package path under the same name and must
    not shadow it (training would silently write scalars.tsv instead of TensorBoard event files).  Looked up with this file's
    directory taken off the search path."""
    import importlib.machinery
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    try:   # (PathFinder on an explicit path list: importlib.util.find_spec would hand back THIS module, already in sys.modules)
        spec = importlib.machinery.PathFinder.find_spec(
            "tensorboard_logger", [p for p in sys.path if os.path.abspath(p or os.getcwd()) != here])
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin or os.path.abspath(spec.origin) == os.path.abspath(__file__):
        return None
    mod = importlib.util.module_from_spec(spec)
    shim = sys.modules.get(__name__)
    sys.modules[__name__] = mod        # `import tensorboard_logger` now yields the real package everywhere
    try:
        spec.loader.exec_module(mod)
    except Exception:                  # a broken installation: fall back to the shim
        if shim is not None:
            sys.modules[__name__] = shim
        else:
            sys.modules.pop(__name__, None)
        return None
    return mod


_REAL = None if __name__ != "tensorboard_logger" else _real_package()
if _REAL is None and os.environ.get("DCN_QUIET_SHIMS") != "1":
    sys.stderr.write("tensorboard_logger: the real package is not installed -- using the dcn_hip shim (scalars are appended "
                     "to <logdir>/scalars.tsv; log_histogram / log_images are not provided)\n")


class Logger(object):
    def __init__(self, logdir, flush_secs=2, **_unused):
        self.logdir = str(logdir)
        os.makedirs(self.logdir, exist_ok=True)
        self._path = os.path.join(self.logdir, "scalars.tsv")
        self._lock = threading.Lock()

    def log_value(self, name, value, step=None):
        if hasattr(value, "item"):
            value = value.item()
        with self._lock, open(self._path, "a") as f:
            f.write("%s\t%s\t%r\n" % ("" if step is None else int(step), name, float(value)))
        return value

    def log_histogram(self, name, value, step=None):
        raise NotImplementedError("tensorboard_logger shim: only scalar logging (log_value) is provided")

    def log_images(self, name, images, step=None):
        raise NotImplementedError("tensorboard_logger shim: only scalar logging (log_value) is provided")


_default = [None]


def configure(logdir, flush_secs=2):
    _default[0] = Logger(logdir, flush_secs)
    return _default[0]


def log_value(name, value, step=None):
    if _default[0] is None:
        raise RuntimeError("tensorboard_logger.configure(logdir) has not been called")
    return _default[0].log_value(name, value, step)
