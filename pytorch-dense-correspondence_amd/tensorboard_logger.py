"""Import-time shim for ``tensorboard_logger`` (``import tensorboard_logger``, dense_correspondence/training/training.py:20;
``tensorboard_logger.Logger(dir)`` :584, ``.log_value(name, value, step)`` :364-411).  The real package is not installed in
this image (no network).  ``Logger`` keeps the reference's two-call surface and appends ``step<TAB>name<TAB>value`` lines to
``<logdir>/scalars.tsv`` -- enough for the training loop to run and for its curves to be read back; it is host-side
bookkeeping, not part of the MI355X path."""
import os
import threading

__all__ = ["Logger", "configure", "log_value"]


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
