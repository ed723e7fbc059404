import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-dense-correspondence_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive sweep, skipped unless DCN_RUN_SLOW=1 (keeps the default GPU suite inside its time limit)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("DCN_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="exhaustive sweep: set DCN_RUN_SLOW=1 to run it")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def dcn_env(monkeypatch):
    """Set DCN_* overrides for one test.  The library reads the environment ONCE (csrc/dcn_tuning.h), so: set -> reload,
    and after the test: restore -> reload."""
    from dcn_hip import _lib

    def set_env(**kw):
        for k, v in kw.items():
            monkeypatch.setenv(k, str(v))
        _lib.get().dcn_reload_env()
    yield set_env
    monkeypatch.undo()
    _lib.get().dcn_reload_env()
