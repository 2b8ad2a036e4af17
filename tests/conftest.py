import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _reload_options():
    """The library reads its RCX_* switches once; tests that flip one reload them (rcx_reload_options)."""
    try:
        from recnext_amd import _lib
        if os.path.exists(_lib.LIB_PATH):
            _lib.load().rcx_reload_options()
    except Exception:
        pass


@pytest.fixture(autouse=True)
def _rcx_switches_follow_the_environment(monkeypatch):
    """monkeypatch.setenv / delenv of an RCX_* switch take effect at once; every test starts from the environment as it is."""
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        orig_set(name, value, prepend)
        if name.startswith("RCX_"):
            _reload_options()

    def delenv(name, raising=True):
        orig_del(name, raising)
        if name.startswith("RCX_"):
            _reload_options()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    _reload_options()
    yield
