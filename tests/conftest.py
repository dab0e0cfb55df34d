import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(autouse=True)
def _sharp_env_options(monkeypatch):
    """libsharp_hip reads its SHARP_* environment switches once; a test that sets one through monkeypatch asks it to read them
    again, and every test starts from the process environment."""
    from sharp_amd import _lib

    def reload():
        if os.path.exists(_lib.so_path()):
            _lib.reload_options()

    real = monkeypatch.setenv

    def setenv(name, value, *a, **k):
        real(name, value, *a, **k)
        if name.startswith("SHARP_"):
            reload()

    real_del = monkeypatch.delenv

    def delenv(name, *a, **k):
        real_del(name, *a, **k)
        if name.startswith("SHARP_"):
            reload()

    monkeypatch.setenv = setenv
    monkeypatch.delenv = delenv
    reload()
    yield
