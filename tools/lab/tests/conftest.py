"""Tests of the LAB build (make -C sharp_amd/csrc LAB=1 -> sharp_amd/variants/libsharp_hip_lab.so): kernels that were measured and not adopted.
Not part of the product's suite (pytest.ini's testpaths = tests): run on a GPU box with  python -m pytest tools/lab/tests -q"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sharp_amd import _lib  # noqa: E402

_lib._SO = os.path.join(os.path.dirname(_lib._SO), "variants", "libsharp_hip_lab.so")
if not os.path.exists(_lib._SO):
    pytest.exit("build the lab library first: make -C sharp_amd/csrc LAB=1", returncode=2)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(autouse=True)
def _sharp_env_options(monkeypatch):
    """the library reads its SHARP_* switches once: a test that sets one through monkeypatch asks it to read them again"""
    real, real_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        real(name, value, *a, **k)
        if name.startswith("SHARP_"):
            _lib.reload_options()

    def delenv(name, *a, **k):
        real_del(name, *a, **k)
        if name.startswith("SHARP_"):
            _lib.reload_options()

    monkeypatch.setenv = setenv
    monkeypatch.delenv = delenv
    _lib.reload_options()
    yield
