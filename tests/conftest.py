import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the suite (the driver runs `pytest -m gpu -x`): the cheap stage-level parity tests first, then the configurations at their full
# sizes, and the tests that start other processes (bench.py lines, RCCL / gloo workers) last -- so that anything that can go wrong around a
# subprocess (a port, a slow box, host memory) cannot take a stage's parity evidence with it.  Files not listed keep their place in front.
_ORDER = ["test_rp_gpu", "test_rp_pc_gpu", "test_linalg_gpu", "test_hclust_gpu", "test_pipeline_gpu", "test_markers_gpu", "test_dotc_gpu",
          "test_rglue_gpu", "test_variants_gpu", "test_sparse_gpu", "test_x_storage_gpu", "test_blocks_gpu", "test_decisions_gpu",
          "test_multi_device_gpu", "test_configs_gpu", "test_fullsize_gpu", "test_dist_gloo", "test_dist_rccl_gpu"]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i + 1 for i, name in enumerate(_ORDER)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, 0)

    items.sort(key=key)          # stable: the order inside a file is the file's


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
