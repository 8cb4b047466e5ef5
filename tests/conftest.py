import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def monkeypatch(monkeypatch):
    """libuwu_hip.so caches its environment switches (UWU_GEMM_* ...): re-read them after every change a test makes,
    and once more after the changes are undone."""
    def refresh():
        from uwudiff_amd import lib

        if os.path.exists(lib.LIB_PATH):
            lib.load().uwu_env_refresh()

    real_set, real_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(*a, **k):
        real_set(*a, **k)
        refresh()

    def delenv(*a, **k):
        real_del(*a, **k)
        refresh()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
    monkeypatch.undo()
    refresh()
