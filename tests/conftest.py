import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from platanus3_amd import capi
        return capi.load().p3_device_count() > 0
    except Exception:  # noqa: BLE001  (library missing or not loadable)
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a HIP device skips the GPU tier instead of failing in capi.Context(0).  An
    explicit `-m gpu` is left alone: on the GPU box a missing device or library must fail loudly."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device here (GPU tier: run with -m gpu on an MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle.p3_oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reflib():
    from oracle.p3_oracle import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/libp3ref.so not present (needs /root/reference to build)")
    return RefLib()
