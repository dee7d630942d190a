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
    from oracle.p3_oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reflib():
    from oracle.p3_oracle import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/libp3ref.so not present (needs /root/reference to build)")
    return RefLib()
