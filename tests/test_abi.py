"""CPU tier: the C-ABI library loads and exports every symbol include/platanus3_amd.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes

import pytest

from platanus3_amd import _build, capi


@pytest.fixture(scope="module")
def lib():
    _build.build_lib()
    return ctypes.CDLL(_build.LIB)


def test_header_symbols_exported(lib):
    names = capi.exported_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export: " + n


def test_no_device_fails_loudly():
    L = capi.load()
    if L.p3_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.P3Error) as e:
        capi.Context(0)
    assert e.value.code == 4  # P3_EHIP: no CPU fallback exists


def test_product_does_not_touch_oracle():
    """The product package must never import/link the oracle (it is test infrastructure)."""
    import os
    root = os.path.dirname(os.path.abspath(capi.__file__))
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".c", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "p3_oracle" not in txt and "p3o_" not in txt and "libp3ref" not in txt, f
