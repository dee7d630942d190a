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


def test_group_of_ranks_reduces_on_the_host():
    """p3_group (several protocol ranks in one process): the members' barrier and the host-side all-reduce need no GPU --
    four threads, sums and maxima, many rounds (the barrier is reused); then a group is aborted by a member that fails"""
    import ctypes as C
    import threading
    from platanus3_amd import capi
    L = capi.load()
    V, rounds = 4, 200
    g = capi.Group(V, None)
    comms = [capi.Comm.member(g, i) for i in range(V)]
    bad = []

    def worker(i):
        for r in range(rounds):
            got = comms[i].all_reduce([i + r, 1, 10 * i], "sum")
            if got != [V * r + V * (V - 1) // 2, V, 10 * V * (V - 1) // 2]:
                bad.append((i, r, got))
            if comms[i].all_reduce([i * r], "max") != [(V - 1) * r]:
                bad.append((i, r, "max"))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(V)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in th) and not bad, bad[:3]
    assert [c.h is not None for c in comms] == [True] * V
    # member ranks and sizes as documented: rank = outer_rank * n_local + i
    for c in comms:
        c.close()
    g.close()
    # bad arguments
    h = C.c_void_p()
    assert L.p3_group_create(C.byref(h), 0, None) != 0
    assert L.p3_group_create(C.byref(h), 65, None) != 0
