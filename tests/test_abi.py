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


def test_probe_partition_layout_stays_within_its_32_bit_offsets():
    """p3_probe_plan (host arithmetic of the Bloom stage, no GPU): whenever the LDS-slice form is chosen, the fixed-stride
    layout of one sweep -- buckets x workgroups x the segment capacity AT THE BOOST LEVEL -- stays below 2^32 records.
    Round 2's rule looked at the positions per sweep only: config #3 (265 M distinct k-mers, H = 19, m = 3.3 G bits) at
    boost level 1 laid out 5.3 G records and the last 200 buckets' offsets wrapped (0.03 % of the vector wrong)."""
    import ctypes as C
    from platanus3_amd import capi
    L = capi.load()
    out = (C.c_uint64 * 6)()
    seen_multi = False
    for n_keys in (5_300_000, 88_000_000, 264_787_650, 400_000_000, 1_200_000_000):
        for m in (66_855_782, 3_342_789_109, 9_057_880_166, 1 << 30, (1 << 32) - 5):
            for H in (10, 19):
                for boost in (0, 1, 2):
                    assert L.p3_probe_plan(n_keys, H, m, boost, out) == 0
                    slice_log2, sub, sweeps, segcap, layout, sliced = [int(x) for x in out]
                    if sliced:
                        assert layout < (1 << 32), (n_keys, m, H, boost, list(out))
                        assert sweeps * 1024 >= (m + (1 << slice_log2) - 1) >> slice_log2
                        seen_multi = seen_multi or sweeps > 1
    assert seen_multi
    # the case that went wrong: level 0 may use buckets of two slices (2 sweeps), level 1 must not
    assert L.p3_probe_plan(264_787_650, 19, 3_342_789_109, 0, out) == 0 and int(out[5]) == 1 and int(out[4]) < (1 << 32)
    lvl0 = (int(out[1]), int(out[2]))
    assert L.p3_probe_plan(264_787_650, 19, 3_342_789_109, 1, out) == 0 and int(out[4]) < (1 << 32)
    assert (int(out[1]), int(out[2])) != lvl0 or int(out[5]) == 0


def test_a_later_position_chunk_never_needs_more_room_than_the_first():
    """p3_segment_capacity (host arithmetic of the partition step, no GPU).  Position chunks cut the stream into C equal
    word ranges -- the last one shorter -- and lay out ONE region size for all of them from the first.  The estimate used to
    divide by the number of workgroups that get ANY word: 8,218 words in 8 chunks = seven of 1,028 words (two workgroups:
    1,024 + 4) and one of 1,022 (one workgroup) -- the last chunk asked for twice the first chunk's segments and the
    partition step regrew what was a window of the chunks' common buffer (memory fault on the device).  With the busiest
    workgroup's share as the divisor the first chunk is the largest, whatever the sizes."""
    import numpy as np
    from platanus3_amd import capi
    L = capi.load()
    rng = np.random.default_rng(5)
    cases = [(8218, 204540, 8)] + [(int(rng.integers(100, 5_000_000)), 0, int(rng.choice([2, 4, 8, 16]))) for _ in range(300)]
    for n_words, windows, C in cases:
        if not windows:
            windows = int(n_words * 32 * rng.uniform(0.3, 1.0))
        wc = (n_words + C - 1) // C
        w0 = [min(i * wc, n_words) for i in range(C + 1)]
        for level in (0, 1, 2):
            caps = []
            for ci in range(C):
                nw = w0[ci + 1] - w0[ci]
                if nw == 0:
                    continue
                hw = int(windows * nw / n_words) + 1
                caps.append(L.p3_segment_capacity(nw, hw, level))
            assert caps and max(caps) == caps[0], (n_words, windows, C, level, caps)
    # the stream-word bound (no window count from the host) is monotonic in the words as well
    assert L.p3_segment_capacity(1028, 0, 0) >= L.p3_segment_capacity(1022, 0, 0)
