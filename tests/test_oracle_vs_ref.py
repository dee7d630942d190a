"""CPU tier, build container only: the oracle against the REAL reference run live
(oracle/_ref/libp3ref.so).  Skipped where the compiled reference is absent."""
import numpy as np
import pytest

import platanus3_amd.synth as synth
from util import random_reads


@pytest.mark.parametrize("k", [21, 27, 31, 32, 33, 48, 63, 64])
def test_random_reads_live(oracle, reflib, k):
    rng = np.random.default_rng(100 + k)
    reads = random_reads(rng, 150)
    all_bases = sum(len(r) for r in reads if len(r) >= k)
    m, H = reflib.estimate_bloom(all_bases, k)
    assert (m, H) == oracle.estimate_bloom(all_bases, k)
    ref = reflib.run(reads, k, m, H)
    keys, counts = oracle.count_short(reads, min_len=k)
    assert np.array_equal(keys, ref["keys"]) and np.array_equal(counts, ref["counts"])
    words, seeds, _, _ = oracle.make_bf(reads, keys, counts, k, m, H)
    assert np.array_equal(words, ref["bloom"])
    assert seeds == ref["seeds"]
    # B2 on a mix of inserted / random keys
    from oracle.p3_oracle import nwords
    nw = nwords(k)
    q = rng.integers(0, 1 << 62, size=(64, nw), dtype=np.uint64)
    if 2 * k - 64 * (nw - 1) < 64:
        q[:, nw - 1] &= np.uint64((1 << (2 * k - 64 * (nw - 1))) - 1)
    for i, s in enumerate(seeds[:32]):
        f, r = oracle.encode_fwd(s), oracle.encode_rc(s)
        q[i] = oracle.canonical(f, r, k)
    assert np.array_equal(oracle.bloom_query(words, m, H, q, k), reflib.bloom_query(words, m, H, q, k))


def test_synth_2k_k31_live(oracle, reflib):
    b, o = synth.reads(seed=7, total_reads=2000, read_len=150)
    m, H = oracle.estimate_bloom(int(b.size), 31)
    ref = reflib.run((b, o), 31, m, H)
    keys, counts = oracle.count_short((b, o), min_len=31)
    assert np.array_equal(keys, ref["keys"]) and np.array_equal(counts, ref["counts"])
    words, seeds, _, _ = oracle.make_bf((b, o), keys, counts, 31, m, H)
    assert np.array_equal(words, ref["bloom"]) and seeds == ref["seeds"]
