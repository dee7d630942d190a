"""The oracle's streaming form (oracle/p3_oracle_stream.cpp: the restatement arranged so that inputs of the BASELINE
configurations' full size fit this container) against the oracle proper on small inputs -- same statistics, same table
(through the digest), same Bloom words, same seeds -- for one- and two-word keys, ragged reads with unknown characters,
partitioned and unpartitioned count passes."""
import numpy as np
import pytest

import platanus3_amd.synth as synth
from util import random_reads


def _check(oracle, reads, k, m, H, part_bits):
    ekeys, ecounts = oracle.count_short(reads, min_len=k)
    ebloom, eseeds, nins, _ = oracle.make_bf(reads, ekeys, ecounts, k, m, H)
    s = oracle.stream_pass(reads, k, m, H, part_bits=part_bits)
    assert s["n_distinct"] == ekeys.size and s["n_windows"] == int(ecounts.sum())
    assert s["n_solid"] == int((ecounts >= 2).sum()) and s["max_count"] == (int(ecounts.max()) if ecounts.size else 0)
    assert (s["digest_sum"], s["digest_xor"]) == oracle.table_digest(ekeys, ecounts)
    assert s["n_passing"] == nins
    assert np.array_equal(s["bloom"], ebloom)
    assert [oracle.decode(row, k) for row in s["seeds"]] == eseeds


@pytest.mark.parametrize("k,part_bits", [(21, 0), (31, 3), (45, 2), (63, 4), (64, 1)])
def test_stream_equals_oracle_on_synthetic_reads(oracle, k, part_bits):
    b, o = synth.reads(seed=40 + k, total_reads=3000, read_len=150)
    m, H = oracle.estimate_bloom(int(b.size), k)
    _check(oracle, (b, o), k, m, H, part_bits)


def test_stream_equals_oracle_on_ragged_reads(oracle):
    reads = random_reads(np.random.default_rng(9), 400, lo=15, hi=260)  # unknown characters, lower case, reads shorter than k
    _check(oracle, reads, 31, 500009, 7, 2)
    _check(oracle, reads, 41, 1 << 20, 10, 0)
