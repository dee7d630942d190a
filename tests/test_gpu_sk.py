"""GPU tier: the run-length ("super-k-mer") record form of the two partition stages (p3_kernels_sk.h, option
run_records = 2; off by default because it measured slower) -- bit-exact against the oracle, and
its fallbacks: unknown bases, k > 32, counts beyond the 22-bit slot field."""
import numpy as np
import pytest

import platanus3_amd.synth as synth
from platanus3_amd import capi

pytestmark = pytest.mark.gpu


def _check(ctx, oracle, b, o, k, m, H):
    keys, counts = ctx.export_counts()
    ek, ec = oracle.count_short((b, o), min_len=k)
    assert np.array_equal(keys, ek) and np.array_equal(counts, ec)
    eb, es, _, _ = oracle.make_bf((b, o), ek, ec, k, m, H)
    assert np.array_equal(ctx.export_bloom(), eb)
    assert [capi.decode_key(r, k) for r in ctx.export_seeds()] == es
    assert ctx.get_info("nonsolid_windows") == int((ec == 1).sum())


@pytest.mark.parametrize("n,k,opts", [(3000, 31, {}), (20000, 21, {}), (20000, 32, {"sk_ts_log2": 9}), (60000, 27, {"sk_waves": 8}),
                                      (200000, 31, {"sk_ts_log2": 11})])
def test_run_records_match_oracle(oracle, n, k, opts):
    b, o = synth.reads(seed=70 + k, total_reads=n, read_len=150)
    m, H = oracle.estimate_bloom(int(b.size), k)
    ctx = capi.Context(0)
    try:
        ctx.set_option("run_records", 2)
        for name, v in opts.items():
            ctx.set_option(name, v)
        ctx.load_reads(b, o, min_len=k)
        ctx.run_pass(k, m, H)
        assert ctx.get_info("run_records") == 3  # both partitions used the record form
        _check(ctx, oracle, b, o, k, m, H)
        ctx.run_pass(k, m, H)                    # the single-synchronisation replay
        assert ctx.get_info("run_records") == 3
        _check(ctx, oracle, b, o, k, m, H)
    finally:
        ctx.close()


def test_run_records_fall_back(oracle):
    """unknown bases (00 on both strands in the reference, src/BitCalc.cpp:13-16,27-30: a record of forward bases cannot
    tell one from an A) and k > 32 take the per-window kernels; the results are the oracle's either way"""
    from util import random_reads
    rng = np.random.default_rng(9)
    from oracle.p3_oracle import _flat
    b, o = _flat(random_reads(rng, 2000, lo=60, hi=200, genome_len=20000, unknown=True))
    ctx = capi.Context(0)
    try:
        ctx.set_option("run_records", 2)
        for k in (31, 41):
            m, H = oracle.estimate_bloom(int(b.size), k)
            ctx.load_reads(b, o, min_len=k)
            ctx.run_pass(k, m, H)
            assert ctx.get_info("run_records") == 0
            _check(ctx, oracle, b, o, k, m, H)
        b2, o2 = synth.reads(seed=77, total_reads=5000, read_len=150)   # clean reads, k > 32: count stage only
        m, H = oracle.estimate_bloom(int(b2.size), 45)
        ctx.load_reads(b2, o2, min_len=45)
        ctx.run_pass(45, m, H)
        assert ctx.get_info("run_records") == 1
        _check(ctx, oracle, b2, o2, 45, m, H)
    finally:
        ctx.close()
