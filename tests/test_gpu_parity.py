"""GPU tier (-m gpu): the HIP path, called through the C ABI, against (a) the committed golden vectors
the real reference produced and (b) the oracle on the same seeded inputs.  Bit-exact everywhere
(integer/byte work): count table, Bloom words, seed set, Bloom queries."""
import os

import numpy as np
import pytest

import platanus3_amd.synth as synth
from platanus3_amd import capi
from util import flat, golden_cases, load_case, random_reads

pytestmark = pytest.mark.gpu


# P3_FUZZ_OFFSET=n shifts the seeds of the randomised sweeps (a one-off wider search; the committed cases are offset 0)
FUZZ_OFFSET = int(os.environ.get("P3_FUZZ_OFFSET", "0"))


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def seeds_as_strings(ctx):
    return [capi.decode_key(row, ctx.k) for row in ctx.export_seeds()]


def run_and_compare(ctx, oracle, reads, k, m, H, expect=None, loaded=False):
    bases, offs = flat(reads) if not isinstance(reads, tuple) else reads
    if not loaded:
        ctx.load_reads(bases, offs, min_len=k)
    st = ctx.count_short()
    keys, counts = ctx.export_counts()
    if expect is None:
        ekeys, ecounts = oracle.count_short((bases, offs), min_len=k)
        ebloom, eseeds, nins, _ = oracle.make_bf((bases, offs), ekeys, ecounts, k, m, H)
    else:
        ekeys, ecounts, ebloom, eseeds = expect["keys"], expect["counts"], expect["bloom"], expect["seeds"]
        nins = None
    assert np.array_equal(keys, ekeys)
    assert np.array_equal(counts, ecounts)
    assert st["n_distinct"] == ekeys.size
    assert st["n_windows"] == int(ecounts.sum())
    assert st["n_solid"] == int((ecounts >= 2).sum())
    bs = ctx.make_bloom(k, m, H)
    words = ctx.export_bloom()
    assert np.array_equal(words, ebloom), "Bloom words differ: %d of %d" % (int((words != ebloom).sum()), words.size)
    assert seeds_as_strings(ctx) == eseeds
    if nins is not None:
        assert bs["n_passing"] == nins
    return words, eseeds


@pytest.mark.parametrize("name", golden_cases())
def test_golden_cases(ctx, oracle, name):
    c = load_case(name)
    words, seeds = run_and_compare(ctx, oracle, c["reads"], c["k"], c["m"], c["H"], expect=c)
    # B2: seeds' canonical forms are in the filter; random keys agree with the oracle's query
    k = c["k"]
    nw = capi.nwords(k)
    rng = np.random.default_rng(5)
    q = rng.integers(0, 1 << 62, size=(128, nw), dtype=np.uint64)
    if 2 * k - 64 * (nw - 1) < 64:
        q[:, nw - 1] &= np.uint64((1 << (2 * k - 64 * (nw - 1))) - 1)
    for i, s in enumerate(seeds[:64]):
        q[i] = oracle.canonical(oracle.encode_fwd(s), oracle.encode_rc(s), k)
    got = ctx.bloom_query(q)
    assert np.array_equal(got, oracle.bloom_query(words, c["m"], c["H"], q, k))
    assert got[: min(64, len(seeds))].all()


@pytest.mark.parametrize("k,n,L,ppm", [(21, 10000, 150, 5000), (31, 20000, 150, 5000), (25, 5000, 150, 5000),
                                       (41, 5000, 150, 5000), (61, 5000, 150, 5000), (63, 300, 5000, 10000),
                                       (32, 3000, 150, 5000), (33, 3000, 150, 5000), (64, 3000, 200, 5000)])
def test_synth_vs_oracle(ctx, oracle, k, n, L, ppm):
    b, o = synth.reads(seed=k, total_reads=n, read_len=L, sub_rate_ppm=ppm)
    m, H = oracle.estimate_bloom(int(b.size), k)
    run_and_compare(ctx, oracle, (b, o), k, m, H)


def test_count_digest_is_the_oracles(ctx, oracle):
    """p3_count_digest: the order-independent (sum, xor) digest of the resident table equals the one computed from the
    oracle's table -- the check that stands in for an export where the table is too large to sort (config #3)."""
    b, o = synth.reads(seed=33, total_reads=20000, read_len=150)
    ctx.load_reads(b, o, min_len=31)
    ctx.count_short()
    ekeys, ecounts = oracle.count_short((b, o), min_len=31)
    assert ctx.count_digest() == oracle.table_digest(ekeys, ecounts)
    ecounts[0] += 1
    assert ctx.count_digest() != oracle.table_digest(ekeys, ecounts)


def test_minus_m_style_run(ctx, oracle):
    # "-m given" -> H stays 10 (src/Options.cpp:51), m = 2^24 here
    b, o = synth.reads(seed=3, total_reads=5000, read_len=150)
    run_and_compare(ctx, oracle, (b, o), 31, 1 << 24, 10)


def test_ragged_unknown_chars(ctx, oracle):
    rng = np.random.default_rng(77)
    reads = random_reads(rng, 3000, lo=1, hi=400, genome_len=20000)
    reads += [b"", b"A", b"ACGT" * 5, b"N" * 100, b"acgt" * 30]
    for k in (21, 31, 47):
        all_bases = sum(len(r) for r in reads if len(r) >= k)
        m, H = oracle.estimate_bloom(all_bases, k)
        run_and_compare(ctx, oracle, reads, k, m, H)


def test_filter_larger_than_2_32_bits(ctx, oracle):
    # m > 2^32 (config #4 regime): probe positions need 64-bit arithmetic end to end
    b, o = synth.reads(seed=9, total_reads=400, read_len=300)
    m = (1 << 32) + 1234567
    run_and_compare(ctx, oracle, (b, o), 63, m, 19)


def test_empty_and_degenerate_inputs(ctx, oracle):
    bases, offs = flat([b"ACGTACGTAC"])  # shorter than 21: no windows at all
    ctx.load_reads(bases, offs, min_len=21)
    st = ctx.count_short()
    assert st["n_distinct"] == 0 and st["n_windows"] == 0
    ctx.make_bloom(21, 1000, 5)
    assert not ctx.export_bloom().any()
    assert ctx.export_seeds(cap=4).shape[0] == 0
    # zero reads
    ctx.load_reads(np.zeros(1, np.uint8), np.zeros(1, np.uint64), min_len=21)
    assert ctx.count_short()["n_distinct"] == 0


@pytest.mark.parametrize("chunks,groups", [(1, 4), (4, 0), (8, 8)])
def test_grouped_reads_give_the_same_results(oracle, chunks, groups):
    """p3_group_reads: the reads put in the order of their shard key's group (for inputs that are processed in position
    chunks: a chunk then holds few genomic neighbourhoods at full coverage) -- every result unchanged"""
    b0, o0 = synth.reads(seed=33, total_reads=24000, read_len=150)
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 151, size=24000)
    b = np.concatenate([b0[int(o0[i]):int(o0[i]) + int(lens[i])] for i in range(24000)])
    o = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    k = 31
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    try:
        c.set_option("chunks", chunks)
        c.load_reads(b, o, min_len=k)
        c.group_reads(groups)
        run_and_compare(c, oracle, (b, o), k, m, H, loaded=True)
    finally:
        c.close()


@pytest.mark.parametrize("sub,m", [(1, 3000017), (2, 3000017), (2, 40000003), (1, 1 << 22)])
def test_probe_buckets_of_several_lds_slices(oracle, sub, m):
    """vectors beyond 2^30 bits put two or four LDS-sized slices into one bucket of the probe partition (fewer sweeps, each
    of which hashes every key again); k_bloom_slices then builds a bucket in parts.  `bloom_sub` forces that on small vectors."""
    b, o = synth.reads(seed=21, total_reads=20000, read_len=150)
    c = capi.Context(0)
    try:
        c.set_option("bloom_sub", sub)
        for k in (31, 45):
            run_and_compare(c, oracle, (b, o), k, m, 7)
            assert c.get_info("bloom_mode") == 3
    finally:
        c.close()


def test_timings_and_byte_counters(ctx):
    """p3_last_timings / p3_last_traffic (per-phase device times and byte counters): the counters
    follow section 8d's accounting -- 2 L + 16 (L - 20) + 16 D / R + m / (8 R) bytes per read in total"""
    n, L, k = 5000, 150, 31
    b, o = synth.reads(seed=9, total_reads=n, read_len=L)
    m, H = 4000037, 7
    ctx.load_reads(b, o, min_len=k)
    ctx.run_pass(k, m, H)
    t, tr, cs = ctx.timings(), ctx.traffic(), ctx.count_stats()
    assert t["ms_total"] > 0 and t["ms_count"] > 0 and t["ms_bloom"] > 0
    assert tr["bytes_pack"] == n * L and tr["bytes_gate"] == 0 and tr["bytes_seeds"] == 0
    assert tr["bytes_count"] == 16 * n * (L - 20) + 16 * cs["n_distinct"]
    assert tr["bytes_bloom"] == n * L + (m + 7) // 8
    assert tr["bytes_total"] == tr["bytes_pack"] + tr["bytes_count"] + tr["bytes_bloom"]


def test_error_codes(ctx):
    b, o = synth.reads(seed=1, total_reads=50, read_len=100)
    ctx.load_reads(b, o, min_len=31)
    ctx.count_short()
    for bad_k in (5, 20, 3002, 5001):
        with pytest.raises(capi.P3Error) as e:
            ctx.make_bloom(bad_k, 1000, 3)
        assert e.value.code == 2  # P3_EUNSUPPORTED_K (the reference silently returns 1)
    with pytest.raises(capi.P3Error) as e:
        ctx.make_bloom(31, 0, 3)
    assert e.value.code == 1
    c2 = capi.Context(0)
    with pytest.raises(capi.P3Error) as e:
        c2.count_short()
    assert e.value.code == 7
    c2.close()


def test_idempotent_rerun(ctx, oracle):
    b, o = synth.reads(seed=11, total_reads=4000, read_len=150)
    m, H = oracle.estimate_bloom(int(b.size), 31)
    ctx.load_reads(b, o, min_len=31)
    ctx.run_pass(31, m, H)
    w1, s1, k1 = ctx.export_bloom(), ctx.export_seeds(), ctx.export_counts()
    ctx.run_pass(31, m, H)
    w2, s2, k2 = ctx.export_bloom(), ctx.export_seeds(), ctx.export_counts()
    assert np.array_equal(w1, w2) and np.array_equal(s1, s2)
    assert np.array_equal(k1[0], k2[0]) and np.array_equal(k1[1], k2[1])


@pytest.mark.parametrize("gate", [1, 2])
@pytest.mark.parametrize("mode,ts_log2,expect_mode", [(1, 14, 1), (2, 14, 2), (0, 8, 2), (0, 5, 1)])
def test_count_paths(oracle, mode, ts_log2, expect_mode, gate):
    """direct global-table path, partitioned LDS path, partitioned with a re-split after LDS-table overflow
    (ts_log2=8), and overflow at every split -> fallback to direct (ts_log2=5): identical results."""
    c = capi.Context(0)
    try:
        c.set_option("count_mode", mode)
        c.set_option("ts_log2", ts_log2)
        c.set_option("gate_mode", gate)  # 1 = solid-table probes, 2 = return path (positions in the records)
        b, o = synth.reads(seed=21, total_reads=30000, read_len=150)
        m, H = oracle.estimate_bloom(int(b.size), 31)
        run_and_compare(c, oracle, (b, o), 31, m, H)
        assert c.get_info("count_mode") == expect_mode
        if expect_mode == 2:
            assert c.get_info("gate_mode") == gate
        if ts_log2 == 8:
            assert c.get_info("sbits") > 1
    finally:
        c.close()


@pytest.mark.parametrize("k,mode,tsk,expect_mode", [(31, 1, 14, 1), (31, 2, 14, 3), (31, 0, 7, 3), (31, 0, 4, 1),
                                                    (63, 2, 14, 3), (63, 0, 8, 3), (41, 1, 14, 1), (31, -2, 14, 2),
                                                    (63, -2, 13, 2)])
def test_bloom_paths(oracle, k, mode, tsk, expect_mode):
    """direct (probe per window) vs dedup (probe per distinct k-mer) Bloom build, including LDS-set overflow
    -> re-split (tsk_log2=7, 8: sub-bucket splits, then the second-level partition of the k-mer records) and, with
    that partition switched off, overflow at every split -> direct fallback (tsk_log2=4)."""
    c = capi.Context(0)
    try:
        if mode < 0:  # dedup, but global atomicOr per probe instead of LDS slices
            c.set_option("bloom_slices", 0)
            mode = -mode
        c.set_option("bloom_mode", mode)
        c.set_option("tsk_log2", tsk)
        if tsk == 4:
            c.set_option("count_split", 1)   # no second-level partitions: 16-slot sets overflow at every split
        b, o = synth.reads(seed=23, total_reads=60000, read_len=150)
        m, H = oracle.estimate_bloom(int(b.size), k)
        run_and_compare(c, oracle, (b, o), k, m, H)
        assert c.get_info("bloom_mode") == expect_mode
        if tsk in (7, 8):
            assert c.get_info("ksbits") > 0 or c.get_info("ksplit_xbits") > 0
    finally:
        c.close()


@pytest.mark.parametrize("gate", [1, 2])
def test_protocol_single_rank(oracle, gate):
    """The multi-rank protocol (partition -> owners count -> solid regions -> rank-local Bloom) driven with
    one rank and no process group: same results as the single-call pass."""
    from platanus3_amd.pipeline import Pipeline
    b, o = synth.reads(seed=31, total_reads=20000, read_len=150)
    k = 31
    m, H = oracle.estimate_bloom(int(b.size), k)
    pipe = Pipeline(device=0, rank=0, world=1, force_protocol=True)
    pipe.ctx.set_option("gate_mode", gate)
    pipe.load_reads(b, o, min_len=k)
    pipe.run_pass(k, m, H)
    c = pipe.ctx
    keys, counts = c.export_counts()
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    assert np.array_equal(keys, ekeys) and np.array_equal(counts, ecounts)
    ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    c.k, c.m, c.H = k, m, H
    assert np.array_equal(c.export_bloom(), ebloom)
    assert [capi.decode_key(r, k) for r in c.export_seeds()] == eseeds
    c.close()


def test_protocol_over_rccl_one_rank(oracle):
    """The protocol's collectives issued through torch.distributed on a REAL RCCL process group of one rank (the
    test box has one GPU): device-pointer tensor views, all_to_all_single with split sizes, all_gather_into_tensor
    and the scalar reductions go through the library bench.py uses at N > 1.  Same results as the oracle."""
    import socket
    import torch
    import torch.distributed as dist
    from platanus3_amd.pipeline import Pipeline, TorchComm
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        b, o = synth.reads(seed=32, total_reads=20000, read_len=150)
        k = 31
        m, H = oracle.estimate_bloom(int(b.size), k)
        for gate in (2, 1):
            pipe = Pipeline(device=0, rank=0, world=1, force_protocol=True, comm=TorchComm(torch.device("cuda", 0)))
            pipe.ctx.set_option("gate_mode", gate)
            pipe.load_reads(b, o, min_len=k)
            pipe.run_pass(k, m, H)
            pipe.run_pass(k, m, H)
            c = pipe.ctx
            keys, counts = c.export_counts()
            ekeys, ecounts = oracle.count_short((b, o), min_len=k)
            assert np.array_equal(keys, ekeys) and np.array_equal(counts, ecounts)
            ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
            c.k, c.m, c.H = k, m, H
            assert np.array_equal(c.export_bloom(), ebloom)
            assert [capi.decode_key(r, k) for r in c.export_seeds()] == eseeds
            assert pipe.global_distinct() == ekeys.size
            c.close()
    finally:
        dist.destroy_process_group()


def test_config2_full_size_against_reference_checksums(ctx):
    """BASELINE config #2 at FULL size (1 M x 150 bp, k=31): sha256 of the exported count table, Bloom words
    (default sizing and the '-m 2^30' H=10 run) and seed list against checksums the REAL reference produced on
    the same generator output (tests/golden/make_golden_config2.py, ~6 min of reference CPU time), plus
    size-independent properties (sum of counts = windows; every seed queries positive)."""
    import hashlib
    import json
    import os
    from util import GOLDEN
    with open(os.path.join(GOLDEN, "config2_checksums.json")) as f:
        g = json.load(f)

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    k = g["k"]
    b, o = synth.reads(seed=g["seed"], total_reads=g["n_reads"], read_len=g["read_len"])
    ctx.load_reads(b, o, min_len=k)
    st = ctx.count_short()
    assert st["n_distinct"] == g["n_distinct"] and st["n_solid"] == g["n_solid"]
    assert st["n_windows"] == g["n_windows"] and st["max_count"] == g["max_count"]
    keys, counts = ctx.export_counts()
    assert int(counts.sum()) == g["n_windows"]
    assert sha(keys) == g["keys_sha256"] and sha(counts) == g["counts_sha256"]
    for run in g["runs"]:
        ctx.make_bloom(k, run["m"], run["H"])
        words = ctx.export_bloom()
        assert int(np.unpackbits(words.view(np.uint8)).sum()) == run["bits_set"]
        assert sha(words) == run["bloom_sha256"]
        seeds = ctx.export_seeds()
        assert seeds.shape[0] == run["n_seeds"]
        sk = seeds[:, 0]
        chars = np.empty((sk.size, k), np.uint8)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        for i in range(k):
            chars[:, i] = lut[((sk >> np.uint64(2 * (k - 1 - i))) & np.uint64(3)).astype(np.int64)]
        assert sha(chars) == run["seeds_sha256"]
        # B2 property: the canonical form of every (sampled) seed is in the filter
        samp = sk[:: max(1, sk.size // 4096)]
        rc = np.zeros_like(samp)
        for i in range(k):
            rc |= (np.uint64(3) - ((samp >> np.uint64(2 * i)) & np.uint64(3))) << np.uint64(2 * (k - 1 - i))
        assert ctx.bloom_query(np.minimum(samp, rc)).all()
    # The path bench.py TIMES is p3_run_pass: a careful first pass over new reads, then the single-synchronisation
    # replay with the splits and sizes it remembered.  Both must give the digests the reference gave.
    run = g["runs"][0]
    ctx.set_option("drop_hints", 1)
    for want_replay in (0, 1, 1):
        ctx.run_pass(k, run["m"], run["H"])
        assert ctx.get_info("pass_replayed") == want_replay
        keys, counts = ctx.export_counts()
        assert sha(keys) == g["keys_sha256"] and sha(counts) == g["counts_sha256"]
        assert sha(ctx.export_bloom()) == run["bloom_sha256"]
        seeds = ctx.export_seeds()
        sk = seeds[:, 0]
        chars = np.empty((sk.size, k), np.uint8)
        for i in range(k):
            chars[:, i] = lut[((sk >> np.uint64(2 * (k - 1 - i))) & np.uint64(3)).astype(np.int64)]
        assert seeds.shape[0] == run["n_seeds"] and sha(chars) == run["seeds_sha256"]


@pytest.mark.parametrize("world,k,gate,agg", [(2, 31, 2, 1), (4, 31, 2, 1), (4, 41, 1, 1), (8, 25, 2, 1), (4, 31, 2, 0), (4, 27, 2, 2), (8, 31, 2, 3),
                                              (4, 31, 2, 10), (2, 29, 2, 11), (4, 31, 2, 20), (2, 33, 2, 21), (8, 31, 2, 20), (4, 31, 2, 21), (8, 27, 2, 21), (2, 31, 2, 31), (4, 29, 2, 31)])
def test_multirank_loopback(oracle, world, k, gate, agg):
    """The REAL multi-rank protocol (HIP kernels, n_ranks = 2/4/8) on one GPU: `world` contexts driven by threads,
    exchanging through tests/loopback_comm.py.  Union of the shards == single-rank oracle on all reads.
    agg: 1 = senders pre-aggregate their 21-mer records (forced), 0 = raw instance records, 2 = even ranks only,
    3 = the library decides from its sampled estimate of the rank's distinct keys (the default);
    10 / 11 = raw / aggregated with SKEWED reads (a third of them low-complexity repeats): the owners' estimated-capacity
    split overflows and must repeat itself exactly, on that rank only."""
    import threading
    from loopback_comm import LoopbackWorld
    from platanus3_amd.pipeline import Pipeline
    n_reads = 24000
    per = n_reads // world
    b, o = synth.reads(seed=41, total_reads=n_reads, read_len=150)
    split = 0
    if agg >= 30:   # 31 = aggregated + large-input form + SKEWED reads: the split forms' estimates overflow and must fall back
        split, agg = 3, agg - 20
    elif agg >= 20:   # 20 / 21 = raw / aggregated, owners count in groups with extra split bits (large-input form)
        split, agg = 3, agg - 20
    skew = agg >= 10
    if skew:
        agg = agg - 10   # 0 = raw, 1 = forced aggregation
        b = b.copy()
        units = [b"A", b"AC", b"ACG", b"AAC", b"ACGT"]
        for i in range(0, n_reads, 3):
            u = units[(i // 3) % len(units)]
            b[int(o[i]):int(o[i + 1])] = np.frombuffer((u * 160)[:150], dtype=np.uint8)
    m, H = oracle.estimate_bloom(n_reads * 150, k)
    lw = LoopbackWorld(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            bb = b[int(o[r * per]):int(o[(r + 1) * per])]
            oo = o[r * per:(r + 1) * per + 1] - o[r * per]
            pipe = Pipeline(device=0, rank=r, world=world, comm=lw.comm(r))
            pipe.ctx.set_option("gate_mode", gate)
            if split:
                pipe.ctx.set_option("count_split", split)
            want_agg = agg == 1 or (agg == 2 and r % 2 == 0)
            pipe.ctx.set_option("mr_aggregate", 1 if agg == 3 else (2 if want_agg else 0))
            pipe.load_reads(bb, oo, min_len=k)
            pipe.run_pass(k, m, H)
            c = pipe.ctx
            if agg != 3:
                assert c.get_info("mr_aggregated") == (1 if (want_agg and gate == 2) else 0)
            c.k, c.m, c.H = k, m, H
            keys, counts = c.export_counts()
            results[r] = (keys.copy(), counts.copy(), c.export_bloom(), [capi.decode_key(x, k) for x in c.export_seeds()],
                          c.get_info("est_fallbacks"))
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            lw.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    keys = np.concatenate([r[0] for r in results])
    counts = np.concatenate([r[1] for r in results])
    assert np.unique(keys).size == keys.size
    order = np.argsort(keys)
    assert np.array_equal(keys[order], ekeys) and np.array_equal(counts[order], ecounts)
    for r in results:
        assert np.array_equal(r[2], ebloom)
    assert sorted(set(s for r in results for s in r[3])) == eseeds
    if skew and gate == 2:
        assert sum(r[4] for r in results) >= 1


def test_beyond_2g_bases_position_chunks_and_direct_kernels():
    """Maximum sizes: 2.16 G bases (> 2^31) exceed what one partition's 30-bit window positions address.  The pass cuts
    the position space into 4 chunks that the count kernel takes as sources (fast path); with "chunks" = 1 the direct
    global-table kernels do the same work ~20x slower.  The input is X||X (the same 7.2 M reads twice), so -- without
    an oracle that large -- the result is pinned three ways: (i) by what X alone gives through the single-partition
    path: the same key set with every count doubled, hence every window passes the gate and the seeds are exactly the
    distinct first k-mers; (ii) chunked and direct runs, two independent device implementations, export identical
    tables, Bloom words and seeds; (iii) the Bloom vector has no false negatives."""
    import hashlib
    from bench import bloom_sizing

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    R, k = 7_200_000, 31
    b, o = synth.reads(seed=9, total_reads=R, read_len=150)
    c = capi.Context(0)
    c.set_option("chunks", 1)   # X alone: one partition (1.08 G positions: table-probe gate), no position chunks
    c.load_reads(b, o, k)
    c.count_short()
    assert c.get_info("count_mode") == 2 and c.get_info("chunks") == 1
    keys, counts = c.export_counts()
    c.close()
    bb = np.concatenate([b, b])
    oo = np.concatenate([o, o[1:] + o[-1]])
    assert bb.size > (1 << 31)
    m, H = bloom_sizing(int(bb.size), k)
    digests = []
    for direct in (False, True):
        c = capi.Context(0)
        if direct:
            c.set_option("chunks", 1)
        c.load_reads(bb, oo, k)
        st = c.count_short()
        assert (c.get_info("count_mode"), c.get_info("chunks")) == ((1, 1) if direct else (2, 4))
        assert st["n_windows"] == 2 * R * 130 and st["n_solid"] == st["n_distinct"] == keys.size
        k2, c2 = c.export_counts()
        assert np.array_equal(keys, k2) and np.array_equal(counts * np.uint64(2), c2)
        c.make_bloom(k, m, H)
        bs = c.bloom_stats()
        assert bs["n_passing"] == bs["n_windows_k"] == 2 * R * 120 and bs["n_seed_reads"] == 2 * R
        seeds = c.export_seeds()
        digests.append((sha(c.export_bloom()), sha(seeds)))
        if not direct:
            lut = np.zeros(256, np.uint64)
            lut[ord("C")], lut[ord("G")], lut[ord("T")] = 1, 2, 3
            starts = o[:-1].astype(np.int64)
            first = np.zeros(R, np.uint64)
            for i in range(k):
                first = (first << np.uint64(2)) | lut[b[starts + i]]
            assert np.array_equal(np.unique(first), seeds[:, 0])
            samp = first[::97]
            rc = np.zeros_like(samp)
            for i in range(k):
                rc |= (np.uint64(3) - ((samp >> np.uint64(2 * i)) & np.uint64(3))) << np.uint64(2 * (k - 1 - i))
            assert c.bloom_query(np.minimum(samp, rc)).all()
        c.close()
    assert digests[0] == digests[1]


@pytest.mark.parametrize("R,L,k,ppm", [(2_000_000, 150, 31, 5000), (4000, 10_000, 63, 10_000), (300_000, 150, 21, 5000),
                                       (300_000, 150, 41, 5000), (300_000, 150, 51, 5000), (300_000, 150, 61, 5000)])
def test_baseline_shapes_partitioned_equals_direct(R, L, k, ppm):
    """BASELINE.json's other configurations, scaled to seconds (#3's per-GPU shard shape, #4's long reads at k=63 with
    2-word keys, #5's k sweep): the partitioned/dedup path and the direct global-table path are two independent
    device implementations of the same stage and must export identical tables, Bloom words and seeds."""
    import hashlib
    from bench import bloom_sizing

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    b, o = synth.reads(seed=3, total_reads=R, read_len=L, sub_rate_ppm=ppm)
    m, H = bloom_sizing(int(b.size), k)
    res = []
    for direct in (False, True):
        c = capi.Context(0)
        if direct:
            c.set_option("count_mode", 1)
            c.set_option("bloom_mode", 1)
        c.load_reads(b, o, k)
        c.run_pass(k, m, H)
        assert c.get_info("count_mode") == (1 if direct else 2)
        keys, counts = c.export_counts()
        res.append((sha(keys), sha(counts), sha(c.export_bloom()), sha(c.export_seeds()), c.bloom_stats()))
        assert int(counts.sum()) == R * (L - 20)
        c.close()
    assert res[0] == res[1]


@pytest.mark.parametrize("k", [31, 45])
def test_skewed_keys_overflow_estimated_segments_and_retry_exactly(oracle, k):
    """Estimated-capacity partitions assume hash-spread keys.  Half of these reads are low-complexity repeats
    (poly-A, (AC)n, (ACG)n, (ACGT)n ...): a handful of 21-mers / k-mers with tens of thousands of instances each, all
    landing in the same few segments, which must overflow the estimate, raise the flag and repeat the stage with
    exact offsets -- with the same results as the oracle, and as a run that was exact from the start."""
    rng = np.random.default_rng(7)
    b0, o0 = synth.reads(seed=17, total_reads=12000, read_len=150)
    units = [b"A", b"AC", b"ACG", b"ACGT", b"AAC", b"AGGT", b"T", b"CCG"]
    reps = []
    for i in range(12000):
        u = units[i % len(units)]
        s = (u * (150 // len(u) + 2))[rng.integers(0, len(u)):][:150]
        reps.append(np.frombuffer(s, dtype=np.uint8))
    reads = [b0[int(o0[i]):int(o0[i + 1])] for i in range(12000)] + reps
    order = rng.permutation(len(reads))
    b = np.concatenate([reads[i] for i in order])
    o = np.concatenate([[0], np.cumsum([reads[i].size for i in order])]).astype(np.uint64)
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    run_and_compare(c, oracle, (b, o), k, m, H)          # default sizing: estimated first
    assert c.get_info("count_mode") == 2 and c.get_info("bloom_mode") == 3
    assert c.get_info("est_fallbacks") >= 1               # ... and at least one stage had to fall back
    assert c.get_info("est_level") == 3                   # ... through the boost levels to exact offsets
    keys, counts = c.export_counts()
    words = c.export_bloom()
    c.close()
    c = capi.Context(0)
    c.set_option("part_sizing", 1)                         # exact from the start
    c.load_reads(b, o, k)
    c.run_pass(k, m, H)
    assert c.get_info("est_fallbacks") == 0
    k2, c2 = c.export_counts()
    assert np.array_equal(keys, k2) and np.array_equal(counts, c2) and np.array_equal(words, c.export_bloom())
    c.close()


def test_mild_skew_is_absorbed_by_a_boosted_estimate(oracle):
    """A poly-A stretch of 25 bases in 3 % of 200 k reads: ONE 21-mer with 30,000 instances, about 120 extra records
    in each segment of its bucket (mean 100, estimated capacity 192).  The graded response repeats the stage with
    larger estimated capacities (level 1: + 50 % of the mean, level 2: + 150 %) instead of giving the estimates up;
    the results equal the oracle's, and the following passes run at the level that was found."""
    k = 31
    b, o = synth.reads(seed=41, total_reads=200000, read_len=150)
    b = b.copy()
    rng = np.random.default_rng(3)
    for r in rng.choice(200000, size=6000, replace=False):
        b[int(o[r]) + 50:int(o[r]) + 75] = ord("A")
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    run_and_compare(c, oracle, (b, o), k, m, H)
    assert c.get_info("est_fallbacks") >= 1 and c.get_info("est_level") in (1, 2)
    assert c.get_info("count_mode") == 2 and c.get_info("bloom_mode") == 3
    n = c.get_info("est_fallbacks")
    keys, counts = c.export_counts()
    words = c.export_bloom()
    c.run_pass(k, m, H)
    c.run_pass(k, m, H)
    assert c.get_info("est_fallbacks") == n            # no further overflow at the level in use
    k2, c2 = c.export_counts()
    assert np.array_equal(keys, k2) and np.array_equal(counts, c2) and np.array_equal(words, c.export_bloom())
    c.close()
    c = capi.Context(0)
    c.set_option("est_boost", 0)                        # without the levels: straight to exact offsets
    c.load_reads(b, o, k)
    c.run_pass(k, m, H)
    assert c.get_info("est_level") == 3
    k2, c2 = c.export_counts()
    assert np.array_equal(keys, k2) and np.array_equal(counts, c2) and np.array_equal(words, c.export_bloom())
    c.close()


@pytest.mark.parametrize("k", [31, 41])
def test_heavy_key_gets_its_own_cell_capacity(oracle, k):
    """A 40-base poly-A stretch in 950 of 150 k reads: one 21-mer with 19,000 instances.  The first-level partition takes
    it at boost level 2 (its bucket's segments double), but under the second-level partition (32 groups, forced here;
    automatic for large inputs) all instances land in ONE of 32,768 cells -- 30x the cell's share, which no boost of a
    uniform capacity covers.  From the first overflow on the cells' capacities come from a sample of the input
    (k_cell_sample / k_var_caps): the run stays on the estimated path, equals the oracle, and without that
    (var_caps = 0) ends on exact offsets with the same results."""
    n = 150000
    b, o = synth.reads(seed=43, total_reads=n, read_len=150)
    b = b.copy()
    rng = np.random.default_rng(5)
    starts = o[:-1][rng.choice(n, size=950, replace=False)].astype(np.int64) + 50
    b[(starts[:, None] + np.arange(40)[None, :]).ravel()] = ord("A")
    m, H = oracle.estimate_bloom(int(b.size), k)
    res = []
    for var in (1, 0):
        c = capi.Context(0)
        c.set_option("count_split", 6)
        c.set_option("kmer_split", 4)
        c.set_option("var_caps", var)
        run_and_compare(c, oracle, (b, o), k, m, H)
        assert c.get_info("est_fallbacks") >= 1
        if var:
            assert c.get_info("est_level") in (1, 2) and c.get_info("split_xbits") == 5
        else:
            assert c.get_info("est_level") == 3
        c.run_pass(k, m, H)
        c.run_pass(k, m, H)
        keys, counts = c.export_counts()
        res.append((keys, counts, c.export_bloom()))
        c.close()
    assert all(np.array_equal(x, y) for x, y in zip(res[0], res[1]))
    assert int(res[0][1].max()) >= 15000


def test_extreme_key_in_position_chunks_uses_exact_first_level(oracle):
    """5 % of the reads carry a 40-base poly-A stretch: one 21-mer with 150,000 instances, 8x its bucket's share, so the
    first-level partition overflows at every boost level.  In the position-chunk form (forced: 4 chunks) the chunks'
    first-level partitions then take exact offsets (histogram + scan) and the second-level ones sampled per-cell
    capacities -- the pass stays chunked and partitioned instead of dropping to the direct kernels, and equals the
    oracle."""
    n, k = 150000, 31
    b, o = synth.reads(seed=44, total_reads=n, read_len=150)
    b = b.copy()
    rng = np.random.default_rng(6)
    starts = o[:-1][rng.choice(n, size=n // 20, replace=False)].astype(np.int64) + 50
    b[(starts[:, None] + np.arange(40)[None, :]).ravel()] = ord("A")
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    c.set_option("chunks", 4)
    run_and_compare(c, oracle, (b, o), k, m, H)
    assert c.get_info("est_level") == 3 and c.get_info("chunks") == 4
    assert c.get_info("count_mode") == 2 and c.get_info("gate_mode") == 2
    n_fb = c.get_info("est_fallbacks")
    keys, counts = c.export_counts()
    words = c.export_bloom()
    c.run_pass(k, m, H)
    assert c.get_info("chunks") == 4 and c.get_info("count_mode") == 2 and c.get_info("est_fallbacks") == n_fb
    k2, c2 = c.export_counts()
    assert np.array_equal(keys, k2) and np.array_equal(counts, c2) and np.array_equal(words, c.export_bloom())
    assert int(counts.max()) >= 100000
    c.close()


def test_speculative_pass_survives_stale_hints(oracle):
    """p3_run_pass re-uses what the previous pass learnt (table splits, passing-window count, non-solid count) and
    checks only at its single synchronisation point whether that still held.  One context is driven through changes
    of k, of the Bloom size and of the reads (larger, tiny, skewed); every pass must still equal the oracle."""
    def check(c, b, o, k, m, H):
        keys, counts = c.export_counts()
        ekeys, ecounts = oracle.count_short((b, o), min_len=k)
        ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
        assert np.array_equal(keys, ekeys) and np.array_equal(counts, ecounts)
        assert np.array_equal(c.export_bloom(), ebloom)
        c.k = k
        assert seeds_as_strings(c) == eseeds

    c = capi.Context(0)
    a = synth.reads(seed=21, total_reads=8000, read_len=150)
    big = synth.reads(seed=22, total_reads=40000, read_len=150)
    tiny = synth.reads(seed=23, total_reads=40, read_len=150)
    rep = np.frombuffer((b"ACGGT" * 40)[:150] * 3000, dtype=np.uint8)
    skew = (np.concatenate([a[0], rep]), np.concatenate([a[1], a[1][-1] + np.arange(1, 3001, dtype=np.uint64) * np.uint64(150)]))
    plan = [(a, 31, 0), (a, 31, 0), (a, 21, 0), (a, 31, 1 << 22), (big, 31, 0), (big, 31, 0), (big, 47, 0), (tiny, 31, 0),
            (skew, 31, 0), (skew, 31, 0), (a, 63, 0), (a, 31, 0)]
    loaded = None
    for reads, k, m_in in plan:
        b, o = reads
        if loaded is not reads or c.min_len != k:
            c.load_reads(b, o, min_len=k)
            loaded, c.min_len = reads, k
        m, H = oracle.estimate_bloom(int(b.size), k) if m_in == 0 else (m_in, 10)
        c.run_pass(k, m, H)
        check(c, b, o, k, m, H)
    c.close()


@pytest.mark.parametrize("split,k", [(2, 31), (3, 31), (5, 25), (6, 41)])
def test_second_level_partition(oracle, split, k):
    """count_split = x+1 forces the large-input form of the count stage (k_split: level-1 buckets split once more
    into 1024 cells per group of 1024 >> x buckets, counted group by group) on an input small enough for the oracle."""
    b, o = synth.reads(seed=31 + split, total_reads=30000, read_len=150)
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    c.set_option("count_split", split)
    run_and_compare(c, oracle, (b, o), k, m, H)
    assert c.get_info("split_xbits") == split - 1 and c.get_info("count_mode") == 2 and c.get_info("gate_mode") == 2
    c.run_pass(k, m, H)   # careful pass, then the speculative one with the remembered split
    c.run_pass(k, m, H)
    assert c.get_info("split_xbits") == split - 1
    keys, counts = c.export_counts()
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    assert np.array_equal(keys, ekeys) and np.array_equal(counts, ecounts)
    ebloom, _, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    assert np.array_equal(c.export_bloom(), ebloom)
    c.close()


@pytest.mark.parametrize("chunks,k", [(2, 31), (4, 25), (8, 31), (4, 45), (16, 31), (16, 63)])
def test_position_chunks_small(oracle, chunks, k):
    """Inputs beyond 2^30 window positions are cut into 2, 4 or 8 position ranges that the count kernel takes as
    sources; gate and Bloom stages then run range by range.  "chunks" forces that form on an input small enough
    for the oracle (reads of mixed length, so that reads straddle the range boundaries)."""
    rng = np.random.default_rng(chunks * 100 + k)
    b0, o0 = synth.reads(seed=50 + chunks, total_reads=24000, read_len=150)
    lens = rng.integers(k, 151, size=24000)
    reads = [b0[int(o0[i]):int(o0[i]) + int(lens[i])] for i in range(24000)]
    b = np.concatenate(reads)
    o = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    c.set_option("chunks", chunks)
    # kmer_collect: 0 = every chunk probes its own k-mers, 2 = the chunks' k-mers are collected and deduplicated once
    # more before the probes (the default, 1, does that from four chunks on)
    # kmer_group: at most that many chunks per collected list (in production: as many as fit in device memory)
    c.set_option("kmer_collect", {2: 2, 4: 1, 8: 0, 16: 2}[chunks])
    if chunks == 16:
        c.set_option("kmer_group", 5)
    if chunks in (4, 16):
        # the count stage's partition buffer is lent to the k-mer stage's scratch buffers (in production: only when
        # device memory is short) -- every pass carves them anew
        c.set_option("lend", 1)
    run_and_compare(c, oracle, (b, o), k, m, H)
    assert c.get_info("chunks") == chunks and c.get_info("count_mode") == 2 and c.get_info("gate_mode") == 2
    assert c.get_info("est_fallbacks") == 0
    # (16 chunks, at most 5 per group: the list's fixed slack dominates at this size, so the groups come out larger)
    assert c.get_info("kmer_groups") in {2: (1,), 4: (1,), 8: (0,), 16: (2, 3, 4)}[chunks]
    if chunks == 4:
        # the count table outputs are sized from an estimate of the distinct keys; far too small a table must be
        # noticed by the count kernel (no write past it) and grown by doubling
        c.set_option("distinct_cap", 3000)
    c.run_pass(k, m, H)
    c.run_pass(k, m, H)
    assert c.get_info("chunks") == chunks and c.get_info("est_fallbacks") == 0
    # once a count of these reads has told how many distinct keys there are, the chunks' cells are counted on 4-byte
    # quotient slots with the chunk of every record as its source tag (p3_kernels_countq.h)
    assert c.get_info("quotient_tables") > 0 and c.get_info("quotient_redone") == 0
    keys, counts = c.export_counts()
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    assert np.array_equal(keys, ekeys) and np.array_equal(counts, ecounts)
    assert np.array_equal(c.export_bloom(), ebloom)
    c.k = k
    assert seeds_as_strings(c) == eseeds
    c.close()


@pytest.mark.parametrize("split,k", [(2, 31), (4, 27), (6, 41), (3, 63)])
def test_kmer_second_level_partition(oracle, split, k):
    """kmer_split = x+1 forces the large-input form of the dedup stage: the passing k-mer records (8 or 16 bytes) are
    partitioned a second time (k_split) into 1024 cells per group of 1024 >> x buckets and every cell is deduplicated
    by one workgroup, instead of 2^s workgroups re-reading every bucket."""
    n, L = (20000, 150) if k < 60 else (400, 5000)
    b, o = synth.reads(seed=70 + split, total_reads=n, read_len=L, sub_rate_ppm=5000 if k < 60 else 10000)
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    c.set_option("kmer_split", split)
    run_and_compare(c, oracle, (b, o), k, m, H)
    assert c.get_info("ksplit_xbits") == split - 1 and c.get_info("bloom_mode") == 3
    c.run_pass(k, m, H)   # careful pass, then the speculative one with the remembered split
    c.run_pass(k, m, H)
    assert c.get_info("ksplit_xbits") == split - 1
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    assert np.array_equal(c.export_bloom(), ebloom)
    c.k = k
    assert seeds_as_strings(c) == eseeds
    c.close()


@pytest.mark.parametrize("k,chunks", [(31, 0), (31, 4), (63, 0)])
def test_kmer_split_learnt_load(oracle, k, chunks):
    """The second-level partition of the k-mer records starts at the smallest split whose LDS sets hold their keys and
    REMEMBERS a deeper one for the next chunk / pass (kmer_load_pct percent of a set's slots, default 50): small sets
    (tsk_log2 = 7, 5) force the split form on a small input; a 5 % target must end deeper than the 100 % one, and every
    pass -- the careful first, the replayed ones with the remembered split -- gives the reference's vector and seeds."""
    b, o = synth.reads(seed=77, total_reads=60000, read_len=150)
    m, H = oracle.estimate_bloom(int(b.size), k)
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    depth = {}
    for pct in (100, 5):
        c = capi.Context(0)
        try:
            c.set_option("tsk_log2", 5 if chunks else 7)   # (a chunk holds a quarter of the keys: smaller sets still)
            c.set_option("kmer_load_pct", pct)
            if chunks:
                c.set_option("chunks", chunks)
            c.load_reads(b, o, k)
            for _ in range(3):
                c.run_pass(k, m, H)
                assert c.get_info("bloom_mode") == 3
                assert np.array_equal(c.export_bloom(), ebloom)
                c.k = k
                assert seeds_as_strings(c) == eseeds
            depth[pct] = c.get_info("ksplit_xbits")
            assert depth[pct] > 0
        finally:
            c.close()
    assert depth[5] > depth[100]


@pytest.mark.parametrize("seed", range(32))
def test_random_configurations(oracle, seed):
    """Randomised sweep over input shapes AND path options: every combination must equal the oracle, on the first
    pass and on a repeated (speculative) one.  Inputs: ragged read lengths (some shorter than k, unknown characters,
    lowercase) or fixed-length synthetic reads; k anywhere in [21,64]; Bloom sizes from a few hundred bits to the
    estimated size; H in [1,24].  Options: forced position chunks, second-level partitions of both stages, LDS table
    sizes that force sub-bucket splits, both gates, exact / estimated sizing, slices on / off, collection and lending."""
    rng = np.random.default_rng(1000 + seed + FUZZ_OFFSET)
    for case in range(6):
        k = int(rng.integers(21, 65))
        if rng.random() < 0.5:
            reads = random_reads(rng, int(rng.integers(1, 2500)), lo=int(rng.integers(1, 40)), hi=int(rng.integers(60, 500)),
                                 genome_len=int(rng.integers(500, 30000)))
            b, o = flat(reads)
        else:
            b, o = synth.reads(seed=int(rng.integers(1, 1 << 30)), total_reads=int(rng.integers(50, 6000)),
                               read_len=int(rng.integers(k + 1, 400)), sub_rate_ppm=int(rng.choice([0, 5000, 30000])))
        lens = np.diff(o.astype(np.int64))
        all_bases = int(lens[lens >= k].sum())
        if rng.random() < 0.5 and all_bases > 0:
            m, H = oracle.estimate_bloom(all_bases, k)
            if m == 0:
                m, H = 1009, 3
        else:
            m, H = int(rng.choice([257, 4099, 65537, 1 << 20, (1 << 22) + 3])), int(rng.integers(1, 25))
        opts = {}
        if rng.random() < 0.4:
            opts["chunks"] = int(rng.choice([2, 4, 8]))
            opts["kmer_collect"] = int(rng.choice([0, 1, 2]))
            opts["lend"] = int(rng.integers(0, 2))
            opts["kmer_group"] = int(rng.choice([0, 2, 3]))
        if rng.random() < 0.3:
            opts["count_split"] = int(rng.choice([1, 2, 3, 4]))
        if rng.random() < 0.3:
            opts["kmer_split"] = int(rng.choice([2, 3, 5]))
        if rng.random() < 0.3:
            opts["ts_log2"] = int(rng.choice([6, 8, 10]))
        if rng.random() < 0.3:
            opts["tsk_log2"] = int(rng.choice([6, 8, 10]))
        if rng.random() < 0.3:
            opts["gate_mode"] = int(rng.choice([1, 2]))
        if rng.random() < 0.2:
            opts["part_sizing"] = 1
        if rng.random() < 0.2:
            opts["bloom_slices"] = 0
        if rng.random() < 0.2:
            opts["est_boost"] = 0
        if (seed + case) % 3 == 0:  # (not drawn from rng: the cases of the sweep stay the ones they were)
            opts["kmer_load_pct"] = (5, 20, 100)[(seed * 7 + case) % 3]
        for name in [x for x in os.environ.get("P3_FUZZ_DROP", "").split(",") if x]:  # (bisecting a failing case by its options)
            opts.pop(name, None)
        for kv in [x for x in os.environ.get("P3_FUZZ_SET", "").split(",") if x]:
            opts[kv.split("=")[0]] = int(kv.split("=")[1])
        if os.environ.get("P3_FUZZ_ONLY_CASE") and int(os.environ["P3_FUZZ_ONLY_CASE"]) != case:
            continue
        if os.environ.get("P3_FUZZ_VERBOSE"):
            print("seed %d case %d: k=%d m=%d H=%d reads=%d bases=%d opts=%r" % (seed, case, k, m, H, o.size - 1, b.size, opts), flush=True)
        c = capi.Context(0)
        try:
            for name, v in opts.items():
                c.set_option(name, v)
            words, _ = run_and_compare(c, oracle, (b, o), k, m, H)
            keys, counts = c.export_counts()
            c.run_pass(k, m, H)
            c.run_pass(k, m, H)
            k2, c2 = c.export_counts()
            assert np.array_equal(keys, k2) and np.array_equal(counts, c2), (seed, case, k, m, H, opts)
            assert np.array_equal(words, c.export_bloom()), (seed, case, k, m, H, opts)
        except Exception as e:
            raise AssertionError("seed %d case %d: k=%d m=%d H=%d reads=%d opts=%r: %s: %s" %
                                 (seed, case, k, m, H, o.size - 1, opts, type(e).__name__, e))
        finally:
            c.close()


@pytest.mark.parametrize("seed", range(48))
def test_multirank_random_configurations(oracle, seed):
    """Randomised sweep of the multi-rank protocol with the real kernels (N contexts on one GPU, loopback exchange):
    world size, k, read shapes (ragged reads shorter than k included), Bloom size, gate form, per-rank aggregation
    choice and table / split options.  Two passes; the union of the shards equals the oracle on all reads."""
    import threading
    from loopback_comm import LoopbackWorld
    from platanus3_amd.pipeline import Pipeline
    rng = np.random.default_rng(5000 + seed + FUZZ_OFFSET)
    world = int(rng.choice([2, 4, 8]))
    k = int(rng.integers(21, 65))
    if rng.random() < 0.5:
        reads = random_reads(rng, int(rng.integers(world, 4000)), lo=int(rng.integers(1, 40)), hi=int(rng.integers(60, 400)),
                             genome_len=int(rng.integers(500, 30000)))
        b, o = flat(reads)
    else:
        b, o = synth.reads(seed=int(rng.integers(1, 1 << 30)), total_reads=int(rng.integers(world, 12000)),
                           read_len=int(rng.integers(k + 1, 300)), sub_rate_ppm=int(rng.choice([0, 5000, 30000])))
    n_reads = (o.size - 1) // world * world
    per = n_reads // world
    b, o = b[:max(1, int(o[n_reads]))], o[:n_reads + 1]
    lens = np.diff(o.astype(np.int64))
    all_bases = int(lens[lens >= k].sum())
    if rng.random() < 0.5 and all_bases > 0:
        m, H = oracle.estimate_bloom(all_bases, k)
        if m == 0:
            m, H = 1009, 3
    else:
        m, H = int(rng.choice([257, 4099, 65537, 1 << 20, (1 << 22) + 3])), int(rng.integers(1, 25))
    common = {"gate_mode": int(rng.choice([1, 2]))}
    if rng.random() < 0.3:
        common["count_split"] = int(rng.choice([1, 2, 3]))
    if rng.random() < 0.3:
        common["ts_log2"] = int(rng.choice([6, 8, 10]))
    per_rank = [{"mr_aggregate": int(rng.choice([0, 1, 2])),
                 **({"tsk_log2": int(rng.choice([6, 8, 10]))} if rng.random() < 0.3 else {}),
                 **({"kmer_split": int(rng.choice([2, 3]))} if rng.random() < 0.3 else {}),
                 **({"bloom_slices": 0} if rng.random() < 0.2 else {})} for _ in range(world)]
    what = "seed %d: world=%d k=%d m=%d H=%d reads=%d common=%r per_rank=%r" % (seed, world, k, m, H, n_reads, common, per_rank)
    lw = LoopbackWorld(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            bb = b[int(o[r * per]):max(int(o[(r + 1) * per]), int(o[r * per]) + 1)]
            oo = o[r * per:(r + 1) * per + 1] - o[r * per]
            pipe = Pipeline(device=0, rank=r, world=world, comm=lw.comm(r))
            for name, v in {**common, **per_rank[r]}.items():
                pipe.ctx.set_option(name, v)
            pipe.load_reads(bb, oo, min_len=k)
            pipe.run_pass(k, m, H)
            pipe.run_pass(k, m, H)
            c = pipe.ctx
            c.k, c.m, c.H = k, m, H
            keys, counts = c.export_counts()
            results[r] = (keys.copy(), counts.copy(), c.export_bloom(), [capi.decode_key(x, k) for x in c.export_seeds()])
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            lw.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, (what, errors)
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    keys = np.concatenate([r[0] for r in results])
    counts = np.concatenate([r[1] for r in results])
    assert np.unique(keys).size == keys.size, what
    order = np.argsort(keys)
    assert np.array_equal(keys[order], ekeys) and np.array_equal(counts[order], ecounts), what
    for r in results:
        assert np.array_equal(r[2], ebloom), what
    assert sorted(set(s for r in results for s in r[3])) == eseeds, what


@pytest.mark.parametrize("seed", range(4))
def test_random_configurations_medium(oracle, seed):
    """The same sweep at 50-250 k reads (segment means of 25-130 records, LDS tables that really fill): forced chunks,
    splits, small tables, skew (a poly-A stretch in some reads) -- against the oracle, two more passes after the first."""
    rng = np.random.default_rng(9000 + seed + FUZZ_OFFSET)
    k = int(rng.choice([21, 25, 31, 33, 41, 55, 63]))
    L = int(rng.choice([100, 150, 250]))
    n = int(rng.integers(50_000, 250_000) * 150 // L)
    b, o = synth.reads(seed=int(rng.integers(1, 1 << 30)), total_reads=n, read_len=L, sub_rate_ppm=int(rng.choice([1000, 5000, 20000])))
    if rng.random() < 0.4:
        b = b.copy()
        for r in rng.choice(n, size=int(n * rng.choice([0.005, 0.03])), replace=False):
            b[int(o[r]) + 30:int(o[r]) + 30 + int(rng.integers(22, 60))] = ord("A")
    m, H = oracle.estimate_bloom(int(b.size), k)
    if rng.random() < 0.3:
        m, H = int(rng.choice([(1 << 24) + 7, 1 << 27])), 10
    opts = {}
    if rng.random() < 0.5:
        opts["chunks"] = int(rng.choice([2, 4, 8, 16]))
        opts["kmer_collect"] = int(rng.choice([0, 1, 2]))
        opts["lend"] = int(rng.integers(0, 2))
        opts["kmer_group"] = int(rng.choice([0, 2, 3]))
    if rng.random() < 0.4:
        opts["count_split"] = int(rng.choice([1, 2, 3, 5]))
    if rng.random() < 0.4:
        opts["kmer_split"] = int(rng.choice([2, 3, 5]))
    if rng.random() < 0.4:
        opts["ts_log2"] = int(rng.choice([8, 10, 12]))
    if rng.random() < 0.4:
        opts["tsk_log2"] = int(rng.choice([8, 10, 12]))
    if rng.random() < 0.2:
        opts["gate_mode"] = 1
    if rng.random() < 0.15:
        opts["bloom_slices"] = 0
    what = "seed %d: k=%d L=%d reads=%d m=%d H=%d opts=%r" % (seed, k, L, n, m, H, opts)
    c = capi.Context(0)
    try:
        for name, v in opts.items():
            c.set_option(name, v)
        words, _ = run_and_compare(c, oracle, (b, o), k, m, H)
        keys, counts = c.export_counts()
        c.run_pass(k, m, H)
        c.run_pass(k, m, H)
        k2, c2 = c.export_counts()
        assert np.array_equal(keys, k2) and np.array_equal(counts, c2) and np.array_equal(words, c.export_bloom())
    except Exception as e:
        raise AssertionError("%s: %s: %s" % (what, type(e).__name__, e))
    finally:
        c.close()


def test_count_overflow_is_reported_not_truncated(oracle):
    """G1 guard: the reference's RMQ passes the 21-mer counts through an int (src/MakeBloomFilter.cpp:11,18), so a count
    >= 2^31 would be truncated there; the ABI reports P3_ECOUNT_OVERFLOW instead of diverging silently.  Reached cheaply
    through the multi-rank protocol's pre-aggregated records (weight = the sender's local count): two records of one key
    are patched to weight 2^31 - 1 each on their way to the owner."""
    import threading
    import torch
    from loopback_comm import LoopbackWorld, LoopbackComm
    from platanus3_amd.pipeline import Pipeline
    world, k, n_reads = 2, 31, 8000
    per = n_reads // world
    b, o = synth.reads(seed=43, total_reads=n_reads, read_len=150)
    m, H = oracle.estimate_bloom(n_reads * 150, k)
    lw = LoopbackWorld(world)

    class PatchComm(LoopbackComm):
        """rank 0's comm: after the record blob and its two descriptor tables have arrived, raise two weights"""
        def __init__(self, w, rank):
            super().__init__(w, rank)
            self.seen = []

        def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
            super().all_to_all_single(out, inp, out_splits, in_splits)
            self.seen.append(out)
            if len(self.seen) == 4:  # counts, records, ghist, segoff
                part, ghist, segoff = self.seen[1], self.seen[2], self.seen[3]
                g = ghist.cpu().numpy()
                so = segoff.cpu().numpy()
                s = int(np.nonzero(g[: g.size // world] >= 2)[0][0])  # a segment of source 0 with two records
                idx = int(so[s]) - int(so[0])
                v = part[idx:idx + 2].cpu().numpy().view(np.uint64)
                low = v[0] & np.uint64(0xffffffff)
                heavy = (np.uint64(1) << np.uint64(63)) | (np.uint64(0x7fffffff) << np.uint64(32)) | low
                part[idx:idx + 2] = torch.from_numpy(np.array([heavy, heavy], np.uint64).view(np.int64)).to(part.device)
                torch.cuda.synchronize()

    errors = [None] * world

    def worker(r):
        try:
            bb = b[int(o[r * per]):int(o[(r + 1) * per])]
            oo = o[r * per:(r + 1) * per + 1] - o[r * per]
            comm = PatchComm(lw, r) if r == 0 else lw.comm(r)
            pipe = Pipeline(device=0, rank=r, world=world, comm=comm)
            pipe.ctx.set_option("mr_aggregate", 2)
            pipe.load_reads(bb, oo, min_len=k)
            pipe.run_pass(k, m, H)
        except Exception as e:  # noqa: BLE001
            errors[r] = e
            lw.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    e0 = errors[0]
    assert isinstance(e0, capi.P3Error) and e0.code == 6, errors  # P3_ECOUNT_OVERFLOW


def test_quotient_slot_count_leaves_heavy_keys_to_the_wide_kernel(oracle):
    """The count stage's 4-byte quotient slots (p3_kernels_countq.h) hold 10-bit counters: a 21-mer with 1024 instances
    or more wraps one, the table's sum check sees it, the table is flagged and k_part_count -- launched right behind,
    idle for every other table -- counts it.  Reads with a poly-A stretch (one key with ~40,000 instances), (AC)n and
    (ACG)n repeats among ordinary reads: same table, vector and seeds as the oracle and as a run on 8-byte slots only;
    ordinary reads alone flag nothing."""
    rng = np.random.default_rng(11)
    b0, o0 = synth.reads(seed=23, total_reads=60000, read_len=150)
    reads = [b0[int(o0[i]):int(o0[i + 1])].copy() for i in range(60000)]
    for i in range(0, 60000, 30):  # 2000 reads: a 40-base stretch of a repeat unit in the middle
        u = (b"A", b"AC", b"ACG")[(i // 30) % 3]
        reads[i][50:90] = np.frombuffer((u * 40)[:40], dtype=np.uint8)
    b = np.concatenate(reads)
    o = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.uint64)
    k = 31
    m, H = oracle.estimate_bloom(int(b.size), k)
    c = capi.Context(0)
    try:
        run_and_compare(c, oracle, (b, o), k, m, H)
        assert c.get_info("quotient_tables") > 0 and c.get_info("quotient_redone") >= 1
        assert c.count_stats()["max_count"] >= 1024
        keys, counts = c.export_counts()
        words = c.export_bloom()
        c.run_pass(k, m, H)  # the replayed pass takes the same route
        k3, c3 = c.export_counts()
        assert np.array_equal(keys, k3) and np.array_equal(counts, c3) and np.array_equal(words, c.export_bloom())
    finally:
        c.close()
    c = capi.Context(0)
    try:
        c.set_option("count_slots", 1)  # 8-byte slots only
        c.load_reads(b, o, k)
        c.run_pass(k, m, H)
        assert c.get_info("quotient_tables") == 0
        k2, c2 = c.export_counts()
        assert np.array_equal(keys, k2) and np.array_equal(counts, c2) and np.array_equal(words, c.export_bloom())
        c.load_reads(b0, o0, k)  # ordinary reads: nothing to redo
        c.set_option("count_slots", 0)
        c.run_pass(k, m, H)
        assert c.get_info("quotient_tables") > 0 and c.get_info("quotient_redone") == 0
    finally:
        c.close()


@pytest.mark.gpu
def test_probe_records_of_two_and_four_bytes_build_the_same_vector(oracle):
    """Vectors of up to 2^26 bits are built from 2-byte probe records (the in-slice offset of a 2^16-bit slice), larger
    ones and the option probe_records=1 from 4-byte records: same vector as the oracle either way, and the vector just
    above the limit takes the 4-byte form by itself."""
    b, o = synth.reads(seed=31, total_reads=40000, read_len=150)
    k = 31
    m, H = oracle.estimate_bloom(int(b.size), k)
    assert m < (1 << 26)
    c = capi.Context(0)
    try:
        run_and_compare(c, oracle, (b, o), k, m, H)
        assert c.get_info("probe_record_bytes") == 2
        w2 = c.export_bloom()
        c.set_option("probe_records", 1)
        c.run_pass(k, m, H)
        assert c.get_info("probe_record_bytes") == 4
        assert np.array_equal(w2, c.export_bloom())
        c.set_option("probe_records", 0)
        m_big = (1 << 26) + 12345  # 1025 slices of 2^16 bits do not fit one sweep: slices of 2^17 bits, 4-byte records
        run_and_compare(c, oracle, (b, o), k, m_big, H)
        assert c.get_info("probe_record_bytes") == 4
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [33, 47, 63, 64])
def test_sixteen_byte_kmer_sets_on_both_slot_layouts(oracle, k):
    """k-mers of 33...63 bases are deduplicated on 16-byte slots whose claim protocol lives in the two spare bits of the
    high key word (k_kmer_dedup16); k = 64 has no spare bits and stays on key + state word (k_kmer_dedup<2>), which the
    option kmer_slots=1 selects for the others too: same vector and seeds as the oracle from both, plain form and
    second-level partition."""
    b, o = synth.reads(seed=80 + k, total_reads=600, read_len=4000, sub_rate_ppm=10000)
    m, H = oracle.estimate_bloom(int(b.size), k)
    for split in (0, 3):
        for slots in (0, 1):
            c = capi.Context(0)
            try:
                c.set_option("kmer_slots", slots)
                if split:
                    c.set_option("kmer_split", split)
                c.set_profiling(True)
                words, _ = run_and_compare(c, oracle, (b, o), k, m, H)
                c.run_pass(k, m, H)
                assert np.array_equal(words, c.export_bloom())
                st = c.kernel_stats()
                used16 = st.get("k_kmer_dedup16", {}).get("launches", 0) > 0
                used20 = st.get("k_kmer_dedup", {}).get("launches", 0) > 0
                assert used16 == (slots == 0 and k <= 63) and used20 == (not used16)
            finally:
                c.close()


@pytest.mark.gpu
def test_probe_sweeps_of_sixteen_byte_keys_share_their_hashes(oracle):
    """A vector of more than 2^32 bits is built in several sweeps over the distinct keys; with 16-byte keys the first
    sweep leaves (h1, h2) in place of every key and the others read them back instead of hashing again (k_pos_scatter,
    hmode).  Same vector as the oracle, and as with every sweep hashing for itself (sweep_hashes=1)."""
    b, o = synth.reads(seed=91, total_reads=300, read_len=2000, sub_rate_ppm=10000)
    k, H = 45, 5
    m = (1 << 32) + 12345
    c = capi.Context(0)
    try:
        c.set_profiling(True)
        words, _ = run_and_compare(c, oracle, (b, o), k, m, H)
        assert c.kernel_stats()["k_pos_scatter"]["launches"] >= 2  # (two sweeps of <= 1024 buckets of four LDS slices)
        c.run_pass(k, m, H)
        assert np.array_equal(words, c.export_bloom())
        c.set_option("sweep_hashes", 1)
        c.run_pass(k, m, H)
        assert np.array_equal(words, c.export_bloom())
    finally:
        c.close()


@pytest.mark.gpu
def test_forced_chunks_of_one_workgroup_share_each(oracle):
    """8,218 stream words in eight forced position chunks: seven of 1,028 words (one full workgroup share of 1,024 + 4
    words for a second workgroup) and a last one of 1,022 (one share).  The segment estimates used to count the
    workgroups that get ANY word -- the last chunk then needed twice the region laid out for the first, and the partition
    step regrew what was a window of the chunks' common buffer (memory fault; found by the randomised sweep at
    P3_FUZZ_OFFSET=1200000, seed 17 case 2).  Same results as the oracle, whatever level the estimates end at."""
    b, o = synth.reads(seed=568349210, total_reads=2922, read_len=90, sub_rate_ppm=30000)
    k, m, H = 59, 1 << 20, 5
    c = capi.Context(0)
    try:
        c.set_option("chunks", 8)
        run_and_compare(c, oracle, (b, o), k, m, H)
        assert c.get_info("chunks") == 8
        c.run_pass(k, m, H)
        ekeys, ecounts = oracle.count_short((b, o), min_len=k)
        keys, counts = c.export_counts()
        assert np.array_equal(keys, ekeys) and np.array_equal(counts, ecounts)
    finally:
        c.close()
