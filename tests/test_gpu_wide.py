"""GPU tier: keys of three and four words (65 <= k <= 128; the reference's k = 101) through the C ABI -- count, gate,
Bloom vector, seeds -- against golden vectors from the REAL reference (tests/golden/wide.json) and against the oracle on
random reads; the de Bruijn stage against the reference's GFA (tests/golden/dbgw_*)."""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

import platanus3_amd.synth as synth
from platanus3_amd import capi
from tests.golden.make_golden_wide import KS, read_set

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wide.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["short150", "long1k", "ragged"])
def test_pass_against_reference_goldens(name):
    b, o = read_set(name)
    cases = [g for g in GOLD["reads"] if g["set"] == name]
    assert {g["k"] for g in cases} == set(KS)
    ctx = capi.Context(0)
    try:
        last_k = None
        for g in cases:
            k = g["k"]
            if k != last_k:
                ctx.load_reads(b, o, min_len=k)
                last_k = k
            ctx.run_pass(k, g["m"], g["H"])
            assert ctx.count_stats()["n_distinct"] == g["n_distinct_21"]
            words = ctx.export_bloom()
            assert int(np.unpackbits(words.view(np.uint8)).sum()) == g["bits_set"], (name, k, g["m"])
            assert sha(words) == g["bloom_sha256"], (name, k, g["m"])
            seeds = [capi.decode_key(x, k) for x in ctx.export_seeds()]
            assert len(seeds) == g["n_seeds"] and seeds[:3] == g["first_seeds"]
            assert hashlib.sha256("".join(seeds).encode()).hexdigest() == g["seeds_sha256"]
    finally:
        ctx.close()


def _random_reads(rng, n, max_len, genome_len=4000, unknown=0.15):
    genome = rng.choice(list(b"ACGT"), genome_len).astype(np.uint8)
    reads = []
    for _ in range(n):
        L = int(rng.randint(0, max_len))
        a = int(rng.randint(0, genome_len - max_len))
        r = genome[a:a + L].copy()
        if L and rng.rand() < unknown:
            r[rng.randint(0, L)] = ord("N")
        if L > 3 and rng.rand() < 0.3:
            r[rng.randint(0, L)] = rng.choice(list(b"ACGT"))  # a substitution: windows that fail the gate
        reads.append(r.tobytes())
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    return bases, offs


def test_random_reads_against_oracle(oracle):
    rng = np.random.RandomState(11)
    ctx = capi.Context(0)
    try:
        for trial in range(14):
            k = int(rng.randint(65, 129))
            b, o = _random_reads(rng, int(rng.randint(50, 400)), int(rng.choice([140, 260, 700])))
            m, H = int(rng.randint(20000, 3000000)), int(rng.randint(1, 20))
            ekeys, ecounts = oracle.count_short((b, o), min_len=k)
            ewords, eseeds, n_ins, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
            ctx.load_reads(b, o, min_len=k)
            for gate in (2, 1):
                ctx.set_option("gate_mode", gate)
                ctx.run_pass(k, m, H)
                assert np.array_equal(ctx.export_bloom(), ewords), (trial, k, gate)
                assert [capi.decode_key(x, k) for x in ctx.export_seeds()] == eseeds, (trial, k, gate)
                assert ctx.bloom_stats()["n_passing"] == n_ins
            ctx.set_option("gate_mode", 0)
            # B2: canonical keys of the seeds are present; random keys match the oracle's answer bit for bit
            nw = capi.nwords(k)
            q = rng.randint(0, 1 << 62, size=(64, nw)).astype(np.uint64)
            q[:, nw - 1] &= np.uint64((1 << (2 * k - 64 * (nw - 1))) - 1) if 2 * k - 64 * (nw - 1) < 64 else np.uint64(2**64 - 1)
            if eseeds:
                canon = np.stack([oracle.canonical(oracle.encode_fwd(s), oracle.encode_rc(s), k) for s in eseeds[:32]])
                q = np.concatenate([q, canon])
            got = ctx.bloom_query(q.reshape(-1))
            assert np.array_equal(got, oracle.bloom_query(ewords, m, H, q.reshape(-1), k))
            if eseeds:
                assert got[64:].all()
    finally:
        ctx.close()


def test_every_wide_k_on_one_read_set(oracle):
    """k = 65 .. 128, all of them: word-boundary cases of the rolling key (2k mod 64 = 2, 62, 0 ...)"""
    b, o = synth.reads(seed=74, total_reads=400, read_len=260, sub_rate_ppm=4000)
    ctx = capi.Context(0)
    try:
        for k in range(65, 129):
            m, H = 400009 + k, 5
            ekeys, ecounts = oracle.count_short((b, o), min_len=k)
            ewords, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
            ctx.load_reads(b, o, min_len=k)
            ctx.run_pass(k, m, H)
            assert np.array_equal(ctx.export_bloom(), ewords), k
            assert [capi.decode_key(x, k) for x in ctx.export_seeds()] == eseeds, k
    finally:
        ctx.close()


def test_k_beyond_the_widest_dispatch_is_refused():
    b, o = synth.reads(seed=75, total_reads=300, read_len=200)
    ctx = capi.Context(0)
    try:
        ctx.load_reads(b, o, min_len=101)
        with pytest.raises(capi.P3Error) as e:
            ctx.run_pass(3002, 1 << 22, 7)   # (the reference's widest case is k = 3001, src/Assemble.cpp:48)
        assert e.value.code == 2  # P3_EUNSUPPORTED_K
    finally:
        ctx.close()


def test_sharded_pass_with_wide_keys(oracle):
    """k = 101 over two protocol ranks on one GPU (p3_group): partial vectors OR-merged, seeds per rank"""
    k, V = 101, 2
    b, o = synth.reads(seed=76, total_reads=4000, read_len=200, sub_rate_ppm=4000)
    m, H = oracle.estimate_bloom(int(b.size), k)
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    ewords, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    grp = capi.Group(V, None)
    per = 4000 // V
    res, errors = [None] * V, []

    def worker(i):
        try:
            comm = capi.Comm.member(grp, i)
            ctx = capi.Context(0)
            ctx.load_reads(b[int(o[i * per]):int(o[(i + 1) * per])], o[i * per:(i + 1) * per + 1] - o[i * per], min_len=k)
            ctx.mr_run_pass(comm, k, m, H)
            res[i] = (ctx.export_bloom(), [capi.decode_key(x, k) for x in ctx.export_seeds()])
            ctx.close()
            comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(V)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.close()
    assert not errors, errors
    for w, _ in res:
        assert np.array_equal(w, ewords)
    assert sorted(set(s for _, sd in res for s in sd)) == eseeds


# ---- the de Bruijn stage for 3- and 4-word keys --------------------------------------------------------------------------
def _dbgw_cases():
    p = os.path.join(os.path.dirname(__file__), "golden", "dbgw_cases.json")
    return json.load(open(p)) if os.path.exists(p) else []


@pytest.mark.parametrize("case", _dbgw_cases(), ids=lambda c: c["name"])
def test_wide_graph_matches_reference_gfa(tmp_path, case):
    """MakeDBG + CountNodeCoverage + PrintGraph at k = 65 .. 128 against the REAL reference's GFA (-t 1) in ID-free
    canonical form, and JointInfo::coverage against what the reference accumulated (tests/golden/make_golden_dbg_wide.py)"""
    import gzip
    from gfa_canon import canonical
    from test_oracle_dbg import canon_lines
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "dbgw_%s_reads.npz" % case["name"]))
    k = case["k"]
    ctx = capi.Context(0)
    try:
        ctx.load_reads(z["bases"], z["offsets"], min_len=k)
        ctx.run_pass(k, case["m"], case["H"])
        st = ctx.build_graph()
        assert st["n_junctions"] == case["n_junctions"] and st["n_straights"] == case["n_straights"]
        path = os.path.join(str(tmp_path), "g.gfa")
        ctx.write_gfa(path)
        nodes, edges = canonical(open(path).read())
        ls = canon_lines(nodes, edges)
        with gzip.open(os.path.join(os.path.dirname(__file__), "golden", "dbgw_%s.canon.gz" % case["name"]), "rt") as f:
            assert ls == f.read().split("\n")
        assert hashlib.sha256("\n".join(ls).encode()).hexdigest() == case["sha256"]
        joints = ctx.joint_coverage()
        jl = ["%s\t%d" % (a, b) for a, b in joints]
        g = case["joints"]
        assert len(joints) == g["n"] and sum(b for _, b in joints) == g["coverage_sum"]
        assert hashlib.sha256("\n".join(jl).encode()).hexdigest() == g["sha256"]
    finally:
        ctx.close()


@pytest.mark.parametrize("k,n,L,m,H", [(65, 3000, 250, 1 << 21, 4), (101, 4000, 250, None, None), (128, 3000, 300, 1 << 20, 3),
                                       (77, 60, 3000, 1 << 21, 5)])
def test_wide_graph_vs_oracle(oracle, tmp_path, k, n, L, m, H):
    """inputs beyond the goldens: the device graph against the oracle's restatement (itself pinned on the ten goldens)"""
    from gfa_canon import canonical
    b, off = synth.reads(seed=90 + k, total_reads=n, read_len=L, sub_rate_ppm=6000)
    if m is None:
        m, H = 1 << 22, 10
    ctx = capi.Context(0)
    try:
        ctx.load_reads(b, off, min_len=k)
        ctx.run_pass(k, m, H)
        st = ctx.build_graph()
        path = os.path.join(str(tmp_path), "g.gfa")
        ctx.write_gfa(path)
        keys, counts = oracle.count_short((b, off), min_len=k)
        words, seeds, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
        want, ost = oracle.dbg_gfa((b, off), words, m, H, seeds, k)
        assert st["n_junctions"] == ost["junctions"] and st["n_straights"] == ost["straights"]
        assert canonical(open(path).read()) == canonical(want)
        assert ctx.joint_coverage() == oracle.dbg_joints((b, off), words, m, H, seeds, k)
    finally:
        ctx.close()


@pytest.mark.parametrize("chunks", [2, 4])
def test_wide_keys_over_position_chunks(oracle, tmp_path, chunks):
    """the form of inputs beyond 2^30 window positions (the count stage works chunk by chunk and assembles S over the whole
    stream) with four-word keys, forced on a small input: vector, seeds and graph against the oracle"""
    from gfa_canon import canonical
    k, m, H = 101, 1 << 22, 10
    b, off = synth.reads(seed=95, total_reads=24000, read_len=250, sub_rate_ppm=5000)
    ctx = capi.Context(0)
    try:
        ctx.set_option("chunks", chunks)
        ctx.load_reads(b, off, min_len=k)
        ctx.run_pass(k, m, H)
        assert ctx.get_info("chunks") == chunks
        keys, counts = oracle.count_short((b, off), min_len=k)
        words, seeds, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
        assert np.array_equal(ctx.export_bloom(), words)
        assert [capi.decode_key(x, k) for x in ctx.export_seeds()] == seeds
        st = ctx.build_graph()
        path = os.path.join(str(tmp_path), "g.gfa")
        ctx.write_gfa(path)
        want, ost = oracle.dbg_gfa((b, off), words, m, H, seeds, k)
        assert st["n_junctions"] == ost["junctions"] and canonical(open(path).read()) == canonical(want)
    finally:
        ctx.close()
