"""CPU tier: the oracle's restatement for keys of five words and more (129 <= k <= 3001, the reference's k = 501, 1001,
3001) against golden vectors from the REAL reference (tests/golden/long.json, made by tests/golden/make_golden_long.py),
and against the live reference where its build exists."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.golden.make_golden_long import KS, read_set

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "long.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def test_known_answers(oracle):
    assert {g["k"] for g in GOLD["kat"]} == set(KS)
    for g in GOLD["kat"]:
        k, s = g["k"], g["seq"]
        f, r = oracle.encode_fwd(s), oracle.encode_rc(s)
        assert [int(x) for x in f] == g["fwd"] and [int(x) for x in r] == g["rc"], (k, s[:20])
        assert [int(x) for x in oracle.complement(f, k)] == g["comp"]
        cn = oracle.canonical(f, r, k)
        assert [int(x) for x in cn] == g["canon"]
        assert oracle.std_hash(cn, k) == g["std_hash"]
        assert oracle.double_hash(cn, k) == (g["h1"], g["h2"])
        assert [int(x) for x in oracle.probe_positions(g["h1"], g["h2"], g["m"], g["H"])] == g["pos"]
        assert oracle.decode(f, k) == "".join(c if c in "ACGT" else "A" for c in s)


@pytest.mark.parametrize("name", ["long4k", "long1k5", "ragged_long"])
def test_make_bf_against_reference_goldens(oracle, name):
    b, o = read_set(name)
    cases = [g for g in GOLD["reads"] if g["set"] == name]
    assert len(cases) >= len(KS)
    for g in cases:
        k = g["k"]
        keys, counts = oracle.count_short((b, o), min_len=k)  # (the loader drops reads shorter than k, src/Load.cpp)
        assert keys.size == g["n_distinct_21"]
        words, seeds, n_ins, ovf = oracle.make_bf((b, o), keys, counts, k, g["m"], g["H"])
        assert sha(words) == g["bloom_sha256"], (name, k, g["m"])
        assert int(np.unpackbits(words.view(np.uint8)).sum()) == g["bits_set"]
        assert len(seeds) == g["n_seeds"]
        assert hashlib.sha256("".join(seeds).encode()).hexdigest() == g["seeds_sha256"]
        if seeds:
            assert hashlib.sha256(seeds[0].encode()).hexdigest() == g["first_seed_sha256"]
            canon = np.concatenate([oracle.canonical(oracle.encode_fwd(s), oracle.encode_rc(s), k) for s in seeds[:20]])
            assert oracle.bloom_query(words, g["m"], g["H"], canon, k).all()


def test_live_reference_random_reads(oracle, reflib):
    rng = np.random.RandomState(15)
    for trial in range(5):
        k = int(KS[trial % len(KS)])
        n = int(rng.randint(10, 40))
        genome = rng.choice(list(b"ACGT"), 9000).astype(np.uint8)
        reads = []
        for _ in range(n):
            L = int(rng.randint(0, 2 * k + 200))
            a = int(rng.randint(0, max(1, 9000 - L)))
            r = genome[a:a + L].copy()
            if L and rng.rand() < 0.2:
                r[rng.randint(0, len(r))] = ord("N")
            reads.append(r.tobytes())
        bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
        offs = np.zeros(n + 1, np.uint64)
        offs[1:] = np.cumsum([len(r) for r in reads])
        m, H = int(rng.randint(5000, 200000)), int(rng.randint(1, 12))
        ref = reflib.run((bases, offs), k, m, H)
        keys, counts = oracle.count_short((bases, offs), min_len=k)
        assert np.array_equal(keys, ref["keys"]) and np.array_equal(counts, ref["counts"])
        words, seeds, _, _ = oracle.make_bf((bases, offs), keys, counts, k, m, H)
        assert np.array_equal(words, ref["bloom"]) and seeds == ref["seeds"], (trial, k)


# ---- the oracle's de Bruijn restatement beyond four key words --------------------------------------------------------------
def _dbgl_cases():
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dbgl_cases.json")))
    # (the restatement reverse-complements base by base over whole words: O(k^2 / 32) per neighbour -- the cases of the
    #  widest keys take a minute and more each on the CPU and are the device tier's alone)
    return [c for c in cases if c["k"] <= 600 or c["name"] == "long120_k1001"]


@pytest.mark.parametrize("case", _dbgl_cases(), ids=lambda c: c["name"])
def test_oracle_long_graph_matches_reference(oracle, case):
    """p3_oracle_dbg.cpp (a template over the key type: UW<16>, UW<32> ... beyond four words, narrower keys in the next wider
    type) against the REAL reference's GFA and JointInfo::coverage for k = 129 .. 1001 (tests/golden/make_golden_dbg_long.py)"""
    import gzip
    from gfa_canon import canonical
    from test_oracle_dbg import canon_lines
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "dbgl_%s_reads.npz" % case["name"]))
    b, off = z["bases"], z["offsets"]
    k, m, H = case["k"], case["m"], case["H"]
    keys, counts = oracle.count_short((b, off), min_len=k)
    words, seeds, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
    gfa, st = oracle.dbg_gfa((b, off), words, m, H, seeds, k)
    assert st["junctions"] == case["n_junctions"] and st["straights"] == case["n_straights"]
    ls = canon_lines(*canonical(gfa))
    with gzip.open(os.path.join(os.path.dirname(__file__), "golden", "dbgl_%s.canon.gz" % case["name"]), "rt") as f:
        assert ls == f.read().split("\n")
    joints = oracle.dbg_joints((b, off), words, m, H, seeds, k)
    jl = ["%s\t%d" % (a, c) for a, c in joints]
    assert len(joints) == case["joints"]["n"] and hashlib.sha256("\n".join(jl).encode()).hexdigest() == case["joints"]["sha256"]
