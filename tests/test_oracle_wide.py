"""CPU tier: the oracle's restatement for keys of three and four words (65 <= k <= 128) against golden vectors from the
REAL reference (tests/golden/wide.json, made by tests/golden/make_golden_wide.py), and against the live reference where
its build exists."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.golden.make_golden_wide import KS, read_set

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wide.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def test_known_answers(oracle):
    assert {g["k"] for g in GOLD["kat"]} == set(KS)
    for g in GOLD["kat"]:
        k, s = g["k"], g["seq"]
        f, r = oracle.encode_fwd(s), oracle.encode_rc(s)
        assert [int(x) for x in f] == g["fwd"] and [int(x) for x in r] == g["rc"], (k, s)
        assert [int(x) for x in oracle.complement(f, k)] == g["comp"]
        cn = oracle.canonical(f, r, k)
        assert [int(x) for x in cn] == g["canon"]
        assert oracle.std_hash(cn, k) == g["std_hash"]
        assert oracle.double_hash(cn, k) == (g["h1"], g["h2"])
        assert [int(x) for x in oracle.probe_positions(g["h1"], g["h2"], g["m"], g["H"])] == g["pos"]
        assert oracle.decode(f, k) == "".join(c if c in "ACGT" else "A" for c in s)


@pytest.mark.parametrize("name", ["short150", "long1k", "ragged"])
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
        assert len(seeds) == g["n_seeds"] and seeds[:3] == g["first_seeds"]
        assert hashlib.sha256("".join(seeds).encode()).hexdigest() == g["seeds_sha256"]
        # B2 on the vector just built: every seed's canonical key is in it
        if seeds:
            canon = np.concatenate([oracle.canonical(oracle.encode_fwd(s), oracle.encode_rc(s), k) for s in seeds[:50]])
            assert oracle.bloom_query(words, g["m"], g["H"], canon, k).all()


def test_live_reference_random_reads(oracle, reflib):
    rng = np.random.RandomState(5)
    for trial in range(6):
        k = int(rng.choice(KS))
        n = int(rng.randint(20, 120))
        genome = rng.choice(list(b"ACGT"), 3000).astype(np.uint8)
        reads = []
        for _ in range(n):
            L = int(rng.randint(0, 400))
            a = int(rng.randint(0, 3000 - 400))
            r = genome[a:a + L].copy()
            if L and rng.rand() < 0.2:
                r[rng.randint(0, L)] = ord("N")
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


# ---- the oracle's de Bruijn restatement for 3- and 4-word keys -------------------------------------------------------------
def _dbgw_cases():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dbgw_cases.json")))


@pytest.mark.parametrize("case", _dbgw_cases(), ids=lambda c: c["name"])
def test_oracle_wide_graph_matches_reference(oracle, case):
    """p3_oracle_dbg.cpp (a template over the key type: u128 up to k = 64, four words beyond) against the REAL reference's GFA
    and JointInfo::coverage for k = 65 .. 128 (tests/golden/make_golden_dbg_wide.py)"""
    import gzip
    from gfa_canon import canonical
    from test_oracle_dbg import canon_lines
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "dbgw_%s_reads.npz" % case["name"]))
    b, off = z["bases"], z["offsets"]
    k, m, H = case["k"], case["m"], case["H"]
    keys, counts = oracle.count_short((b, off), min_len=k)
    words, seeds, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
    gfa, st = oracle.dbg_gfa((b, off), words, m, H, seeds, k)
    assert st["junctions"] == case["n_junctions"] and st["straights"] == case["n_straights"]
    ls = canon_lines(*canonical(gfa))
    with gzip.open(os.path.join(os.path.dirname(__file__), "golden", "dbgw_%s.canon.gz" % case["name"]), "rt") as f:
        assert ls == f.read().split("\n")
    joints = oracle.dbg_joints((b, off), words, m, H, seeds, k)
    jl = ["%s\t%d" % (a, c) for a, c in joints]
    assert len(joints) == case["joints"]["n"] and hashlib.sha256("\n".join(jl).encode()).hexdigest() == case["joints"]["sha256"]
