"""CPU tier: the oracle's de Bruijn stage (oracle/p3_oracle_dbg.cpp) against the
golden canonical GFA forms the REAL reference produced with -t 1 (tests/golden/make_golden_dbg.py)."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from gfa_canon import canonical
from util import GOLDEN


def dbg_cases():
    with open(os.path.join(GOLDEN, "dbg_cases.json")) as f:
        return json.load(f)


def canon_lines(nodes, edges):
    out = []
    for (kind, seq, kc), n in sorted(nodes.items()):
        out.append("N\t%s\t%s\t%s\t%d" % (kind, seq, kc, n))
    for (a, b, ov), n in sorted(edges.items()):
        out.append("E\t%s\t%s\t%d\t%s\t%s\t%d\t%s\t%d" % (a[0], a[1], a[2], b[0], b[1], b[2], ov, n))
    return out


def load_dbg_reads(name):
    z = np.load(os.path.join(GOLDEN, "dbg_" + name + "_reads.npz"))
    return z["bases"], z["offsets"]


def check_against_golden(case, gfa_text):
    nodes, edges = canonical(gfa_text)
    ls = canon_lines(nodes, edges)
    assert sum(nodes.values()) == case["n_nodes"] and sum(edges.values()) == case["n_edges"]
    if case["full"]:
        with gzip.open(os.path.join(GOLDEN, "dbg_" + case["name"] + ".canon.gz"), "rt") as f:
            want = f.read().split("\n")
        assert ls == want
    assert hashlib.sha256("\n".join(ls).encode()).hexdigest() == case["sha256"]


@pytest.mark.parametrize("case", dbg_cases(), ids=lambda c: c["name"])
def test_oracle_dbg_matches_reference_gfa(oracle, case):
    b, off = load_dbg_reads(case["name"])
    k, m, H = case["k"], case["m"], case["H"]
    keys, counts = oracle.count_short((b, off), min_len=k)
    words, seeds, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
    gfa, st = oracle.dbg_gfa((b, off), words, m, H, seeds, k)
    assert st["junctions"] == case["n_junctions"] and st["straights"] == case["n_straights"]
    check_against_golden(case, gfa)


def joint_golden(name):
    with open(os.path.join(GOLDEN, "dbg_joints.json")) as f:
        return json.load(f)[name]


def check_joints(name, joints):
    """joints: [(joint k-mer in its smaller orientation, JointInfo::coverage)] sorted"""
    g = joint_golden(name)
    lines = ["%s\t%d" % (a, b) for a, b in joints]
    assert len(joints) == g["n"] and sum(b for _, b in joints) == g["coverage_sum"]
    if "lines" in g:
        assert lines == g["lines"]
    assert hashlib.sha256("\n".join(lines).encode()).hexdigest() == g["sha256"]


@pytest.mark.parametrize("case", dbg_cases(), ids=lambda c: c["name"])
def test_oracle_joint_coverage_matches_reference(oracle, case):
    """G5: JointInfo::coverage (src/DeBruijnGraph.cpp:441-449), a field the reference fills but never prints, against
    what the real reference accumulated (tests/golden/make_golden_joints.py)"""
    b, off = load_dbg_reads(case["name"])
    k, m, H = case["k"], case["m"], case["H"]
    keys, counts = oracle.count_short((b, off), min_len=k)
    words, seeds, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
    check_joints(case["name"], oracle.dbg_joints((b, off), words, m, H, seeds, k))
