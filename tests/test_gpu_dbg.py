"""GPU tier: the de Bruijn stage (p3_build_graph / p3_write_gfa) against the golden
canonical GFA forms the REAL reference produced with -t 1, and against the oracle on other inputs."""
import os

import numpy as np
import pytest

import platanus3_amd.synth as synth
from platanus3_amd import capi
from gfa_canon import canonical
from test_oracle_dbg import check_against_golden, check_joints, dbg_cases, load_dbg_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def gpu_gfa(ctx, b, off, k, m, H, tmp_path):
    ctx.load_reads(b, off, min_len=k)
    ctx.run_pass(k, m, H)
    st = ctx.build_graph()
    path = os.path.join(str(tmp_path), "g.gfa")
    ctx.write_gfa(path)
    return open(path).read(), st


@pytest.mark.parametrize("case", dbg_cases(), ids=lambda c: c["name"])
def test_graph_matches_reference_gfa(ctx, tmp_path, case):
    b, off = load_dbg_reads(case["name"])
    gfa, st = gpu_gfa(ctx, b, off, case["k"], case["m"], case["H"], tmp_path)
    assert st["n_junctions"] == case["n_junctions"] and st["n_straights"] == case["n_straights"]
    check_against_golden(case, gfa)
    check_joints(case["name"], ctx.joint_coverage())  # G5: JointInfo::coverage


@pytest.mark.parametrize("k,n,m,H", [(31, 20000, None, None), (27, 5000, 1 << 22, 10), (45, 5000, 1 << 23, 10),
                                     (63, 4000, 1 << 23, 10)])
def test_graph_vs_oracle(ctx, oracle, tmp_path, k, n, m, H):
    b, off = synth.reads(seed=50 + k, total_reads=n, read_len=150)
    if m is None:
        m, H = oracle.estimate_bloom(int(b.size), k)
    gfa, st = gpu_gfa(ctx, b, off, k, m, H, tmp_path)
    keys, counts = oracle.count_short((b, off), min_len=k)
    words, seeds, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
    want, ost = oracle.dbg_gfa((b, off), words, m, H, seeds, k)
    assert st["n_junctions"] == ost["junctions"] and st["n_straights"] == ost["straights"]
    gn, ge = canonical(gfa)
    wn, we = canonical(want)
    assert gn == wn
    assert ge == we
    assert ctx.joint_coverage() == oracle.dbg_joints((b, off), words, m, H, seeds, k)


def test_dense_filter_is_refused_not_hung(ctx):
    """With a nearly full Bloom vector false-positive neighbours breed without limit (the reference does not
    terminate on such inputs); the device stage must stop with P3_ECAPACITY instead of growing forever."""
    b, off = synth.reads(seed=77, total_reads=300, read_len=150)
    ctx.load_reads(b, off, min_len=31)
    ctx.run_pass(31, 3000, 19)  # ~all bits set
    with pytest.raises(capi.P3Error) as e:
        ctx.build_graph()
    assert e.value.code == 3


def test_graph_from_a_seed_set_of_the_callers(ctx, oracle, tmp_path):
    """MakeDBG traverses from the seed set it is HANDED (src/DeBruijnGraph.cpp:93-105).  Reads of two unrelated genomes,
    one Bloom vector over both, seeds of the first genome only: p3_build_graph_from yields that genome's component
    alone -- the oracle's graph for the same set -- and fewer nodes than the graph from the resident seeds."""
    k = 31
    b1, o1 = synth.reads(seed=301, total_reads=3000, read_len=150)
    b2, o2 = synth.reads(seed=302, total_reads=3000, read_len=150)
    b = np.concatenate([b1, b2])
    off = np.concatenate([o1, o2[1:] + o1[-1]]).astype(np.uint64)
    m, H = oracle.estimate_bloom(int(b.size), k)
    ctx.load_reads(b, off, min_len=k)
    ctx.run_pass(k, m, H)
    full = ctx.build_graph()
    keys, counts = oracle.count_short((b, off), min_len=k)
    words, seeds_all, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
    k1, c1 = oracle.count_short((b1, o1), min_len=k)
    _, seeds1, _, _ = oracle.make_bf((b1, o1), k1, c1, k, m, H)
    assert 0 < len(seeds1) < len(seeds_all) and set(seeds1) <= set(seeds_all)
    nw = capi.nwords(k)
    sk = np.zeros((len(seeds1), nw), np.uint64)
    for i, s in enumerate(seeds1):
        sk[i] = oracle.encode_fwd(s, k)
    st = ctx.build_graph_from(sk)
    assert st["n_nodes"] < full["n_nodes"]
    path = os.path.join(str(tmp_path), "sub.gfa")
    ctx.write_gfa(path)
    want, ost = oracle.dbg_gfa((b, off), words, m, H, seeds1, k)
    assert st["n_junctions"] == ost["junctions"] and st["n_straights"] == ost["straights"]
    gn, ge = canonical(open(path).read())
    wn, we = canonical(want)
    assert gn == wn and ge == we
