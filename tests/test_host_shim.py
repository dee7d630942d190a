"""The C++ host shim (platanus3_amd/host/platanus3_shim.hpp) that re-creates the reference's call surface
(ReadFile::CountShortKmer, MakeBF<B>, BF<B>) on top of the C ABI: compile check on CPU, end-to-end vs the
golden vectors of the real reference on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from platanus3_amd import _build
from util import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "platanus3_amd", "host")


_exe = []


def _compile(tmp_path, syntax_only=False):
    if _exe and not syntax_only:   # (one build per session: ~3.5 s each)
        return _exe[0]
    _build.build_lib()
    import tempfile
    exe = os.path.join(str(tmp_path) if syntax_only else tempfile.mkdtemp(prefix="p3shim_"), "assemble_demo")
    cmd = ["g++", "-std=c++17", "-O2", "-I" + _build.INCLUDE, "-I" + HOST, os.path.join(HOST, "assemble_demo.cpp")]
    if syntax_only:
        cmd.insert(1, "-fsyntax-only")
    else:
        cmd += ["-o", exe, "-L" + _build.CSRC, "-lplatanus3_amd", "-Wl,-rpath," + _build.CSRC, "-pthread"]
    subprocess.check_call(cmd)
    if not syntax_only:
        _exe.append(exe)
    return exe


def test_shim_compiles(tmp_path):
    _compile(tmp_path, syntax_only=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["synth300_k31", "synth200_k41", "ragged_k25", "long24_k63"])
def test_shim_matches_reference_golden(tmp_path, name):
    exe = _compile(tmp_path)
    c = load_case(name)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i, r in enumerate(c["reads"]):
            f.write(">r%d\n%s\n" % (i, r.decode()))
    out = os.path.join(str(tmp_path), "out")
    subprocess.check_call([exe, fa, str(c["k"]), str(c["m"]), str(c["H"]), out], cwd=str(tmp_path), timeout=120)
    kc = np.loadtxt(out + ".counts", dtype=np.uint64, ndmin=2)
    assert np.array_equal(kc[:, 0], c["keys"]) and np.array_equal(kc[:, 1], c["counts"])
    assert np.array_equal(np.fromfile(out + ".bloom", dtype=np.uint64), c["bloom"])
    seeds = [s for s in open(out + ".seeds").read().split("\n") if s]
    assert seeds == c["seeds"]


@pytest.mark.gpu
def test_shim_k101_matches_reference_golden(tmp_path):
    """MakeBF<std::bitset<202>> -- the reference's own k = 101 (src/Assemble.cpp:40) -- through the shim, against the real
    reference's vector and seed set (tests/golden/wide.json)"""
    import hashlib
    import json
    from tests.golden.make_golden_wide import read_set
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "wide.json")))
    g = [c for c in gold["reads"] if c["set"] == "short150" and c["k"] == 101 and c["m"] == 5000011][0]
    b, o = read_set("short150")
    exe = _compile(tmp_path)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(o.size - 1):
            f.write(">r%d\n%s\n" % (i, bytes(b[int(o[i]):int(o[i + 1])]).decode()))
    out = os.path.join(str(tmp_path), "out")
    subprocess.check_call([exe, fa, "101", str(g["m"]), str(g["H"]), out], cwd=str(tmp_path), timeout=120)
    words = np.fromfile(out + ".bloom", dtype=np.uint64)
    assert hashlib.sha256(words.tobytes()).hexdigest() == g["bloom_sha256"]
    seeds = [s for s in open(out + ".seeds").read().split("\n") if s]
    assert len(seeds) == g["n_seeds"] and hashlib.sha256("".join(seeds).encode()).hexdigest() == g["seeds_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [501, 1001])
def test_shim_long_keys_match_reference_golden(tmp_path, k):
    """MakeBF<std::bitset<1002>> / <2002> -- the reference's long-read cases (src/Assemble.cpp:42-45) -- through the shim,
    against the real reference's vector and seed set (tests/golden/long.json)"""
    import hashlib
    import json
    from tests.golden.make_golden_long import read_set
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "long.json")))
    g = [c for c in gold["reads"] if c["set"] == "long4k" and c["k"] == k][0]
    b, o = read_set("long4k")
    exe = _compile(tmp_path)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(o.size - 1):
            f.write(">r%d\n%s\n" % (i, bytes(b[int(o[i]):int(o[i + 1])]).decode()))
    out = os.path.join(str(tmp_path), "out")
    subprocess.check_call([exe, fa, str(k), str(g["m"]), str(g["H"]), out], cwd=str(tmp_path), timeout=120)
    words = np.fromfile(out + ".bloom", dtype=np.uint64)
    assert hashlib.sha256(words.tobytes()).hexdigest() == g["bloom_sha256"]
    seeds = [s for s in open(out + ".seeds").read().split("\n") if s]
    assert len(seeds) == g["n_seeds"] and hashlib.sha256("".join(seeds).encode()).hexdigest() == g["seeds_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["synth2000_k31", "synth1000_k41", "long40_k63"])
def test_shim_full_assemble_gfa(tmp_path, name):
    """The whole Assemble<B> sequence through the shim (CountShortKmer, MakeBF, DeBruijnGraph::MakeDBG,
    CountNodeCoverage, PrintGraph): the GFA equals the reference's in canonical form."""
    import json
    from test_oracle_dbg import check_against_golden, load_dbg_reads
    from util import GOLDEN
    case = [c for c in json.load(open(os.path.join(GOLDEN, "dbg_cases.json"))) if c["name"] == name][0]
    exe = _compile(tmp_path)
    b, off = load_dbg_reads(name)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(off.size - 1):
            f.write(">r%d\n%s\n" % (i, b[int(off[i]):int(off[i + 1])].tobytes().decode()))
    out = os.path.join(str(tmp_path), "out")
    subprocess.check_call([exe, fa, str(case["k"]), str(case["m"]), str(case["H"]), out, "graph"], cwd=str(tmp_path),
                          timeout=120)
    check_against_golden(case, open(os.path.join(str(tmp_path), "de_bruijn_graph.gfa")).read())
    nj, njo, ns, covsum, jointsum = [int(x) for x in open(out + ".graph").read().split()]
    assert nj == case["n_junctions"] and ns == case["n_straights"] and njo <= 2 * ns
    # JointInfo::coverage materialised into the shim's joints map == what the real reference accumulated
    gj = json.load(open(os.path.join(GOLDEN, "dbg_joints.json")))[name]
    assert njo == gj["n"] and jointsum == gj["coverage_sum"]


@pytest.mark.gpu
@pytest.mark.parametrize("prefix,name", [("dbgw", "synth1500_k101"), ("dbgl", "long120_k501"), ("dbgl", "long120_k1001")])
def test_shim_full_assemble_gfa_wide_keys(tmp_path, prefix, name):
    """Assemble<std::bitset<202>> / <1002> / <2002> through the shim, graph stage included (junction keys of 4, 16 and 32
    words materialised into the reference's maps): the GFA equals the reference's in canonical form."""
    import gzip
    import hashlib
    import json
    from gfa_canon import canonical
    from test_oracle_dbg import canon_lines
    from util import GOLDEN
    cases = json.load(open(os.path.join(GOLDEN, prefix + "_cases.json")))
    case = [c for c in cases if c["name"] == name][0]
    z = np.load(os.path.join(GOLDEN, "%s_%s_reads.npz" % (prefix, name)))
    b, off = z["bases"], z["offsets"]
    exe = _compile(tmp_path)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(off.size - 1):
            f.write(">r%d\n%s\n" % (i, b[int(off[i]):int(off[i + 1])].tobytes().decode()))
    out = os.path.join(str(tmp_path), "out")
    subprocess.check_call([exe, fa, str(case["k"]), str(case["m"]), str(case["H"]), out, "graph"], cwd=str(tmp_path),
                          timeout=300)
    nodes, edges = canonical(open(os.path.join(str(tmp_path), "de_bruijn_graph.gfa")).read())
    ls = canon_lines(nodes, edges)
    with gzip.open(os.path.join(GOLDEN, "%s_%s.canon.gz" % (prefix, name)), "rt") as f:
        assert ls == f.read().split("\n")
    assert hashlib.sha256("\n".join(ls).encode()).hexdigest() == case["sha256"]


@pytest.mark.gpu
def test_shim_makedbg_honours_the_seed_set_it_is_handed(tmp_path, oracle):
    """DeBruijnGraph::MakeDBG(seedkmer, ...) traverses from the set it is GIVEN (src/DeBruijnGraph.cpp:93-105).  A caller
    that replaces the set MakeBF returned by the seeds of one of two unrelated genomes gets that genome's component --
    the oracle's graph for that set -- not the graph of the resident seeds."""
    import platanus3_amd.synth as synth
    from gfa_canon import canonical
    k = 31
    b1, o1 = synth.reads(seed=301, total_reads=2000, read_len=150)
    b2, o2 = synth.reads(seed=302, total_reads=2000, read_len=150)
    b = np.concatenate([b1, b2])
    off = np.concatenate([o1, o2[1:] + o1[-1]]).astype(np.uint64)
    m, H = oracle.estimate_bloom(int(b.size), k)
    keys, counts = oracle.count_short((b, off), min_len=k)
    words, seeds_all, _, _ = oracle.make_bf((b, off), keys, counts, k, m, H)
    k1, c1 = oracle.count_short((b1, o1), min_len=k)
    _, seeds1, _, _ = oracle.make_bf((b1, o1), k1, c1, k, m, H)
    exe = _compile(tmp_path)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(off.size - 1):
            f.write(">r%d\n%s\n" % (i, b[int(off[i]):int(off[i + 1])].tobytes().decode()))
    sf = os.path.join(str(tmp_path), "seeds1.txt")
    with open(sf, "w") as f:
        f.write("\n".join(seeds1) + "\n")
    out = os.path.join(str(tmp_path), "out")
    env = dict(os.environ, P3_DEMO_SEEDS=sf)
    subprocess.check_call([exe, fa, str(k), str(m), str(H), out, "graph"], cwd=str(tmp_path), timeout=120, env=env)
    got = open(os.path.join(str(tmp_path), "de_bruijn_graph.gfa")).read()
    want, _ = oracle.dbg_gfa((b, off), words, m, H, seeds1, k)
    full, _ = oracle.dbg_gfa((b, off), words, m, H, seeds_all, k)
    assert canonical(got) == canonical(want)
    assert canonical(got) != canonical(full)
