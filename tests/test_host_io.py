"""Host side of the call surface: the loader (ReadFile ctor + LoadFile, reference
src/Load.cpp:23-103), Options::Parse / EstimateBloomfilter (src/Options.cpp:23-60) and the command line
(main.cpp:11-34) as re-stated in platanus3_amd/host/platanus3_shim.hpp + platanus3_main.cpp.

CPU tier: the loader and the sizing arithmetic against fixtures the REAL reference produced
(tests/golden/io/expected.json, generator tests/golden/make_golden_io.py) and, where the reference build is
present, against the reference itself on randomly generated files.  GPU tier: the command line end to end --
FASTA in, ./de_bruijn_graph.gfa out -- against the reference's GFA in canonical form."""
import base64
import json
import os
import random
import subprocess

import numpy as np
import pytest

from platanus3_amd import _build
from util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "platanus3_amd", "host")
IO = os.path.join(GOLDEN, "io")


_compiled = {}


def _compile(tmp_path_factory, src, name):
    if (src, name) in _compiled:   # (platanus3_main.cpp instantiates the shim for 44 values of k: ~8 s per build)
        return _compiled[(src, name)]
    _build.build_lib()
    exe = os.path.join(str(tmp_path_factory.mktemp("hostbin")), name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + _build.INCLUDE, "-I" + HOST, os.path.join(HOST, src), "-o", exe,
                           "-L" + _build.CSRC, "-lplatanus3_amd", "-Wl,-rpath," + _build.CSRC, "-pthread"])
    _compiled[(src, name)] = exe
    return exe


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    return _compile(tmp_path_factory, "io_probe.cpp", "io_probe")


def _load(probe, path, k):
    out = subprocess.run([probe, "load", path, str(k)], check=True, capture_output=True, timeout=60).stdout
    head, _, rest = out.partition(b"\n")
    all_bases, n, suffix_ok = [int(x) for x in head.split()]
    lines = rest.split(b"\n")[:-1] if rest else []
    assert len(lines) == 2 * n
    return dict(all_bases=all_bases, suffix_ok=suffix_ok, entries=[(lines[i], lines[i + 1]) for i in range(0, len(lines), 2)])


EXPECTED = json.load(open(os.path.join(IO, "expected.json")))


@pytest.mark.parametrize("case", EXPECTED["loads"], ids=lambda c: "%s-k%d" % (c["file"], c["k"]))
def test_loader_matches_reference_fixture(probe, case):
    got = _load(probe, os.path.join(IO, case["file"]), case["k"])
    want = [(base64.b64decode(n), base64.b64decode(s)) for n, s in case["entries"]]
    assert got["entries"] == want
    assert got["all_bases"] == case["all_bases"] and got["suffix_ok"] == case["suffix_ok"]


def test_estimate_matches_reference_fixture(probe):
    for e in EXPECTED["estimates"]:
        args = [probe, "estimate", str(e["all_bases"]), str(e["k"])] + ([str(e["m_in"])] if e["m_in"] else [])
        m, H = [int(x) for x in subprocess.run(args, check=True, capture_output=True).stdout.split()]
        assert (m, H) == (e["m"], e["H"]), e


def test_option_parsing(probe):
    def parse(*a):
        return subprocess.run([probe, "parse"] + list(a), check=True, capture_output=True).stdout.decode().strip()
    assert parse() == "1 [] 0 10 8 25 21"                                       # defaults, src/Options.cpp:10-16
    assert parse("-i", "r.fasta", "-k", "31", "-t", "3") == "1 [r.fasta] 0 10 3 31 21"
    assert parse("-m", "1073741824", "-i", "x.fastq") == "1 [x.fastq] 1073741824 10 8 25 21"
    assert parse("-i", "r.fasta", "-z").startswith("0 [r.fasta]")               # unknown letter -> false


def test_short_file_name_throws_like_substr(probe):
    r = subprocess.run([probe, "load", "abc", "21"], capture_output=True)
    assert r.returncode != 0   # the reference's substr(size-5,5) throws std::out_of_range as well (src/Load.cpp:26)


def _random_text(rng, fastq):
    alphabet = "ACGT" * 6 + "Nacgt\r "
    out = []
    for _ in range(rng.randint(0, 12)):
        roll = rng.random()
        if roll < 0.35:
            out.append((">" if not fastq or rng.random() < 0.2 else "@") + "n%d %s" % (rng.randint(0, 5), "x" * rng.randint(0, 3)))
        elif roll < 0.45:
            out.append("")
        elif roll < 0.55:
            out.append("+" + "q" * rng.randint(0, 3))
        else:
            out.append("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 60))))
    text = "\n".join(out)
    if rng.random() < 0.7:
        text += "\n"
    first = "@" if fastq else ">"
    return (first + "first\n" if rng.random() < 0.9 else "") + text


def test_loader_fuzz_against_reference(probe, tmp_path):
    from oracle.p3_oracle import RefLib
    if not RefLib.available():
        pytest.skip("reference build not present")
    ref = RefLib()
    rng = random.Random(1234)
    for i in range(120):
        fastq = i % 2 == 1
        path = os.path.join(str(tmp_path), "f%d.%s" % (i, "fastq" if fastq else "fasta"))
        with open(path, "wb") as f:
            f.write(_random_text(rng, fastq).encode())
        for k in (21, 33):
            want = ref.load_file(path, k)
            got = _load(probe, path, k)
            assert got["entries"] == want["entries"], path
            assert got["all_bases"] == want["all_bases"], path


@pytest.mark.gpu
@pytest.mark.parametrize("materialise", [0, 1])
@pytest.mark.parametrize("name", ["synth400_k21", "synth2000_k31", "synth2000_k31_m24", "synth3000_k51", "long40_k63"])
def test_command_line_end_to_end(tmp_path_factory, tmp_path, name, materialise):
    """platanus3 -i reads.fasta -k K [-m M] -t 1: the GFA written to the working directory equals the reference's
    (canonical form); the Bloom sizing is the command line's own (EstimateBloomfilter or -m with H = 10).
    materialise=0: the stages hand their results on in HBM (AssembleResident, the default); 1: through the
    reference-typed containers of Assemble<B> (PLATANUS3_MATERIALISE=1).  Same log lines either way."""
    from test_oracle_dbg import check_against_golden, load_dbg_reads
    if materialise and name not in ("synth2000_k31", "long40_k63"):
        pytest.skip("the reference-typed path is run on one 1-word and one 2-word case (suite time)")
    case = [c for c in json.load(open(os.path.join(GOLDEN, "dbg_cases.json"))) if c["name"] == name][0]
    exe = _compile(tmp_path_factory, "platanus3_main.cpp", "platanus3")
    b, off = load_dbg_reads(name)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(off.size - 1):
            f.write(">r%d\n%s\n" % (i, b[int(off[i]):int(off[i + 1])].tobytes().decode()))
    args = [exe, "-i", fa, "-k", str(case["k"]), "-t", "1"] + (["-m", str(case["m"])] if case["H"] == 10 else [])
    subprocess.check_call(args, cwd=str(tmp_path), timeout=180, env=dict(os.environ, PLATANUS3_MATERIALISE=str(materialise)))
    check_against_golden(case, open(os.path.join(str(tmp_path), "de_bruijn_graph.gfa")).read())
    log = open(os.path.join(str(tmp_path), "platanus3.log")).read()
    assert "filter_size : %d" % case["m"] in log and "num_hashes : %d" % case["H"] in log and log.rstrip().endswith("finish")
    steps = [l for l in log.split("\n") if not l.startswith(("all_bases", "item_number", "get filter_size", "readfile_name",
                                                             "filter_size", "num_hashes", "kmer_length", "error_rate"))]
    assert [l.split("=")[0] for l in steps if l] == [
        "read file loaded", "Assemble", "counted short kmer", "bloom filter loaded", "get seed kmer", "seed kmer num",
        "start graph extention", "de bruijn graph loaded", "count node coverage", "finish"]
    if "n_seeds" in case:
        assert "seed kmer num= %d" % case["n_seeds"] in log


@pytest.mark.gpu
@pytest.mark.parametrize("materialise", [0, 1])
@pytest.mark.parametrize("name", ["synth1500_k101", "long40_k127", "dbgl:long120_k501", "dbgl:long40_k3001", "dbgl:mid400_k160"])
def test_command_line_wide_keys(tmp_path_factory, tmp_path, name, materialise):
    """platanus3 -k 101, -k 501, -k 3001 -- cases of the reference's own switch (src/Assemble.cpp:40-49) -- and -k 127: keys
    of four, sixteen and ninety-four words through the command line, GFA against the real reference's
    (tests/golden/dbgw_*, dbgl_*).  The reference-typed containers (PLATANUS3_MATERIALISE=1: Assemble<std::bitset<2k>>)
    exist for the k the reference dispatches, as there."""
    import gzip
    import hashlib
    from gfa_canon import canonical
    from test_oracle_dbg import canon_lines
    prefix = "dbgl" if name.startswith("dbgl:") else "dbgw"
    name = name.split(":")[-1]
    if materialise and name in ("long40_k127", "mid400_k160"):
        pytest.skip("Assemble_k dispatches k = 101, 501, 1001, 2001, 3001 of the wide keys")
    case = [c for c in json.load(open(os.path.join(GOLDEN, prefix + "_cases.json"))) if c["name"] == name][0]
    exe = _compile(tmp_path_factory, "platanus3_main.cpp", "platanus3")
    z = np.load(os.path.join(GOLDEN, "%s_%s_reads.npz" % (prefix, name)))
    b, off = z["bases"], z["offsets"]
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(off.size - 1):
            f.write(">r%d\n%s\n" % (i, b[int(off[i]):int(off[i + 1])].tobytes().decode()))
    assert case["H"] == 10
    subprocess.check_call([exe, "-i", fa, "-k", str(case["k"]), "-t", "1", "-m", str(case["m"])], cwd=str(tmp_path), timeout=180,
                          env=dict(os.environ, PLATANUS3_MATERIALISE=str(materialise)))
    nodes, edges = canonical(open(os.path.join(str(tmp_path), "de_bruijn_graph.gfa")).read())
    ls = canon_lines(nodes, edges)
    with gzip.open(os.path.join(GOLDEN, "%s_%s.canon.gz" % (prefix, name)), "rt") as f:
        assert ls == f.read().split("\n")
    assert hashlib.sha256("\n".join(ls).encode()).hexdigest() == case["sha256"]
    assert open(os.path.join(str(tmp_path), "platanus3.log")).read().rstrip().endswith("finish")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["synth2000_k31", "wide:synth1500_k101"])
def test_command_line_sharded_path(tmp_path_factory, tmp_path, name):
    """PLATANUS3_GPUS=1: the run goes through AssembleSharded -- one thread and one REAL RCCL rank per GPU (the box has
    one), p3_mr_run_pass + p3_mr_build_graph -- and writes the reference's GFA; more GPUs than present: exit status 3."""
    import gzip
    from gfa_canon import canonical
    from test_oracle_dbg import canon_lines
    wide = name.startswith("wide:")
    nm = name.split(":")[-1]
    prefix = "dbgw_" if wide else "dbg_"
    case = [c for c in json.load(open(os.path.join(GOLDEN, prefix + "cases.json"))) if c["name"] == nm][0]
    exe = _compile(tmp_path_factory, "platanus3_main.cpp", "platanus3")
    z = np.load(os.path.join(GOLDEN, prefix + nm + "_reads.npz"))
    b, off = z["bases"], z["offsets"]
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for i in range(off.size - 1):
            f.write(">r%d\n%s\n" % (i, b[int(off[i]):int(off[i + 1])].tobytes().decode()))
    args = [exe, "-i", fa, "-k", str(case["k"]), "-t", "1"] + (["-m", str(case["m"])] if case["H"] == 10 else [])
    subprocess.check_call(args, cwd=str(tmp_path), timeout=180, env=dict(os.environ, PLATANUS3_GPUS="1"))
    ls = canon_lines(*canonical(open(os.path.join(str(tmp_path), "de_bruijn_graph.gfa")).read()))
    with gzip.open(os.path.join(GOLDEN, prefix + nm + ".canon.gz"), "rt") as f:
        assert ls == f.read().split("\n")
    log = open(os.path.join(str(tmp_path), "platanus3.log")).read()
    assert "seed kmer num= " in log and log.rstrip().endswith("finish")
    r = subprocess.run(args, cwd=str(tmp_path), timeout=180, env=dict(os.environ, PLATANUS3_GPUS="64"), capture_output=True)
    assert r.returncode == 3 and b"GPUs asked for" in r.stderr


@pytest.mark.gpu
def test_command_line_fastq_and_multiline_fasta(tmp_path_factory, tmp_path):
    """The same reads as a FASTQ file (four lines per record) and as a FASTA file with the sequences wrapped at 60
    columns give the reference's GFA too (row T1: both loaders feed the same ReadSet)."""
    from test_oracle_dbg import check_against_golden, load_dbg_reads
    name = "synth2000_k31"
    case = [c for c in json.load(open(os.path.join(GOLDEN, "dbg_cases.json"))) if c["name"] == name][0]
    exe = _compile(tmp_path_factory, "platanus3_main.cpp", "platanus3")
    b, off = load_dbg_reads(name)
    seqs = [b[int(off[i]):int(off[i + 1])].tobytes().decode() for i in range(off.size - 1)]
    for sub, fname, text in (
            ("fq", "reads.fastq", "".join("@r%d\n%s\n+\n%s\n" % (i, q, "I" * len(q)) for i, q in enumerate(seqs))),
            ("wrapped", "reads.fasta", "".join(">r%d\n%s\n" % (i, "\n".join(q[j:j + 60] for j in range(0, len(q), 60)))
                                               for i, q in enumerate(seqs)))):
        wd = os.path.join(str(tmp_path), sub)
        os.makedirs(wd)
        with open(os.path.join(wd, fname), "w") as f:
            f.write(text)
        args = [exe, "-i", os.path.join(wd, fname), "-k", str(case["k"]), "-t", "1"] + (["-m", str(case["m"])] if case["H"] == 10 else [])
        subprocess.check_call(args, cwd=wd, timeout=180)
        check_against_golden(case, open(os.path.join(wd, "de_bruijn_graph.gfa")).read())


def test_command_line_without_input_prints_usage(tmp_path_factory, tmp_path):
    exe = _compile(tmp_path_factory, "platanus3_main.cpp", "platanus3")
    r = subprocess.run([exe], capture_output=True, cwd=str(tmp_path), timeout=60)
    assert r.returncode == 0 and r.stdout.decode().startswith("Usage: platanus3 -i")
