"""GPU tier: the exchange INSIDE the library (p3_exchange.hip: p3_comm_* + p3_mr_run_pass, SURVEY.md sections 8b / 8e).

  * one rank over a REAL RCCL communicator (the only world size one GPU allows): librccl opened by the library, id,
    ncclCommInitRank, grouped send/recv to self, all-gather, all-reduce on the context's stream
  * 2 / 4 / 8 ranks on one GPU through the callback communicator: N contexts driven by threads, the callbacks copy
    between the ranks' DEVICE buffers -- the same p3_mr_run_pass code path as over RCCL, minus the wire
Results must equal the single-rank oracle on the union of the reads, bit for bit."""
import os
import threading

import numpy as np
import pytest
import torch

import platanus3_amd.synth as synth
from platanus3_amd import capi
from platanus3_amd.pipeline import _DevMem

pytestmark = pytest.mark.gpu


def _bytes(ptr, n):
    if n == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda:0")
    return torch.as_tensor(_DevMem(ptr, n), device="cuda:0")


class PtrWorld:
    """rendezvous of N rank threads; the callbacks get raw device pointers"""

    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slot = [None] * n

    def callbacks(self, rank):
        w, n = self, self.n

        def a2av(d_send, soff, scnt, d_recv, roff, rcnt, elem, stream):
            torch.cuda.synchronize()
            w.slot[rank] = (d_send, [soff[i] for i in range(n)], [scnt[i] for i in range(n)])
            w.barrier.wait()
            for s in range(n):
                ptr, so, sc = w.slot[s]
                nb = sc[rank] * elem
                assert sc[rank] == rcnt[s], (rank, s, sc[rank], rcnt[s])
                if nb:
                    _bytes(d_recv + roff[s] * elem, nb).copy_(_bytes(ptr + so[rank] * elem, nb))
            torch.cuda.synchronize()
            w.barrier.wait()
            return 0

        def allgather(d_send, d_recv, nbytes, stream):
            torch.cuda.synchronize()
            w.slot[rank] = d_send
            w.barrier.wait()
            for s in range(n):
                if nbytes:
                    _bytes(d_recv + s * nbytes, nbytes).copy_(_bytes(w.slot[s], nbytes))
            torch.cuda.synchronize()
            w.barrier.wait()
            return 0

        def allreduce(values, nv, op):
            w.slot[rank] = [values[i] for i in range(nv)]
            w.barrier.wait()
            cols = list(zip(*w.slot))
            res = [max(c) if op == 0 else sum(c) for c in cols]
            w.barrier.wait()
            for i in range(nv):
                values[i] = res[i]
            return 0

        return a2av, allgather, allreduce


def _check_union(oracle, b, o, k, m, H, results):
    ekeys, ecounts = oracle.count_short((b, o), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
    keys = np.concatenate([r[0] for r in results])
    counts = np.concatenate([r[1] for r in results])
    assert np.unique(keys).size == keys.size, "a key is owned by two ranks"
    order = np.argsort(keys)
    assert np.array_equal(keys[order], ekeys) and np.array_equal(counts[order], ecounts)
    for r in results:
        assert np.array_equal(r[2], ebloom)
    assert sorted(set(s for r in results for s in r[3])) == eseeds


@pytest.mark.parametrize("gate", [2, 1])
def test_run_pass_over_rccl_one_rank(oracle, gate):
    b, o = synth.reads(seed=61, total_reads=20000, read_len=150)
    k = 31
    m, H = oracle.estimate_bloom(int(b.size), k)
    comm = capi.Comm.rccl(capi.comm_unique_id(), 0, 1, 0)
    ctx = capi.Context(0)
    try:
        ctx.set_option("gate_mode", gate)
        ctx.load_reads(b, o, min_len=k)
        ctx.mr_run_pass(comm, k, m, H)
        ctx.mr_run_pass(comm, k, m, H)
        keys, counts = ctx.export_counts()
        res = [(keys, counts, ctx.export_bloom(), [capi.decode_key(x, k) for x in ctx.export_seeds()])]
        _check_union(oracle, b, o, k, m, H, res)
    finally:
        ctx.close()
        comm.close()


@pytest.mark.parametrize("world,k,gate,agg", [(2, 31, 2, 1), (4, 27, 2, 2), (4, 41, 1, 0), (8, 31, 2, 1)])
def test_run_pass_through_callbacks(oracle, world, k, gate, agg):
    n_reads = 24000
    per = n_reads // world
    b, o = synth.reads(seed=62, total_reads=n_reads, read_len=150)
    m, H = oracle.estimate_bloom(n_reads * 150, k)
    pw = PtrWorld(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            comm = capi.Comm.callbacks(r, world, *pw.callbacks(r))
            ctx = capi.Context(0)
            ctx.set_option("gate_mode", gate)
            ctx.set_option("mr_aggregate", 2 if (agg == 1 or (agg == 2 and r % 2 == 0)) else 0)
            bb = b[int(o[r * per]):int(o[(r + 1) * per])]
            oo = o[r * per:(r + 1) * per + 1] - o[r * per]
            ctx.load_reads(bb, oo, min_len=k)
            ctx.mr_run_pass(comm, k, m, H)
            keys, counts = ctx.export_counts()
            results[r] = (keys.copy(), counts.copy(), ctx.export_bloom(), [capi.decode_key(x, k) for x in ctx.export_seeds()])
            ctx.close()
            comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            pw.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    _check_union(oracle, b, o, k, m, H, results)


@pytest.mark.parametrize("procs,v,k,gate", [(1, 4, 31, 2), (1, 2, 31, 1), (2, 2, 31, 2), (2, 4, 27, 2), (4, 2, 41, 1)])
def test_run_pass_in_groups(oracle, procs, v, k, gate):
    """p3_group: v protocol ranks per "process" (a GPU whose shard exceeds one rank's 2^30 positions hosts several), `procs`
    processes joined by an outer communicator (callbacks here, RCCL in a job) that only member 0 of each group uses.
    All procs x v ranks are threads of this one process on this one GPU."""
    V = procs * v
    n_reads = 24000
    per = n_reads // V
    b, o = synth.reads(seed=63, total_reads=n_reads, read_len=150)
    m, H = oracle.estimate_bloom(n_reads * 150, k)
    pw = PtrWorld(procs)
    outers = [capi.Comm.callbacks(p, procs, *pw.callbacks(p)) if procs > 1 else None for p in range(procs)]
    groups = [capi.Group(v, outers[p]) for p in range(procs)]
    results, errors = [None] * V, []

    def worker(p, i):
        r = p * v + i
        try:
            comm = capi.Comm.member(groups[p], i)
            ctx = capi.Context(0)
            ctx.set_option("gate_mode", gate)
            bb = b[int(o[r * per]):int(o[(r + 1) * per])]
            oo = o[r * per:(r + 1) * per + 1] - o[r * per]
            ctx.load_reads(bb, oo, min_len=k)
            ctx.mr_run_pass(comm, k, m, H)
            ctx.mr_run_pass(comm, k, m, H)
            assert comm.all_reduce([r, 1], "sum") == [V * (V - 1) // 2, V]
            assert comm.all_reduce([r], "max") == [V - 1]
            keys, counts = ctx.export_counts()
            results[r] = (keys.copy(), counts.copy(), ctx.export_bloom(), [capi.decode_key(x, k) for x in ctx.export_seeds()],
                          ctx.get_info("quotient_tables"))
            ctx.close()
            comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            pw.barrier.abort()

    th = [threading.Thread(target=worker, args=(p, i)) for p in range(procs) for i in range(v)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in groups:
        g.close()
    for c in outers:
        if c is not None:
            c.close()
    assert not errors, errors
    _check_union(oracle, b, o, k, m, H, results)
    if gate == 2:  # (second pass of a rank: its owned cells counted on quotient slots, the sending ranks as source tags)
        assert all(r[4] > 0 for r in results), [r[4] for r in results]


@pytest.mark.parametrize("case_name,procs,v", [("synth2000_k31", 1, 4), ("synth3000_k51", 2, 2), ("long40_k63", 1, 2),
                                                ("wide:synth1500_k101", 1, 4), ("wide:dense3000_k80", 2, 2),
                                                ("long:dense300_k501", 1, 2), ("long:long40_k3001", 2, 2), ("long:mid400_k129", 1, 4)])
def test_sharded_graph_equals_reference_gfa(tmp_path, case_name, procs, v):
    """p3_mr_run_pass + p3_mr_build_graph over procs x v ranks (threads on this one GPU): rank 0's GFA against the REAL
    reference's for the union of the reads (canonical form), JointInfo::coverage included -- keys of one to four words and of
    the reference's long-read widths (k = 501: 16 words, k = 3001: 94)"""
    import gzip
    import hashlib
    import json
    import os
    from gfa_canon import canonical
    from test_oracle_dbg import canon_lines
    golden = os.path.join(os.path.dirname(__file__), "golden")
    wide = case_name.startswith("wide:") or case_name.startswith("long:")
    name = case_name.split(":")[-1]
    prefix = "dbgl_" if case_name.startswith("long:") else "dbgw_" if wide else "dbg_"
    case = [c for c in json.load(open(os.path.join(golden, prefix + "cases.json"))) if c["name"] == name][0]
    z = np.load(os.path.join(golden, prefix + name + "_reads.npz"))
    b, o = z["bases"], z["offsets"]
    k, m, H = case["k"], case["m"], case["H"]
    V = procs * v
    n_reads = o.size - 1
    cuts = [n_reads * i // V for i in range(V + 1)]
    pw = PtrWorld(procs)
    outers = [capi.Comm.callbacks(p, procs, *pw.callbacks(p)) if procs > 1 else None for p in range(procs)]
    groups = [capi.Group(v, outers[p]) for p in range(procs)]
    out, errors = {}, []

    def worker(p, i):
        r = p * v + i
        try:
            comm = capi.Comm.member(groups[p], i)
            ctx = capi.Context(0)
            a, e = cuts[r], cuts[r + 1]
            ctx.load_reads(b[int(o[a]):int(o[e])], o[a:e + 1] - o[a], min_len=k)
            ctx.mr_run_pass(comm, k, m, H)
            ctx.mr_build_graph(comm)
            if r == 0:
                path = os.path.join(str(tmp_path), "g.gfa")
                ctx.write_gfa(path)
                out["gfa"] = open(path).read()
                out["stats"] = ctx.graph_stats()
                out["joints"] = ctx.joint_coverage()
            ctx.close()
            comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            pw.barrier.abort()

    th = [threading.Thread(target=worker, args=(p, i)) for p in range(procs) for i in range(v)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in groups:
        g.close()
    for c in outers:
        if c is not None:
            c.close()
    assert not errors, errors
    assert out["stats"]["n_junctions"] == case["n_junctions"] and out["stats"]["n_straights"] == case["n_straights"]
    ls = canon_lines(*canonical(out["gfa"]))
    with gzip.open(os.path.join(golden, prefix + name + ".canon.gz"), "rt") as f:
        assert ls == f.read().split("\n")
    jl = ["%s\t%d" % (a, c) for a, c in out["joints"]]
    jg = case["joints"] if wide else json.load(open(os.path.join(golden, "dbg_joints.json")))[name]
    assert len(jl) == jg["n"] and hashlib.sha256("\n".join(jl).encode()).hexdigest() == jg["sha256"]


@pytest.mark.parametrize("procs,v,k", [(1, 4, 31), (2, 2, 45), (1, 8, 31), (2, 4, 101)])
def test_resharded_reads_give_the_same_results(oracle, procs, v, k):
    """p3_mr_reshard_reads: every read moves to rank (its smallest canonical-21-mer hash) mod N before the passes.  The read
    multiset is conserved and count table, vector and seed set equal the oracle's for the union -- ragged input: reads
    shorter than 21, of unknown characters only, empty ones."""
    V = procs * v
    rng = np.random.RandomState(7)
    b0, o0 = synth.reads(seed=65, total_reads=12000, read_len=200, sub_rate_ppm=4000)
    reads = []
    for i in range(12000):
        r = bytearray(b0[int(o0[i]):int(o0[i + 1])].tobytes())
        L = [200, 150, 101, 30, 20, 0, 200, 180][i % 8]
        r = r[:L]
        if i % 13 == 0 and L:
            r[rng.randint(0, L)] = ord("N")
        if i % 97 == 0:
            r = bytearray(b"N" * L)
        reads.append(bytes(r))
    b = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    o = np.zeros(len(reads) + 1, np.uint64)
    o[1:] = np.cumsum([len(r) for r in reads])
    n_reads = len(reads)
    m, H = oracle.estimate_bloom(int(b.size), k)
    cuts = [n_reads * i // V for i in range(V + 1)]
    pw = PtrWorld(procs)
    outers = [capi.Comm.callbacks(p, procs, *pw.callbacks(p)) if procs > 1 else None for p in range(procs)]
    groups = [capi.Group(v, outers[p]) for p in range(procs)]
    results, sizes, errors = [None] * V, [None] * V, []

    def worker(p, i):
        r = p * v + i
        try:
            comm = capi.Comm.member(groups[p], i)
            ctx = capi.Context(0)
            a, e = cuts[r], cuts[r + 1]
            ctx.load_reads(b[int(o[a]):int(o[e])], o[a:e + 1] - o[a], min_len=k)
            ctx.mr_reshard_reads(comm)
            ctx.mr_run_pass(comm, k, m, H)
            ctx.mr_run_pass(comm, k, m, H)
            keys, counts = ctx.export_counts()
            results[r] = (keys.copy(), counts.copy(), ctx.export_bloom(), [capi.decode_key(x, k) for x in ctx.export_seeds()])
            sizes[r] = ctx.count_stats()["n_windows"]
            ctx.close()
            comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            pw.barrier.abort()

    th = [threading.Thread(target=worker, args=(p, i)) for p in range(procs) for i in range(v)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in groups:
        g.close()
    for c in outers:
        if c is not None:
            c.close()
    assert not errors, errors
    _check_union(oracle, b, o, k, m, H, results)
    # every valid window is still somewhere, exactly once
    assert sum(sizes) == sum(max(0, len(r) - 20) for r in reads if len(r) >= k)
    assert min(sizes) > 0.5 * sum(sizes) / V  # and the deal is even enough


def test_reshard_spreads_the_reads_of_a_heavy_key(oracle):
    """a third of the reads are copies of ONE read (a repeat: one shard key for all of them): they are dealt round robin
    instead of overloading one rank; results as ever equal the oracle's"""
    V, k = 4, 31
    b0, o0 = synth.reads(seed=66, total_reads=9000, read_len=150)
    reads = [b0[int(o0[i]):int(o0[i + 1])].tobytes() for i in range(9000)]
    reads = reads[:6000] + [reads[7]] * 3000
    order = np.random.RandomState(3).permutation(len(reads))
    reads = [reads[i] for i in order]
    b = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    o = (np.arange(len(reads) + 1, dtype=np.uint64) * 150)
    m, H = oracle.estimate_bloom(int(b.size), k)
    g = capi.Group(V, None)
    per = len(reads) // V
    results, sizes, errors = [None] * V, [None] * V, []

    def worker(i):
        try:
            comm = capi.Comm.member(g, i)
            ctx = capi.Context(0)
            ctx.load_reads(b[int(o[i * per]):int(o[(i + 1) * per])], o[i * per:(i + 1) * per + 1] - o[i * per], min_len=k)
            ctx.mr_reshard_reads(comm)
            ctx.mr_run_pass(comm, k, m, H)
            keys, counts = ctx.export_counts()
            results[i] = (keys.copy(), counts.copy(), ctx.export_bloom(), [capi.decode_key(x, k) for x in ctx.export_seeds()])
            sizes[i] = ctx.count_stats()["n_windows"]
            ctx.close()
            comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(V)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    g.close()
    assert not errors, errors
    _check_union(oracle, b, o, k, m, H, results)
    assert max(sizes) < 1.15 * sum(sizes) / V, sizes   # (a third of the reads on one rank would be 1.75 x the mean)


def test_group_member_failure_does_not_hang():
    """a member that fails (no reads loaded) aborts the group: the other member returns an error instead of waiting"""
    g = capi.Group(2, None)
    b, o = synth.reads(seed=64, total_reads=4000, read_len=150)
    errors = [None, None]

    def worker(i):
        comm = capi.Comm.member(g, i)
        ctx = capi.Context(0)
        try:
            if i == 0:
                ctx.load_reads(b, o, min_len=31)
            ctx.mr_run_pass(comm, 31, 1 << 24, 3)
        except capi.P3Error as e:
            errors[i] = e
        finally:
            ctx.close()
            comm.close()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in th), "a group member is still waiting"
    g.close()
    assert errors[0] is not None and errors[1] is not None


def test_cpp_host_drives_the_sharded_pass(tmp_path):
    """platanus3_amd/host/mr_demo.cpp: a C++ program -- one thread per GPU, the library's own RCCL communicators,
    p3_mr_run_pass -- with no Python in the loop (one GPU here; its numbers must equal the single-rank pass)."""
    import os
    import subprocess
    from platanus3_amd import _build
    from bench import bloom_sizing
    host = os.path.join(os.path.dirname(_build.CSRC), "host")
    exe = os.path.join(str(tmp_path), "mr_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + _build.INCLUDE, os.path.join(host, "mr_demo.cpp"), "-o", exe,
                           "-L" + _build.CSRC, "-lplatanus3_amd", "-Wl,-rpath," + _build.CSRC, "-pthread"])
    n, k = 30000, 31
    out = subprocess.check_output([exe, "1", str(n), str(k)], timeout=120).decode()
    last = out.strip().split("\n")[-1].split()
    total, passing = int(last[1]), int(last[4])
    b, o = synth.reads(seed=3, total_reads=n, read_len=150)
    m, H = bloom_sizing(int(b.size), k)
    ctx = capi.Context(0)
    try:
        ctx.load_reads(b, o, min_len=k)
        ctx.run_pass(k, m, H)
        assert ctx.count_stats()["n_distinct"] == total
        assert ctx.bloom_stats()["n_passing"] == passing
    finally:
        ctx.close()


@pytest.mark.parametrize("world,k,reshard,graph", [(2, 31, 0, 1), (4, 31, 1, 0), (4, 45, 0, 0)])
def test_run_pass_in_separate_processes(oracle, tmp_path, world, k, reshard, graph):
    """p3_mr_run_pass with every rank in its OWN process (python -m torch.distributed.run, one GPU shared by all):
    the collectives of a callbacks communicator go through gloo with host staging (tests/mp_mr_worker.py).  Checks
    what the thread-driven tests cannot: the launch contract, processes that each own a HIP context on the device,
    ids and results crossing process boundaries, and the teardown order.  Union of the ranks' tables, every rank's
    merged vector and the union of the seeds equal the oracle's single-process results; with `graph`, rank 0's sharded
    de Bruijn graph equals the oracle's."""
    import subprocess
    import sys
    import socket
    from gfa_canon import canonical
    n_reads, seed = 16000, 64
    m, H = oracle.estimate_bloom(n_reads * 150, k)
    with socket.socket() as s:  # a free port for the rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "mp_mr_worker.py"), str(tmp_path), str(n_reads), str(k), str(m), str(H),
           str(seed), str(reshard), str(graph)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-4000:]
    b, o = synth.reads(seed=seed, total_reads=n_reads, read_len=150)
    res = []
    for rk in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rk))
        res.append((z["keys"], z["counts"], z["bloom"], [capi.decode_key(x, k) for x in z["seeds"]]))
    _check_union(oracle, b, o, k, m, H, res)
    if graph:
        ekeys, ecounts = oracle.count_short((b, o), min_len=k)
        words, eseeds, _, _ = oracle.make_bf((b, o), ekeys, ecounts, k, m, H)
        want, _ = oracle.dbg_gfa((b, o), words, m, H, eseeds, k)
        got = open(os.path.join(str(tmp_path), "graph.gfa")).read()
        assert canonical(got) == canonical(want)
