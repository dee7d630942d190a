"""CPU tier: the N>1 path.  world_size-2 and -4 gloo jobs run platanus3_amd.pipeline.Pipeline.run_pass -- the
code that routes canonical 21-mers to their bucket owners, agrees on the split, all-gathers the solid regions
and OR-merges the partial Bloom vectors -- with the CPU stand-in backend (tests/cpu_backend.py) in place of the
HIP library.  The merged results must equal the single-rank oracle on the union of the reads, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, n_reads, tmpdir, gate, agg):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import platanus3_amd.synth as synth
        from platanus3_amd.pipeline import Pipeline
        from oracle.p3_oracle import Oracle
        from cpu_backend import CpuBackend
        o = Oracle()
        per = n_reads // world
        b, offs = synth.reads(seed=5, total_reads=n_reads, first_read=rank * per, n_reads=per, read_len=150)
        m, H = o.estimate_bloom(n_reads * 150, k)
        # agg: 1 = every rank pre-aggregates what it sends, 0 = raw instance records, 2 = mixed (even ranks only):
        # the owners decode both forms record by record
        be = CpuBackend(o, gate=gate, aggregate=(agg == 1) or (agg == 2 and rank % 2 == 0))
        pipe = Pipeline(rank=rank, world=world, backend=be)
        pipe.load_reads(b, offs, min_len=k)
        pipe.run_pass(k, m, H)
        np.savez(os.path.join(tmpdir, "rank%d.npz" % rank), keys=be.shard_keys, counts=be.shard_counts,
                 bloom=be.bloom[:be.n_bloom_words], seeds=np.array(be.seeds, dtype="S") if be.seeds else np.zeros(0, "S1"))
        # the pieces used to move across ranks must be owned-by-exactly-one-rank: a cheap global invariant
        t = torch.tensor([be.shard_keys.size], dtype=torch.int64)
        dist.all_reduce(t)
        np.save(os.path.join(tmpdir, "total%d.npy" % rank), t.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,gate,agg", [(2, 31, 2, 1), (4, 25, 2, 1), (2, 31, 2, 0), (4, 41, 2, 2), (2, 47, 1, 0),
                                              (4, 31, 1, 0)])
def test_sharded_pass_equals_single_rank(tmp_path, oracle, world, k, gate, agg):
    import platanus3_amd.synth as synth
    n_reads = 1200
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, n_reads, str(tmp_path), gate, agg), nprocs=world, join=True)
    b, offs = synth.reads(seed=5, total_reads=n_reads, read_len=150)
    m, H = oracle.estimate_bloom(n_reads * 150, k)
    ekeys, ecounts = oracle.count_short((b, offs), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, offs), ekeys, ecounts, k, m, H)
    shards = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]
    keys = np.concatenate([s["keys"] for s in shards])
    counts = np.concatenate([s["counts"] for s in shards])
    assert np.unique(keys).size == keys.size, "a key is owned by two ranks"
    order = np.argsort(keys)
    assert np.array_equal(keys[order], ekeys) and np.array_equal(counts[order], ecounts)
    for s in shards:  # after the OR-merge + all-gather every rank holds the full vector
        assert np.array_equal(s["bloom"], ebloom)
    seeds = sorted(set(x.decode() for s in shards for x in s["seeds"]))
    assert seeds == eseeds
    assert int(np.load(os.path.join(tmp_path, "total0.npy"))[0]) == ekeys.size



def _vworker(rank, world, port, k, n_reads, tmpdir, per_rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import platanus3_amd.synth as synth
        from platanus3_amd.pipeline import VirtualRanks
        from oracle.p3_oracle import Oracle
        from cpu_backend import CpuBackend
        o = Oracle()
        V = world * per_rank
        per = n_reads // V
        reads = [synth.reads(seed=6, total_reads=n_reads, first_read=(rank * per_rank + i) * per, n_reads=per, read_len=150)
                 for i in range(per_rank)]
        m, H = o.estimate_bloom(n_reads * 150, k)
        bes = [CpuBackend(o, gate=2, aggregate=(i % 2 == 0)) for i in range(per_rank)]
        vr = VirtualRanks(rank=rank, world=world, per_rank=per_rank, backends=bes, reads=reads, min_len=k)
        vr.run_pass(k, m, H)
        for i, be in enumerate(bes):
            np.savez(os.path.join(tmpdir, "rank%d.npz" % (rank * per_rank + i)), keys=be.shard_keys, counts=be.shard_counts,
                     bloom=be.bloom[:be.n_bloom_words], seeds=np.array(be.seeds, dtype="S") if be.seeds else np.zeros(0, "S1"))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("world,per_rank,k", [(2, 2, 31), (1, 4, 25), (2, 4, 41)])
def test_virtual_ranks_equal_single_rank(tmp_path, oracle, world, per_rank, k):
    """several protocol ranks per process (pipeline.VirtualRanks: how bench.py hosts a config-#3 shard that exceeds one
    rank's 2^30 window positions on 2 or 4 GPUs): V = world x per_rank virtual ranks, collectives inside a process by
    copies and between processes over gloo.  Union of the shards == single-rank oracle."""
    import platanus3_amd.synth as synth
    n_reads = 1600
    V = world * per_rank
    port = _free_port()
    if world > 1:
        mp.spawn(_vworker, args=(world, port, k, n_reads, str(tmp_path), per_rank), nprocs=world, join=True)
    else:
        _vworker(0, 1, port, k, n_reads, str(tmp_path), per_rank)
    b, offs = synth.reads(seed=6, total_reads=n_reads, read_len=150)
    m, H = oracle.estimate_bloom(n_reads * 150, k)
    ekeys, ecounts = oracle.count_short((b, offs), min_len=k)
    ebloom, eseeds, _, _ = oracle.make_bf((b, offs), ekeys, ecounts, k, m, H)
    shards = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(V)]
    keys = np.concatenate([s["keys"] for s in shards])
    counts = np.concatenate([s["counts"] for s in shards])
    assert np.unique(keys).size == keys.size, "a key is owned by two ranks"
    order = np.argsort(keys)
    assert np.array_equal(keys[order], ekeys) and np.array_equal(counts[order], ecounts)
    for s in shards:
        assert np.array_equal(s["bloom"], ebloom)
    assert sorted(set(x.decode() for s in shards for x in s["seeds"])) == eseeds
