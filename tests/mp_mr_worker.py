"""One protocol rank in its own PROCESS (GPU tier helper, started by tests/test_gpu_exchange.py through
`python -m torch.distributed.run`): p3_mr_run_pass over a callbacks communicator whose three collectives go through
torch.distributed's gloo backend with host staging -- device -> host copy, gloo collective between the processes,
host -> device copy.  Slow on purpose: what it exercises is what the thread-driven tests cannot -- separate processes
sharing one GPU, the launch contract (RANK / WORLD_SIZE / MASTER_* from the environment), the exchange of results
between processes and the teardown order (contexts and communicators closed before the process group).

usage: mp_mr_worker.py <out_dir> <n_reads> <k> <m> <H> <seed> <reshard 0|1> <graph 0|1>
Rank r writes <out_dir>/rank<r>.npz (its share of the count table, the merged Bloom vector, its seeds)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import platanus3_amd.synth as synth
from platanus3_amd import capi

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


def d2h(ptr, nbytes):
    out = np.empty(max(1, nbytes), np.uint8)
    if nbytes:
        rc = hip().hipMemcpy(out.ctypes.data, C.c_void_p(ptr), nbytes, 2)
        assert rc == 0, "hipMemcpy D2H failed: %d" % rc
    return out[:nbytes]


def h2d(ptr, arr):
    arr = np.ascontiguousarray(arr)
    if arr.nbytes:
        rc = hip().hipMemcpy(C.c_void_p(ptr), arr.ctypes.data, arr.nbytes, 1)
        assert rc == 0, "hipMemcpy H2D failed: %d" % rc


def gloo_callbacks(rank, world):
    def a2av(d_send, soff, scnt, d_recv, roff, rcnt, elem, stream):
        hip().hipDeviceSynchronize()
        ins = [torch.from_numpy(d2h(d_send + soff[s] * elem, scnt[s] * elem).copy()) for s in range(world)]
        outs = [torch.empty(rcnt[s] * elem, dtype=torch.uint8) for s in range(world)]
        # (gloo has no variable all-to-all for lists of different sizes on every version: pairwise exchange)
        reqs = []
        for s in range(world):
            if s == rank:
                outs[s].copy_(ins[s])
                continue
            if ins[s].numel():
                reqs.append(dist.isend(ins[s], dst=s))
            if outs[s].numel():
                reqs.append(dist.irecv(outs[s], src=s))
        for q in reqs:
            q.wait()
        for s in range(world):
            h2d(d_recv + roff[s] * elem, outs[s].numpy())
        hip().hipDeviceSynchronize()
        return 0

    def allgather(d_send, d_recv, nbytes, stream):
        hip().hipDeviceSynchronize()
        mine = torch.from_numpy(d2h(d_send, nbytes).copy())
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, mine)
        for s in range(world):
            h2d(d_recv + s * nbytes, parts[s].numpy())
        hip().hipDeviceSynchronize()
        return 0

    def allreduce(values, nv, op):
        # (u64 values: reduce as two 32-bit halves would break max; every value here is far below 2^63)
        t = torch.tensor([int(values[i]) for i in range(nv)], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 0 else dist.ReduceOp.SUM)
        for i in range(nv):
            values[i] = int(t[i])
        return 0

    return a2av, allgather, allreduce


def main():
    out_dir, n_reads, k, m, H, seed, reshard, graph = sys.argv[1], *[int(x) for x in sys.argv[2:9]]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = comm = None
    try:
        per = n_reads // world
        # every rank generates ITS reads of the global set (the generator is counter-based: no exchange needed)
        b, o = synth.reads(seed=seed, total_reads=n_reads, n_reads=per, first_read=rank * per, read_len=150)
        comm = capi.Comm.callbacks(rank, world, *gloo_callbacks(rank, world))
        ctx = capi.Context(0)
        ctx.load_reads(b, o, min_len=k)
        if reshard:
            ctx.mr_reshard_reads(comm)
        ctx.mr_run_pass(comm, k, m, H)
        ctx.mr_run_pass(comm, k, m, H)  # a second pass over the same reads (remembered splits, cached verdicts)
        keys, counts = ctx.export_counts()
        seeds = ctx.export_seeds()
        extra = {}
        if graph:
            ctx.mr_build_graph(comm)
            if rank == 0:
                ctx.write_gfa(os.path.join(out_dir, "graph.gfa"))
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), keys=keys, counts=counts, bloom=ctx.export_bloom(), seeds=seeds, **extra)
    finally:
        # order matters: device contexts and the communicator first, the process group last
        if ctx is not None:
            ctx.close()
        if comm is not None:
            comm.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
