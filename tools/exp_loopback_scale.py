"""Per-rank kernel times of the multi-rank protocol at weak scale, emulated on ONE GPU (N contexts, threads).
Ranks share the GPU, so absolute times are inflated; the point is to spot work that grows with N."""
import sys, threading, json
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import platanus3_amd.synth as synth
from platanus3_amd.pipeline import Pipeline
from loopback_comm import LoopbackWorld
from bench import bloom_sizing

world = int(sys.argv[1]); per = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
agg = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # mr_aggregate: 0 raw, 1 automatic, 2 always
k = 31
total = per * world
m, H = bloom_sizing(total * 150, k)
lw = LoopbackWorld(world, serialize=True)
out = [None] * world
# SHARD=minimizer: the reads dealt out by a read-level minimizer (what p3_mr_reshard_reads does; here on the host, because
# this harness drives the Python form of the protocol) instead of by index
shards = None
if os.environ.get("SHARD", "") == "minimizer":
    import ctypes as C, subprocess
    import numpy as np
    so = "/tmp/libshardkeys.so"
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", os.path.join(R, "tools", "shard_keys.c"), "-o", so])
    lib = C.CDLL(so)
    ball, oall = synth.reads(seed=2, total_reads=total, read_len=150)
    keys = np.zeros(total, np.uint64)
    lib.shard_keys(ball.ctypes.data_as(C.c_void_p), oall.ctypes.data_as(C.c_void_p), C.c_uint64(total), keys.ctypes.data_as(C.c_void_p))
    owner = (keys % np.uint64(world)).astype(np.int64)
    r2d = ball.reshape(total, 150)
    shards = []
    for r_ in range(world):
        idx = np.nonzero(owner == r_)[0]
        shards.append((np.ascontiguousarray(r2d[idx]).reshape(-1), np.arange(idx.size + 1, dtype=np.uint64) * 150))
def worker(r):
    lw.begin(r)
    try:
        _worker(r)
    finally:
        lw.end(r)


def _worker(r):
    b, o = shards[r] if shards is not None else synth.reads(seed=2, total_reads=total, first_read=r * per, n_reads=per, read_len=150)
    pipe = Pipeline(device=0, rank=r, world=world, comm=lw.comm(r))
    pipe.ctx.set_option('mr_aggregate', agg)
    if os.environ.get("AGG_LAYOUT"):
        pipe.ctx.set_option('agg_layout', int(os.environ["AGG_LAYOUT"]))
    pipe.load_reads(b, o, min_len=k)
    pipe.run_pass(k, m, H)
    if r == 0:
        pipe.ctx.reset_kernel_stats(); pipe.ctx.set_profiling(True)
    pipe.run_pass(k, m, H)
    if r == 0:
        pipe.ctx.set_profiling(False)
    import time as _t
    h0, t_open = lw.held[r], lw._t0[r]
    pipe.run_pass(k, m, H)
    held = lw.held[r] - h0 + (_t.perf_counter() - lw._t0[r])   # this rank's own time between collectives, one pass
    if r == 0:
        out[0] = {n: round(s["total_ms"], 3) for n, s in sorted(pipe.ctx.kernel_stats().items())}
        out[0]["_sbits"] = pipe.ctx.get_info("sbits"); out[0]["_ksbits"] = pipe.ctx.get_info("ksbits")
        out[0]["_distinct_k"] = pipe.ctx.get_info("distinct_kmers")
        out[0]["_send_records"] = pipe.ctx.get_info("mr_send_records")
        out[0]["_sum_kernels_ms"] = round(sum(v for n, v in out[0].items() if n.startswith("k_")), 3)
        out[0]["_rank_busy_ms"] = round(held * 1e3, 2)
th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
print(json.dumps(out[0]))
