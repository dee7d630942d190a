"""Experiment: V protocol ranks on ONE GPU with the reads distributed (a) by index, as a job gets them, or (b) by a
read-level minimizer (tools/shard_keys.c), so that a rank's reads cover FEWER genomic neighbourhoods more deeply.
usage: python tools/exp_minimizer_shards.py [total_reads] [V]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import bloom_sizing  # noqa: E402
from platanus3_amd import synth  # noqa: E402
from platanus3_amd.pipeline import VirtualRanks  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k, L = 31, 150
so = "/tmp/libshardkeys.so"
subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", os.path.join(ROOT, "tools", "shard_keys.c"), "-o", so])
lib = C.CDLL(so)
b, o = synth.reads(seed=1, total_reads=n, read_len=L)
t0 = time.time()
keys = np.zeros(n, np.uint64)
lib.shard_keys(b.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), C.c_uint64(n), keys.ctypes.data_as(C.c_void_p))
print("shard keys: %.1f s" % (time.time() - t0), flush=True)
m, H = bloom_sizing(n * L, k)
reads2d = b.reshape(n, L)


def run(name, owner):
    parts = []
    for v in range(V):
        idx = np.nonzero(owner == v)[0]
        bb = np.ascontiguousarray(reads2d[idx]).reshape(-1)
        oo = (np.arange(idx.size + 1, dtype=np.uint64) * L)
        parts.append((bb, oo))
    sizes = [p[1].size - 1 for p in parts]
    vr = VirtualRanks(device=0, rank=0, world=1, per_rank=V, reads=parts, min_len=21)
    ts = []
    for step in range(6):
        t0 = time.perf_counter()
        vr.run_pass(k, m, H)
        ts.append((time.perf_counter() - t0) * 1e3)
    d, p = vr.global_distinct(), vr.passing_kmers()
    local = [pp.ctx.get_info("distinct_kmers") for pp in vr.pipes]
    vr.close()
    print("%-10s reads per rank %d..%d; passes %s ms; distinct 21-mers %d, passing %d; local distinct passing k-mers per rank ~%d"
          % (name, min(sizes), max(sizes), " ".join("%.1f" % t for t in ts), d, p, int(np.mean(local))), flush=True)
    return d, p


a = run("by index", (np.arange(n) * V // n).astype(np.int64))
c = run("minimizer", (keys % np.uint64(V)).astype(np.int64))
assert a == c
