"""Time of the device-side grouping of reads by shard key (p3_group_reads, the local half of p3_mr_reshard_reads) next to
one pass over the same reads (GPU box only).  usage: python tools/exp_reshard.py <n_reads> [groups]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = 31
b, o = synth.reads(seed=3, total_reads=n, read_len=150)
m, H = bloom_sizing(int(b.size), k)
c = capi.Context(0)
c.load_reads(b, o, k)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    c.group_reads(groups)
    ts.append((time.perf_counter() - t0) * 1e3)
print("group_reads(%d) of %d reads: %s ms" % (groups, n, [round(x, 1) for x in ts]), flush=True)
c.run_pass(k, m, H)
tp = []
for _ in range(3):
    t0 = time.perf_counter()
    c.run_pass(k, m, H)
    tp.append((time.perf_counter() - t0) * 1e3)
print("pass: %s ms; distinct %d" % ([round(x, 1) for x in tp], c.count_stats()["n_distinct"]))
c.close()
