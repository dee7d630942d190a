"""Timing of one pass on a large single-GPU input (position chunks), no exports.  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing
R = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
ppm = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
t0 = time.time()
b, o = synth.reads(seed=3, total_reads=R, read_len=L, sub_rate_ppm=ppm)
skew = float(os.environ.get("P3_SKEW", "0"))   # fraction of the reads that get a 25-base poly-A stretch (skewed keys)
if skew > 0:
    import numpy as np
    starts = o[:-1][np.random.default_rng(1).choice(R, size=int(R * skew), replace=False)].astype(np.int64) + 40
    idx = (starts[:, None] + np.arange(25)[None, :]).ravel()
    b[idx] = ord("A")
m, H = bloom_sizing(int(b.size), k)
print("%d x %d bp = %.2f G bases, m=%d H=%d (synth %.1f s)" % (R, L, b.size / 1e9, m, H, time.time() - t0), flush=True)
c = capi.Context(0)
for kv in os.environ.get("P3_OPTS", "").split(","):   # e.g. P3_OPTS=kmer_split=3,kmer_collect=0
    if kv:
        c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
t0 = time.time(); c.load_reads(b, o, k); print("load %.1f s" % (time.time() - t0), flush=True)
if os.environ.get("P3_GROUP", "0") != "0":   # reads put in the order of their shard key's group (p3_group_reads), once
    t0 = time.time(); c.group_reads(int(os.environ["P3_GROUP"]) if os.environ["P3_GROUP"] != "1" else 0); print("group_reads %.2f s" % (time.time() - t0), flush=True)
for i in range(int(os.environ.get('P3_PASSES', '2'))):
    t0 = time.perf_counter(); c.run_pass(k, m, H); dt = time.perf_counter() - t0
    nk = R * (L - k + 1)
    print("pass %d: %.1f ms device, %.1f ms wall = %.2f G k-mers/s; chunks=%d split_xbits=%d count_mode=%d bloom_mode=%d" %
          (i, c.timings()["ms_total"], dt * 1e3, nk / dt / 1e9, c.get_info("chunks"), c.get_info("split_xbits"), c.get_info("count_mode"), c.get_info("bloom_mode")), flush=True)
    import ctypes as _C
    c.L.p3_last_error.restype = _C.c_char_p
    print("   est_fallbacks=%d est_level=%d ksplit_xbits=%d ksbits=%d kmer_groups=%d distinct=%d last_error=%r" % (c.get_info("est_fallbacks"), c.get_info("est_level"), c.get_info("ksplit_xbits"), c.get_info("ksbits"), c.get_info("kmer_groups"), c.get_info("distinct_kmers"), c.L.p3_last_error(c.h)), flush=True)
print(c.count_stats(), c.bloom_stats(), flush=True)
assert c.count_stats()["n_windows"] == R * (L - 20)
c.reset_kernel_stats(); c.set_profiling(True); c.run_pass(k, m, H); c.set_profiling(False)
st = c.kernel_stats()
print(", ".join("%s=%.1f(%d)" % (n, v["total_ms"], v["launches"]) for n, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"])[:12]), flush=True)
