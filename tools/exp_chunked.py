"""Input beyond 2^30 window positions on one GPU: position chunks (fast path) vs the direct kernels.  GPU box only."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
ppm = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
b, o = synth.reads(seed=11, total_reads=R, read_len=L, sub_rate_ppm=ppm)
m, H = bloom_sizing(int(b.size), k)
print("%d reads, %.2f G bases, m=%d" % (R, b.size / 1e9, m), flush=True)
h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
res = {}
for mode in ("chunked", "direct"):
    c = capi.Context(0)
    if mode == "direct":
        c.set_option("chunks", 1)
        c.set_option("bloom_slices", 0)   # and global atomicOr per probe instead of the (multi-sweep) slices
    c.load_reads(b, o, k)
    c.run_pass(k, m, H)
    t0 = time.perf_counter()
    c.run_pass(k, m, H)
    dt = time.perf_counter() - t0
    print("  %s: pass %.1f ms (wall %.1f ms) chunks=%d count_mode=%d bloom_mode=%d split_xbits=%d %s" %
          (mode, c.timings()["ms_total"], dt * 1e3, c.get_info("chunks"), c.get_info("count_mode"), c.get_info("bloom_mode"),
           c.get_info("split_xbits"), c.bloom_stats()), flush=True)
    if mode == "chunked":
        c.reset_kernel_stats(); c.set_profiling(True); c.run_pass(k, m, H); c.set_profiling(False)
        st = c.kernel_stats()
        print("     " + ", ".join("%s=%.1f" % (n, v["total_ms"]) for n, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"])[:9]), flush=True)
    keys, counts = c.export_counts()
    res[mode] = (h(keys), h(counts), h(c.export_bloom()), h(c.export_seeds()), int(keys.size))
    print("  ", res[mode], flush=True)
    c.close()
print("MATCH" if res["chunked"] == res["direct"] else "*** MISMATCH ***")
