"""Debug driver: one config-#2-shaped pass with failure tracing and per-kernel times (run on the GPU box)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("P3_TRACE_ERR", "1")
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
opts = sys.argv[3:]
b, o = synth.reads(seed=2, total_reads=n, read_len=150)
m, H = bloom_sizing(int(b.size), k)
ctx = capi.Context(0)
for kv in opts:
    name, v = kv.split("=")
    ctx.set_option(name, int(v))
ctx.load_reads(b, o, min_len=k)
for it in range(4):
    ctx.reset_kernel_stats()
    ctx.set_profiling(it >= 2)
    t = time.perf_counter()
    ctx.run_pass(k, m, H)
    dt = time.perf_counter() - t
    print("pass %d: %.3f ms host; run_records=%s sbits=%s ksbits=%s est_level=%s" % (
        it, dt * 1e3, ctx.get_info("run_records"), ctx.get_info("sbits"), ctx.get_info("ksbits"), ctx.get_info("est_level")), flush=True)
ks = ctx.kernel_stats()
tot = 0.0
for name, s in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("  %-22s %3d launches %8.3f ms" % (name, s["launches"], s["total_ms"]))
    tot += s["total_ms"]
print("  kernels total %.3f ms" % tot)
print("count", ctx.count_stats(), "bloom", ctx.bloom_stats())
