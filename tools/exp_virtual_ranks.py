"""Time pipeline.VirtualRanks (several protocol ranks on ONE GPU) against the single-context pass on the same reads.
usage: [RESHARD=1] [VR_STEPS=n] python tools/exp_virtual_ranks.py [total_reads] [per_rank] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bloom_sizing  # noqa: E402
from platanus3_amd import synth  # noqa: E402
from platanus3_amd.pipeline import Pipeline, VirtualRanks  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
v = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
m, H = bloom_sizing(n * 150, k)
vr = VirtualRanks(device=0, rank=0, world=1, per_rank=v, seed=1, total_reads=n, first_read=0, n_reads=n)
for kv in [x for x in os.environ.get("VR_OPTS", "").split(",") if x]:  # e.g. VR_OPTS=mr_aggregate=0
    name, val = kv.split("=")
    for p in vr.pipes:
        p.ctx.set_option(name, int(val))
if os.environ.get("RESHARD", "0") == "1":
    t0 = time.perf_counter()
    vr.reshard()
    print("re-sharded by read minimizer in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
print("form:", vr.parallelism(), flush=True)
for step in range(int(os.environ.get("VR_STEPS", "4"))):
    t0 = time.perf_counter()
    vr.run_pass(k, m, H)
    print("virtual ranks pass %d: %.1f ms" % (step, (time.perf_counter() - t0) * 1e3), flush=True)
if os.environ.get("VR_PROFILE", "0") == "1":   # per-kernel times of rank 0 (events around every launch; the other ranks share the GPU)
    c0 = vr.pipes[0].ctx
    c0.reset_kernel_stats()
    c0.set_profiling(True)
    vr.run_pass(k, m, H)
    c0.set_profiling(False)
    st = c0.kernel_stats()
    print("rank 0 kernels: " + ", ".join("%s=%.2f(%d)" % (nm, v_["total_ms"], v_["launches"])
                                         for nm, v_ in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"])[:16]), flush=True)
    print("rank 0 kernel total %.2f ms" % sum(v_["total_ms"] for v_ in st.values()), flush=True)
dv, pv = vr.global_distinct(), vr.passing_kmers()
vr.close()
pipe = Pipeline(device=0)
b, o = synth.reads(seed=1, total_reads=n, first_read=0, n_reads=n, read_len=150)
pipe.load_reads(b, o, min_len=21)
for step in range(3):
    t0 = time.perf_counter()
    pipe.run_pass(k, m, H)
    print("single context pass %d: %.1f ms" % (step, (time.perf_counter() - t0) * 1e3), flush=True)
d1, p1 = pipe.global_distinct(), pipe.passing_kmers()
print("distinct 21-mers: %d vs %d; passing windows: %d vs %d" % (dv, d1, pv, p1))
assert dv == d1 and pv == p1
