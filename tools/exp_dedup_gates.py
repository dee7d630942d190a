"""Timing experiment (GPU box only; results of the gated runs are WRONG by design): where a k_kmer_dedup<2> workgroup's
time goes -- the kernel's dbg gates return early after the clear (8), before the cursor (16), before the emission (4), or
replace the global cursor by a fixed base (2).  usage: exp_dedup_gates.py <reads> <len> <k>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing
R, L, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b, o = synth.reads(seed=3, total_reads=R, read_len=L, sub_rate_ppm=10000)
m, H = bloom_sizing(int(b.size), k)
c = capi.Context(0)
for kv in os.environ.get("P3_OPTS", "").split(","):
    if kv:
        c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
c.load_reads(b, o, k)
for _ in range(3):
    c.run_pass(k, m, H)
for dbg in [int(x) for x in os.environ.get('GATES', '0,32,64,0').split(',')]:
    c.set_option("dbg", dbg)
    c.reset_kernel_stats(); c.set_profiling(True)
    try:
        c.run_pass(k, m, H)
    except Exception as e:  # the gated runs may trip the pass's own checks: the kernel times were taken all the same
        print("   (pass raised: %s)" % str(e)[:80])
    c.set_profiling(False)
    st = c.kernel_stats()
    print("dbg=%2d  " % dbg + "  ".join("%s=%.2f ms/%d = %.1f us" % (n, v["total_ms"], v["launches"], 1e3 * v["total_ms"] / max(1, v["launches"]))
                                       for n, v in st.items() if n in ("k_kmer_dedup", "k_kmer_dedup16", "k_split")), flush=True)
    c.set_option("dbg", 0)
    c.drop_hints() if hasattr(c, "drop_hints") else None
    c.run_pass(k, m, H)
