"""Experiment: may a probe partition hold 2^32 positions instead of 2^31?  Same input, both limits: pass time and the
sha256 of the Bloom vector must agree.  usage: python tools/exp_pos_limit.py [reads]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bloom_sizing
from platanus3_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
k = 31
b, o = synth.reads(seed=3, total_reads=n, read_len=150)
m, H = bloom_sizing(n * 150, k)
c = capi.Context(0)
c.load_reads(b, o, k)
for lim in (31, 32):
    c.set_option("pos_limit_log2", lim)
    for i in range(3):
        t0 = time.perf_counter(); c.run_pass(k, m, H); dt = time.perf_counter() - t0
    st = c.kernel_stats()
    w = c.export_bloom()
    print("limit 2^%d: %.1f ms per pass, bloom sha256 %s, bloom_mode %d" % (lim, dt * 1e3, hashlib.sha256(w.tobytes()).hexdigest()[:16], c.get_info("bloom_mode")), flush=True)
