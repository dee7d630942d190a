"""Per-kernel average durations of the single-GPU pass on config#2 (experiment helper; GPU box only)."""
import sys, json, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing

k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
b, o = synth.reads(seed=2, total_reads=1_000_000, read_len=150)
m, H = bloom_sizing(int(b.size), k)
c = capi.Context(0)
for a in sys.argv[2:]:
    if '=' in a:
        n, v = a.split('=')
        c.set_option(n, int(v))
c.load_reads(b, o, k)
quick = 'quick' in sys.argv
for _ in range(1 if quick else 3):
    c.run_pass(k, m, H)
if quick:
    c.run_pass(k, m, H)
    sys.exit(0)
t = []
for _ in range(10):
    c.run_pass(k, m, H)
    t.append(c.timings()["ms_total"])
print("ms_total", [round(x, 3) for x in t], "min", round(min(t), 3))
c.reset_kernel_stats(); c.set_profiling(True)
for _ in range(3):
    c.run_pass(k, m, H)
c.set_profiling(False)
st = c.kernel_stats()
tot = 0
for name, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
    per = v["total_ms"] / 3
    tot += per
    print(f"{name:24s} {v['launches']//3:3d} launches  {per:7.3f} ms/pass")
print("sum", round(tot, 3), "distinct", c.get_info("distinct_kmers"))
