"""Debug driver: stage-by-stage comparison of the run-record path with the oracle on a small input (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("P3_TRACE_ERR", "1")
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi
from oracle.p3_oracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
b, o = synth.reads(seed=7, total_reads=n, read_len=150)
orc = Oracle()
m, H = orc.estimate_bloom(int(b.size), k)
ctx = capi.Context(0)
ctx.set_option("run_records", 2)
for kv in sys.argv[3:]:
    name, v = kv.split("=")
    ctx.set_option(name, int(v))
ctx.load_reads(b, o, min_len=k)
ctx.count_short()
print("after count: run_records", ctx.get_info("run_records"), "sbits", ctx.get_info("sbits"), ctx.count_stats(), flush=True)
keys, counts = ctx.export_counts()
ek, ec = orc.count_short((b, o), min_len=k)
print("distinct", keys.size, "expected", ek.size, "sum", int(counts.sum()), "expected", int(ec.sum()))
if not (np.array_equal(keys, ek) and np.array_equal(counts, ec)):
    common = np.intersect1d(keys, ek)
    print("MISMATCH: common keys", common.size, "only ours", keys.size - common.size, "only oracle", ek.size - common.size)
    if keys.size == ek.size and np.array_equal(keys, ek):
        bad = np.nonzero(counts != ec)[0]
        print("count diffs", bad.size, [(hex(int(keys[i])), int(counts[i]), int(ec[i])) for i in bad[:8]])
    else:
        print("ours not in oracle:", [hex(int(x)) for x in np.setdiff1d(keys, ek)[:6]])
        print("oracle not in ours:", [hex(int(x)) for x in np.setdiff1d(ek, keys)[:6]])
else:
    print("count table OK")
print("nonsolid windows", ctx.get_info("nonsolid_windows"), "expected", int((ec == 1).sum()))
ctx.make_bloom(k, m, H)
print("after bloom: run_records", ctx.get_info("run_records"), "ksbits", ctx.get_info("ksbits"), ctx.bloom_stats(), flush=True)
eb, es, _, _ = orc.make_bf((b, o), ek, ec, k, m, H)
ob = ctx.export_bloom()
print("bloom words equal:", np.array_equal(ob, eb), "bits ours", int(np.unpackbits(ob.view(np.uint8)).sum()), "expected", int(np.unpackbits(eb.view(np.uint8)).sum()))
seeds = [capi.decode_key(r, k) for r in ctx.export_seeds()]
print("seeds equal:", seeds == es, len(seeds), len(es))
