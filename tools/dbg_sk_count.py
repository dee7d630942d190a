"""Timing experiments on the count stage only (GPU box): python tools/dbg_sk_count.py <reads> opt=val ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import platanus3_amd.synth as synth
from platanus3_amd import capi

n = int(sys.argv[1])
b, o = synth.reads(seed=2, total_reads=n, read_len=150)
for variant in sys.argv[2:]:
    ctx = capi.Context(0)
    for kv in variant.split(","):
        if kv:
            name, v = kv.split("=")
            ctx.set_option(name, int(v))
    ctx.load_reads(b, o, min_len=31)
    try:
        ctx.count_short()
        ctx.reset_kernel_stats()
        ctx.set_profiling(True)
        ctx.count_short()
        ctx.set_profiling(False)
        ks = ctx.kernel_stats()
        print(variant, "sbits", ctx.get_info("sbits"), " ".join("%s=%.3f" % (k, v["total_ms"]) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"])[:4]), flush=True)
    except Exception as e:
        print(variant, "FAILED", e, flush=True)
    ctx.close()
