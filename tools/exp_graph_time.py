import sys, time, json
sys.path.insert(0,'.')
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing
for n,k in ((10_000,21),(1_000_000,31)):
    b,o=synth.reads(seed=1 if n==10000 else 2,total_reads=n,read_len=150)
    m,H=bloom_sizing(int(b.size),k)
    c=capi.Context(0); c.load_reads(b,o,k); c.run_pass(k,m,H)
    c.build_graph()
    c.reset_kernel_stats(); c.set_profiling(True)
    t=time.perf_counter(); st=c.build_graph(); dt=time.perf_counter()-t
    c.set_profiling(False)
    ks={kk:round(v["total_ms"],3) for kk,v in c.kernel_stats().items() if kk.startswith("k_dbg")}
    print(json.dumps(dict(reads=n,k=k,m=m,H=H,wall_ms=round(dt*1e3,2),stats=st,kernels_ms=ks)))
    c.close()
