import sys, json
sys.path.insert(0,'.')
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing
b,o=synth.reads(seed=2,total_reads=1_000_000,read_len=150)
m,H=bloom_sizing(int(b.size),31)
c=capi.Context(0)
c.load_reads(b,o,31)
for dbg in (0,1,2,4,5):
    c.set_option("dbg",dbg)
    c.run_pass(31,m,H)
    c.reset_kernel_stats(); c.set_profiling(True)
    for _ in range(3): c.run_pass(31,m,H)
    c.set_profiling(False)
    st=c.kernel_stats()
    print("dbg",dbg, {k:round(v["total_ms"]/v["launches"],3) for k,v in st.items() if k in ("k_kmer_dedup","k_part_count")}, c.get_info("distinct_kmers"))
