"""diagnostic: the Bloom vector must not depend on the order of the reads (p3_group_reads) -- compare sha / bits set"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi
from bench import bloom_sizing
n = int(sys.argv[1])
k = 31
b, o = synth.reads(seed=3, total_reads=n, read_len=150)
m, H = bloom_sizing(int(b.size), k)
lut = np.array([bin(i).count("1") for i in range(256)], np.uint64)
ref = None
for spec in sys.argv[2:]:
    grp, _, opts = spec.partition(":")
    c = capi.Context(0)
    for kv in [x for x in opts.split(",") if x]:
        a, v = kv.split("=")
        c.set_option(a, int(v))
    c.load_reads(b, o, k)
    if int(grp):
        c.group_reads(int(grp))
    c.run_pass(k, m, H)
    w = c.export_bloom()
    bits = int(lut[w.view(np.uint8)].sum())
    extra = ""
    if ref is None:
        ref = w
    else:
        extra = " | missing %d extra %d" % (int(lut[(ref & ~w).view(np.uint8)].sum()), int(lut[(w & ~ref).view(np.uint8)].sum()))
    print("[%s] chunks=%d groups=%d ksplit=%d ksbits=%d distinct_k=%d bits=%d sha=%s%s" % (
        spec, c.get_info("chunks"), c.get_info("kmer_groups"), c.get_info("ksplit_xbits"), c.get_info("ksbits"),
        c.get_info("distinct_kmers"), bits, hashlib.sha256(w.tobytes()).hexdigest()[:12], extra), flush=True)
    c.close()
