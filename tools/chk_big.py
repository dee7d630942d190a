"""diagnostic: Bloom vector of config #3 at full size under several option sets (sha256 + bits set)"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import platanus3_amd.synth as synth
from platanus3_amd import capi
g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "full_config3.json")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else g["n_reads"]
k, m, H = g["k"], g["m"], g["H"]
b, o = synth.reads(seed=3, total_reads=g["n_reads"], n_reads=n, first_read=0, read_len=150)
lut = np.array([bin(i).count("1") for i in range(256)], np.uint64)
ref = None
for opts in sys.argv[2:] or [""]:
    c = capi.Context(0)
    for kv in [x for x in opts.split(",") if x]:
        a, v = kv.split("=")
        c.set_option(a, int(v))
    c.load_reads(b, o, k)
    npass = 1
    if c.__dict__.get("_grp"):
        pass
    if "GROUP" in os.environ.get("CHK", ""):
        c.group_reads(0)
    if "TWICE" in os.environ.get("CHK", ""):
        c.run_pass(k, m, H)
    c.run_pass(k, m, H)
    w = c.export_bloom()
    bits = int(lut[w.view(np.uint8)].sum())
    sha = hashlib.sha256(w.tobytes()).hexdigest()[:16]
    extra = ""
    if ref is None:
        ref = w
    else:
        d = w ^ ref
        extra = " differs from the first in %d words; first-only bits %d, this-only bits %d" % (
            int((d != 0).sum()), int(lut[(ref & ~w).view(np.uint8)].sum()), int(lut[(w & ~ref).view(np.uint8)].sum()))
    print("[%s] chunks=%d kmer_groups=%d bits=%d sha=%s passing=%d%s" % (opts, c.get_info("chunks"), c.get_info("kmer_groups"), bits, sha,
                                                                  c.bloom_stats()["n_passing"], extra), flush=True)
    c.close()
print("golden bits", g["bits_set"], g["bloom_sha256"][:16])
