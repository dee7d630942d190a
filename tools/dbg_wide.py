import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platanus3_amd import capi
from oracle.p3_oracle import Oracle
from tests.golden.make_golden_wide import read_set
o_ = Oracle()
name, k = sys.argv[1], int(sys.argv[2])
b, o = read_set(name)
m, H = 50000017, 10
ekeys, ecounts = o_.count_short((b, o), min_len=k)
ewords, eseeds, n_ins, _ = o_.make_bf((b, o), ekeys, ecounts, k, m, H)
ctx = capi.Context(0)
ctx.load_reads(b, o, min_len=k)
ctx.run_pass(k, m, H)
w = ctx.export_bloom()
print("passing", ctx.bloom_stats(), "oracle inserted", n_ins)
dbits = np.unpackbits(w.view(np.uint8), bitorder="little")
ebits = np.unpackbits(ewords.view(np.uint8), bitorder="little")
extra = set(np.nonzero(dbits & ~ebits)[0].tolist())
missing = set(np.nonzero(~dbits & ebits)[0].tolist())
print("extra bits", len(extra), "missing", len(missing))
n = o.size - 1
for r in range(n):
    s = bytes(b[int(o[r]):int(o[r + 1])])
    L = len(s)
    if L < k:
        continue
    for p in range(L - k + 1):
        km = s[p:p + k]
        cn = o_.canonical(o_.encode_fwd(km), o_.encode_rc(km), k)
        h1, h2 = o_.double_hash(cn, k)
        pos = [int(x) for x in o_.probe_positions(h1, h2, m, H)]
        if any(x in extra for x in pos) and all(dbits[x] for x in pos):
            flags = o_.gate_flags(s, ekeys, ecounts, k)
            print("read", r, "L", L, "window", p, "oracle pass flag", int(flags[p]), "stream pos", int(o[r]) + p, "word", (int(o[r]) + p) // 32, "j", (int(o[r]) + p) % 32, km[:30])
