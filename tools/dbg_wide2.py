import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platanus3_amd import capi
from oracle.p3_oracle import Oracle
from tests.golden.make_golden_wide import read_set
o_ = Oracle()
b, o = read_set("ragged")
ctx = capi.Context(0)
for k in (31, 63, 64, 65, 66, 80):
    for gate in (2, 1):
        m, H = 50000017, 10
        ekeys, ecounts = o_.count_short((b, o), min_len=k)
        ewords, eseeds, n_ins, _ = o_.make_bf((b, o), ekeys, ecounts, k, m, H)
        ctx.set_option("gate_mode", gate)
        ctx.load_reads(b, o, min_len=k)
        ctx.run_pass(k, m, H)
        w = ctx.export_bloom()
        print(k, "gate", gate, "passing", ctx.bloom_stats()["n_passing"], "oracle", n_ins, "equal", np.array_equal(w, ewords), flush=True)
