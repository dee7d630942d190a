#!/usr/bin/env python3
"""De Bruijn goldens for the reference's long-read key widths -- k = 501, 1001, 2001, 3001 (src/Assemble.cpp:42-49: keys of
16, 32, 63 and 94 words) -- and for widths in between (k = 129, 160, 600, 1500, 2500) from the REAL reference, `-t 1`: Assemble<std::bitset<2k>> (src/Assemble.cpp:6-28) -> GFA in
ID-free canonical form + the JointInfo::coverage lines, exactly as make_golden_dbg_wide.py does for 65 <= k <= 128.
ACGT-only inputs.  Output: tests/golden/dbgl_cases.json, dbgl_<name>.canon.gz, dbgl_<name>_reads.npz.  Build container only."""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle.p3_oracle import _flat  # noqa: E402
import platanus3_amd.synth as synth  # noqa: E402
from gfa_canon import canonical  # noqa: E402
from make_golden_dbg import lines_of  # noqa: E402
from make_golden_dbg_wide import ref_run  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    path = os.path.join(OUT, "dbgl_cases.json")
    cases = json.load(open(path)) if os.path.exists(path) else []   # (the reference takes minutes per case: new ones only)

    def add(name, reads, k, m, H, timeout=1200):
        if any(c["name"] == name for c in cases):
            return
        b, off = _flat(reads)
        gfa, joints = ref_run(b, off, k, m, H, timeout)
        nodes, edges = canonical(gfa)
        ls = lines_of(nodes, edges)
        jl = ["%s\t%d" % (a, c) for a, c in joints]
        meta = dict(name=name, k=k, m=int(m), H=int(H), n_nodes=sum(nodes.values()), n_edges=sum(edges.values()),
                    n_junctions=sum(v for (kd, _, _), v in nodes.items() if kd == "Junction"),
                    n_straights=sum(v for (kd, _, _), v in nodes.items() if kd == "Straight"),
                    sha256=hashlib.sha256("\n".join(ls).encode()).hexdigest(), full=True,
                    joints=dict(n=len(joints), coverage_sum=int(sum(c for _, c in joints)),
                                sha256=hashlib.sha256("\n".join(jl).encode()).hexdigest()))
        np.savez_compressed(os.path.join(OUT, "dbgl_" + name + "_reads.npz"), bases=b, offsets=off)
        with gzip.open(os.path.join(OUT, "dbgl_" + name + ".canon.gz"), "wt") as f:
            f.write("\n".join(ls))
        cases.append(meta)
        print(meta, flush=True)

    # long reads with few errors (a window of k bases must hold none to pass the gate)
    b, off = synth.reads(seed=91, total_reads=120, read_len=2500, sub_rate_ppm=600)
    add("long120_k501", (b, off), 501, 1 << 22, 10)
    add("long120_k1001", (b, off), 1001, 1 << 22, 10)
    b, off = synth.reads(seed=92, total_reads=40, read_len=5000, sub_rate_ppm=250)
    add("long40_k2001", (b, off), 2001, 1 << 22, 10, timeout=2400)
    add("long40_k3001", (b, off), 3001, 1 << 22, 10, timeout=2400)
    add("long40_k501", (b, off), 501, 1 << 22, 10)
    # a denser vector: false-positive neighbours add junctions and short branches
    b, off = synth.reads(seed=93, total_reads=300, read_len=1500, sub_rate_ppm=1500)
    add("dense300_k501", (b, off), 501, 1 << 19, 3)
    add("dense300_k1001", (b, off), 1001, 1 << 19, 3)
    # widths between the reference's own: they run in the next wider instance of the graph kernels, upper key words zero
    b, off = synth.reads(seed=94, total_reads=400, read_len=600, sub_rate_ppm=2000)
    add("mid400_k129", (b, off), 129, 1 << 19, 3)
    add("mid400_k160", (b, off), 160, 1 << 22, 10)
    b, off = synth.reads(seed=95, total_reads=60, read_len=4000, sub_rate_ppm=400)
    add("mid60_k600", (b, off), 600, 1 << 22, 10)
    add("mid60_k1500", (b, off), 1500, 1 << 19, 3, timeout=2400)
    add("mid60_k2500", (b, off), 2500, 1 << 22, 10, timeout=2400)
    with open(os.path.join(OUT, "dbgl_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)


if __name__ == "__main__":
    main()
