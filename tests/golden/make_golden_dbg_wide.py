#!/usr/bin/env python3
"""De Bruijn goldens for keys of three and four words (65 <= k <= 128; the reference's own k = 101) from the REAL
reference, `-t 1`: Assemble<std::bitset<2k>> (src/Assemble.cpp:6-28) -> GFA in ID-free canonical form + the
JointInfo::coverage lines (as make_golden_dbg.py / make_golden_joints.py do for the short keys).  ACGT-only inputs.
Output: tests/golden/dbgw_cases.json, dbgw_<name>.canon.gz, dbgw_<name>_reads.npz.  Build container only."""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.p3_oracle import _flat  # noqa: E402
import platanus3_amd.synth as synth  # noqa: E402
from gfa_canon import canonical  # noqa: E402
from util import random_reads  # noqa: E402
from make_golden_dbg import lines_of  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
WORKER = """
import sys, json, tempfile, numpy as np
sys.path.insert(0, %r)
from oracle.p3_oracle import RefLib
z = np.load(sys.argv[1])
gfa, j = RefLib().assemble_gfa((z["b"], z["off"]), int(z["k"]), int(z["m"]), int(z["H"]), 1, tempfile.mkdtemp(), joints=True)
open(sys.argv[2], "w").write(gfa)
json.dump(j, open(sys.argv[2] + ".joints", "w"))
""" % ROOT


def ref_run(b, off, k, m, H, timeout):
    d = tempfile.mkdtemp()
    np.savez(d + "/in.npz", b=b, off=off, k=k, m=m, H=H)
    with open(d + "/w.py", "w") as f:
        f.write(WORKER)
    subprocess.run([sys.executable, d + "/w.py", d + "/in.npz", d + "/out.gfa"], timeout=timeout, check=True, capture_output=True)
    return open(d + "/out.gfa").read(), json.load(open(d + "/out.gfa.joints"))


def main():
    rng = np.random.default_rng(9)
    cases = []

    def add(name, reads, k, m, H, timeout=600):
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
        np.savez_compressed(os.path.join(OUT, "dbgw_" + name + "_reads.npz"), bases=b, offsets=off)
        with gzip.open(os.path.join(OUT, "dbgw_" + name + ".canon.gz"), "wt") as f:
            f.write("\n".join(ls))
        cases.append(meta)
        print(meta, flush=True)

    b, off = synth.reads(seed=81, total_reads=1500, read_len=250, sub_rate_ppm=4000)
    add("synth1500_k101", (b, off), 101, 1 << 23, 10)
    add("synth1500_k65", (b, off), 65, 1 << 23, 10)
    add("synth1500_k128", (b, off), 128, 1 << 23, 10)
    add("synth1500_k96", (b, off), 96, 1 << 22, 7)
    b, off = synth.reads(seed=82, total_reads=40, read_len=3000, sub_rate_ppm=8000)
    add("long40_k101", (b, off), 101, 1 << 22, 10)
    add("long40_k127", (b, off), 127, 1 << 22, 10)
    rr = random_reads(rng, 250, lo=60, hi=500, genome_len=8000, unknown=False)
    add("ragged_k97", rr, 97, 1 << 21, 10)
    add("ragged_k80", rr, 80, 1 << 21, 10)
    # denser vectors: false-positive neighbours add junctions and short branches
    b, off = synth.reads(seed=83, total_reads=3000, read_len=250, sub_rate_ppm=6000)
    add("dense3000_k101", (b, off), 101, 1 << 20, 3)
    add("dense3000_k80", (b, off), 80, 1 << 20, 3)
    with open(os.path.join(OUT, "dbgw_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)


if __name__ == "__main__":
    main()
