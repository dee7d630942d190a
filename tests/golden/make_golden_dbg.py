#!/usr/bin/env python3
"""Golden vectors for the de Bruijn stage from the REAL reference, `-t 1`:
the GFA written by Assemble<B> (reference src/Assemble.cpp:6-28) reduced to its ID-free canonical form
(tests/gfa_canon.py).  Small cases keep the whole canonical form; config #1 (10 k x 150 bp, k=21) keeps counts and a
sha256 of the sorted form.  The reference is run in a subprocess with a time limit: on some small repeat-rich inputs
its ExtendLeft loop never terminates (an unbranched cycle) -- those are not fixtures.
Inputs are ACGT-only: with unknown characters the reference's junction coverage depends on the orientation in which
a junction happened to be discovered (CountNodeCoverage compares the two strands of a window separately and an
unknown base is 00 on both), so no order-free expected value exists."""
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
from oracle.p3_oracle import Oracle, _flat  # noqa: E402
import platanus3_amd.synth as synth  # noqa: E402
from gfa_canon import canonical  # noqa: E402
from util import random_reads  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
WORKER = """
import sys, tempfile, numpy as np
sys.path.insert(0, %r)
from oracle.p3_oracle import RefLib
z = np.load(sys.argv[1])
gfa = RefLib().assemble_gfa((z["b"], z["off"]), int(z["k"]), int(z["m"]), int(z["H"]), 1, tempfile.mkdtemp())
open(sys.argv[2], "w").write(gfa)
""" % ROOT


def ref_gfa(b, off, k, m, H, timeout):
    d = tempfile.mkdtemp()
    np.savez(d + "/in.npz", b=b, off=off, k=k, m=m, H=H)
    with open(d + "/w.py", "w") as f:
        f.write(WORKER)
    subprocess.run([sys.executable, d + "/w.py", d + "/in.npz", d + "/out.gfa"], timeout=timeout, check=True,
                   capture_output=True)
    return open(d + "/out.gfa").read()


def lines_of(nodes, edges):
    out = []
    for (kind, seq, kc), n in sorted(nodes.items()):
        out.append("N\t%s\t%s\t%s\t%d" % (kind, seq, kc, n))
    for (a, b, ov), n in sorted(edges.items()):
        out.append("E\t%s\t%s\t%d\t%s\t%s\t%d\t%s\t%d" % (a[0], a[1], a[2], b[0], b[1], b[2], ov, n))
    return out


def main():
    o = Oracle()
    rng = np.random.default_rng(5)
    cases = []

    def add(name, reads, k, m=None, H=None, full=True, timeout=300):
        b, off = _flat(reads)
        ab = sum(int(off[i + 1] - off[i]) for i in range(off.size - 1) if off[i + 1] - off[i] >= k)
        if m is None:
            m, H = o.estimate_bloom(ab, k)
        gfa = ref_gfa(b, off, k, m, H, timeout)
        nodes, edges = canonical(gfa)
        ls = lines_of(nodes, edges)
        sha = hashlib.sha256("\n".join(ls).encode()).hexdigest()
        meta = dict(name=name, k=k, m=int(m), H=int(H), n_nodes=sum(nodes.values()), n_edges=sum(edges.values()),
                    n_junctions=sum(v for (kd, _, _), v in nodes.items() if kd == "Junction"),
                    n_straights=sum(v for (kd, _, _), v in nodes.items() if kd == "Straight"), sha256=sha, full=full)
        np.savez_compressed(os.path.join(OUT, "dbg_" + name + "_reads.npz"), bases=b, offsets=off)
        if full:
            with gzip.open(os.path.join(OUT, "dbg_" + name + ".canon.gz"), "wt") as f:
                f.write("\n".join(ls))
        cases.append(meta)
        print(meta, flush=True)

    b, off = synth.reads(seed=1, total_reads=400, read_len=150)
    add("synth400_k21", (b, off), 21)
    b, off = synth.reads(seed=1, total_reads=2000, read_len=150)
    add("synth2000_k31", (b, off), 31)
    add("synth2000_k31_m24", (b, off), 31, 1 << 24, 10)
    b, off = synth.reads(seed=3, total_reads=1000, read_len=150)
    add("synth1000_k41", (b, off), 41, 1 << 22, 10)
    b, off = synth.reads(seed=4, total_reads=40, read_len=2000, sub_rate_ppm=10000)
    add("long40_k63", (b, off), 63, 1 << 22, 10)
    b, off = synth.reads(seed=6, total_reads=3000, read_len=150)
    add("synth3000_k32", (b, off), 32)
    add("synth3000_k51", (b, off), 51, 1 << 23, 10)
    add("synth3000_k64", (b, off), 64, 1 << 23, 10)
    rr = random_reads(rng, 300, lo=40, hi=320, genome_len=6000, unknown=False)
    add("ragged_k25_m20", rr, 25, 1 << 20, 10)
    # BASELINE config #1: 10 k x 150 bp, k = 21, default sizing -- checksum only (105 k nodes)
    b, off = synth.reads(seed=1, total_reads=10000, read_len=150)
    add("config1_k21", (b, off), 21, full=False)
    with open(os.path.join(OUT, "dbg_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)


if __name__ == "__main__":
    main()
