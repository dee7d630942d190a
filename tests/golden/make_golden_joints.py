#!/usr/bin/env python3
"""JointInfo::coverage goldens from the REAL reference (`-t 1`), for the de Bruijn cases of make_golden_dbg.py.
The reference fills the field in AddNodeCoverage (src/DeBruijnGraph.cpp:441-449) but never prints it; the harness reads
it from DeBruijnGraph::joints after CountNodeCoverage.  Stored ID-free: sha256 over the sorted lines
"<joint k-mer in its smaller orientation>\\t<coverage>", plus the count and the coverage sum (small cases keep the lines).
Output: tests/golden/dbg_joints.json (data only).  Build container only."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.p3_oracle import RefLib  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    with open(os.path.join(OUT, "dbg_cases.json")) as f:
        cases = json.load(f)
    out = {}
    for c in cases:
        z = np.load(os.path.join(OUT, "dbg_%s_reads.npz" % c["name"]))
        _, j = RefLib().assemble_gfa((z["bases"], z["offsets"]), c["k"], c["m"], c["H"], 1, tempfile.mkdtemp(), joints=True)
        lines = ["%s\t%d" % (a, b) for a, b in j]
        e = dict(n=len(j), coverage_sum=int(sum(b for _, b in j)), sha256=hashlib.sha256("\n".join(lines).encode()).hexdigest())
        if len(j) <= 200:
            e["lines"] = lines
        out[c["name"]] = e
        print(c["name"], e["n"], e["coverage_sum"], flush=True)
    with open(os.path.join(OUT, "dbg_joints.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
