#!/usr/bin/env python3
"""Full-size golden CHECKSUMS for BASELINE config #2 (1 M x 150 bp, k=31) from the REAL reference
(oracle/_ref/libp3ref.so): sha256 of the sorted key / count arrays, of the Bloom words (default sizing and the
"-m 2^30" H=10 run) and of the seed list.  ~6 minutes of single-threaded reference time; build container only.
Output: tests/golden/config2_checksums.json (data only)."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.p3_oracle import RefLib  # noqa: E402
import platanus3_amd.synth as synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = RefLib()
    L = ref.lib
    k, n_reads = 31, 1_000_000
    b, o = synth.reads(seed=2, total_reads=n_reads, read_len=150)
    h = L.p3ref_open(b, o, n_reads, k)
    cap = int(b.size)
    keys, counts = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    n = C.c_uint64()
    t = time.time()
    assert L.p3ref_count_short(h, keys, counts, cap, C.byref(n)) == 0
    t_count = time.time() - t
    keys, counts = keys[:n.value], counts[:n.value]
    out = dict(seed=2, n_reads=n_reads, read_len=150, k=k, n_distinct=int(n.value), n_solid=int((counts >= 2).sum()),
               n_windows=int(counts.sum()), max_count=int(counts.max()), keys_sha256=sha(keys), counts_sha256=sha(counts),
               t_count_s=t_count, runs=[])
    for m_in in (0, 1 << 30):
        m, H = ref.estimate_bloom(int(b.size), k, m_in)
        words = np.zeros((m + 63) // 64, np.uint64)
        seed_chars = np.zeros(n_reads * k, np.uint8)
        ns = C.c_uint64()
        t = time.time()
        assert L.p3ref_make_bf(h, k, m, H, words, seed_chars, n_reads, C.byref(ns)) == 0
        seeds = seed_chars[:ns.value * k]
        out["runs"].append(dict(m=int(m), H=int(H), bloom_sha256=sha(words),
                                bits_set=int(np.unpackbits(words.view(np.uint8)).sum()), n_seeds=int(ns.value),
                                seeds_sha256=sha(seeds), t_makebf_s=time.time() - t))
        print(out["runs"][-1], flush=True)
    L.p3ref_close(h)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config2_checksums.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
