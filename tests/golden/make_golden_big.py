#!/usr/bin/env python3
"""Large-input golden CHECKSUMS from the REAL reference (oracle/_ref/libp3ref.so), build container only.

The inputs are the sizes at which the device path leaves its one-partition form (second-level partitions, position
chunks, collected k-mer groups, 2-word keys, m > 2^32), so that machinery is pinned against the reference itself and
not against another device path:

  shard3   6.25 M x 150 bp, k=31          BASELINE config #3's per-GPU shard at 8 GPUs      (~25 min, ~14 GB)
  c5k41    10 M x 150 bp,  k=41           BASELINE config #5 with 2-word keys               (~60 min, ~22 GB)
  long63   120 k x 10 kbp, k=63, -m 5e9   config #4's shape, > 2^30 window positions, m > 2^32, H=10 (~30 min, ~30 GB)

Usage: make_golden_big.py <case> [...]   Output: tests/golden/big_<case>.json (data only).
sha256 of: sorted canonical 21-mer keys (u64), their counts (u64), Bloom words (u64, bit i = word i/64 >> i%64),
seed strings (sorted, concatenated, k chars each) -- the same forms tests/test_gpu_parity.py hashes on the device side.
"""
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

OUTDIR = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "shard3": dict(seed=3, n_reads=6_250_000, read_len=150, sub_rate_ppm=5000, k=31, m_in=0),
    "c5k41": dict(seed=5, n_reads=10_000_000, read_len=150, sub_rate_ppm=5000, k=41, m_in=0),
    "long63": dict(seed=4, n_reads=120_000, read_len=10_000, sub_rate_ppm=10_000, k=63, m_in=5_000_000_011),
}


def sha(a):
    h = hashlib.sha256()
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    step = 1 << 28
    for i in range(0, a.size, step):
        h.update(a[i:i + step].tobytes())
    return h.hexdigest()


def run(name):
    c = CASES[name]
    ref = RefLib()
    L = ref.lib
    k, n_reads = c["k"], c["n_reads"]
    b, o = synth.reads(seed=c["seed"], total_reads=n_reads, read_len=c["read_len"], sub_rate_ppm=c["sub_rate_ppm"])
    h = L.p3ref_open(b, o, n_reads, k)
    cap = int(b.size)
    keys, counts = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    n = C.c_uint64()
    t = time.time()
    assert L.p3ref_count_short(h, keys, counts, cap, C.byref(n)) == 0
    t_count = time.time() - t
    keys, counts = keys[:n.value], counts[:n.value]
    out = dict(c, n_distinct=int(n.value), n_solid=int((counts >= 2).sum()), n_windows=int(counts.sum()),
               max_count=int(counts.max()), keys_sha256=sha(keys), counts_sha256=sha(counts), t_count_s=t_count)
    print(name, "count done", t_count, out["n_distinct"], flush=True)
    del keys, counts
    m, H = ref.estimate_bloom(int(b.size), k, c["m_in"])
    words = np.zeros((m + 63) // 64, np.uint64)
    seed_chars = np.zeros(n_reads * k, np.uint8)
    ns = C.c_uint64()
    t = time.time()
    assert L.p3ref_make_bf(h, k, m, H, words, seed_chars, n_reads, C.byref(ns)) == 0
    seeds = seed_chars[:ns.value * k]
    bits = 0
    for i in range(0, words.size, 1 << 24):
        bits += int(np.unpackbits(words[i:i + (1 << 24)].view(np.uint8)).sum())
    out.update(m=int(m), H=int(H), bloom_sha256=sha(words), bits_set=bits, n_seeds=int(ns.value),
               seeds_sha256=sha(seeds), t_makebf_s=time.time() - t)
    L.p3ref_close(h)
    with open(os.path.join(OUTDIR, "big_%s.json" % name), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1), flush=True)


if __name__ == "__main__":
    for nm in sys.argv[1:]:
        run(nm)
