"""Fixtures for the BASELINE configurations at their FULL size (run here, in the build container; ~40 min and ~25 GB each).

The reference itself cannot run them in this container (config #3's unordered_map of ~1 G entries, built twice, needs
> 100 GB), so these come from the oracle's streaming form (oracle/p3_oracle_stream.cpp), which
tests/test_oracle_stream.py pins against the oracle proper -- itself pinned on the real reference's outputs.  Data only:
statistics, an order-independent digest of the count table, sha256 of the Bloom words and of the seed strings.

usage: python tests/golden/make_golden_full.py 3|4 [reads]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
import platanus3_amd.synth as synth
from oracle.p3_oracle import Oracle

CONFIGS = {3: dict(reads=50_000_000, read_len=150, k=31, sub_ppm=5000, part_bits=5),
           4: dict(reads=1_000_000, read_len=10_000, k=63, sub_ppm=10_000, part_bits=6)}


def main():
    cfg = int(sys.argv[1])
    c = dict(CONFIGS[cfg])
    if len(sys.argv) > 2:
        c["reads"] = int(sys.argv[2])
    t0 = time.time()
    b, o = synth.reads(seed=cfg, total_reads=c["reads"], read_len=c["read_len"], sub_rate_ppm=c["sub_ppm"])
    orc = Oracle()
    k = c["k"]
    m, H = orc.estimate_bloom(int(b.size), k)
    print("config #%d: %d reads x %d, k=%d, m=%d, H=%d (generated in %.0f s)" % (cfg, c["reads"], c["read_len"], k, m, H, time.time() - t0), flush=True)
    s = orc.stream_pass((b, o), k, m, H, part_bits=c["part_bits"])
    words, seeds = s.pop("bloom"), s.pop("seeds")
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint64)
    bits = int(lut[words.view(np.uint8)].sum())
    nw = seeds.shape[1]
    chars = np.empty((seeds.shape[0], k), np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(k):
        bit = 2 * (k - 1 - i)
        chars[:, i] = acgt[((seeds[:, bit >> 6] >> np.uint64(bit & 63)) & np.uint64(3)).astype(np.int64)]
    out = dict(config=cfg, seed=cfg, n_reads=c["reads"], read_len=c["read_len"], sub_rate_ppm=c["sub_ppm"], k=k, m=m, H=H, **s,
               bits_set=bits, bloom_sha256=hashlib.sha256(words.tobytes()).hexdigest(),
               seeds_sha256=hashlib.sha256(np.ascontiguousarray(chars).tobytes()).hexdigest(),
               produced_by="oracle/p3_oracle_stream.cpp (streaming form of the oracle), %.0f s" % (time.time() - t0))
    name = "full_config%d.json" % cfg if c["reads"] == CONFIGS[cfg]["reads"] else "full_config%d_%d.json" % (cfg, c["reads"])
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
