"""Full-size inputs against checksums of the REAL reference (tests/golden/make_golden_big.py, build container):
the sizes at which the device path leaves its one-partition form -- second-level partitions, position chunks,
collected k-mer groups, 2-word keys, a Bloom vector of more than 2^32 bits -- so that machinery is pinned against
the reference itself (src/Load.cpp:105-127, src/MakeBloomFilter.cpp:24-89) and not against another device path."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import platanus3_amd.synth as synth
from platanus3_amd import capi
from util import GOLDEN

pytestmark = pytest.mark.gpu


def _sha(a):
    h = hashlib.sha256()
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    for i in range(0, a.size, 1 << 28):
        h.update(a[i:i + (1 << 28)].tobytes())
    return h.hexdigest()


def _cases():
    return sorted(os.path.basename(p)[4:-5] for p in glob.glob(os.path.join(GOLDEN, "big_*.json")))


def _seed_chars(seeds, k):
    """(n, nw) forward keys [lo, hi] -> (n, k) ASCII, first base first"""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    chars = np.empty((seeds.shape[0], k), np.uint8)
    for i in range(k):
        bit = 2 * (k - 1 - i)
        w = seeds[:, bit // 64]
        chars[:, i] = lut[((w >> np.uint64(bit % 64)) & np.uint64(3)).astype(np.int64)]
    return chars


@pytest.mark.parametrize("name", _cases())
def test_big_input_against_reference_checksums(name):
    with open(os.path.join(GOLDEN, "big_%s.json" % name)) as f:
        g = json.load(f)
    k = g["k"]
    b, o = synth.reads(seed=g["seed"], total_reads=g["n_reads"], read_len=g["read_len"], sub_rate_ppm=g["sub_rate_ppm"])
    ctx = capi.Context(0)
    try:
        ctx.load_reads(b, o, min_len=k)
        del b, o
        ctx.run_pass(k, g["m"], g["H"])
        st = ctx.count_stats()
        assert st["n_distinct"] == g["n_distinct"] and st["n_solid"] == g["n_solid"]
        assert st["n_windows"] == g["n_windows"] and st["max_count"] == g["max_count"]
        bs = ctx.bloom_stats()
        assert bs["n_seed_reads"] >= g["n_seeds"]
        words = ctx.export_bloom()
        assert _sha(words) == g["bloom_sha256"]
        del words
        seeds = ctx.export_seeds()
        assert seeds.shape[0] == g["n_seeds"]
        assert _sha(_seed_chars(seeds, k)) == g["seeds_sha256"]
        del seeds
        keys, counts = ctx.export_counts()
        assert int(counts.sum()) == g["n_windows"]
        assert _sha(keys) == g["keys_sha256"] and _sha(counts) == g["counts_sha256"]
        info = {n: ctx.get_info(n) for n in ("chunks", "split_xbits", "ksplit_xbits", "kmer_groups", "est_level")}
        print(name, info)
    finally:
        ctx.close()


def _full_cases():
    return sorted(os.path.basename(p)[5:-5] for p in glob.glob(os.path.join(GOLDEN, "full_*.json")))


@pytest.mark.parametrize("name", _full_cases())
def test_baseline_configuration_at_full_size(name):
    """BASELINE configurations #3 (50 M x 150 bp, k=31) and #4 (1 M x 10 kbp, k=63, m > 2^32, H = 19) at their STATED
    size on one GPU -- position chunks, collected k-mer groups, several probe sweeps, the reads grouped on the device first
    as bench.py's `scale_base` does -- against what the oracle's streaming form computed for the same reads
    (tests/golden/make_golden_full.py; that form is pinned against the oracle proper by tests/test_oracle_stream.py):
    statistics, the order-independent digest of the whole count table, sha256 of the Bloom words and of the seed
    strings.  Careful first pass, then the replay."""
    with open(os.path.join(GOLDEN, "full_%s.json" % name)) as f:
        g = json.load(f)
    k = g["k"]
    b, o = synth.reads(seed=g["seed"], total_reads=g["n_reads"], read_len=g["read_len"], sub_rate_ppm=g["sub_rate_ppm"])
    ctx = capi.Context(0)
    try:
        ctx.load_reads(b, o, min_len=k)
        del b, o
        ctx.group_reads(0)  # (no result depends on the order of the reads)
        for _ in range(2):
            ctx.run_pass(k, g["m"], g["H"])
            st = ctx.count_stats()
            assert st["n_distinct"] == g["n_distinct"] and st["n_solid"] == g["n_solid"]
            assert st["n_windows"] == g["n_windows"] and st["max_count"] == g["max_count"]
            assert ctx.count_digest() == (g["digest_sum"], g["digest_xor"])
            bs = ctx.bloom_stats()
            assert bs["n_windows_k"] == g["n_windows_k"] and bs["n_passing"] == g["n_passing"]
            assert bs["n_seed_reads"] == g["n_seed_reads"]
        words = ctx.export_bloom()
        assert _sha(words) == g["bloom_sha256"]
        lut = np.array([bin(i).count("1") for i in range(256)], np.uint64)
        assert int(lut[words.view(np.uint8)].sum()) == g["bits_set"]
        del words
        seeds = ctx.export_seeds()
        assert seeds.shape[0] == g["n_seeds"]
        assert _sha(_seed_chars(seeds, k)) == g["seeds_sha256"]
    finally:
        ctx.close()
