"""Shared helpers for the parity tests: golden fixture loading, read sets for edge cases."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return [c["name"] for c in json.load(f)]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    reads = [bytes(r) for r in z["reads"]]
    seeds = [bytes(s).decode() for s in z["seeds"]] if z["seeds"].size else []
    return dict(reads=reads, k=int(z["k"]), m=int(z["m"]), H=int(z["H"]), keys=z["keys"], counts=z["counts"],
                bloom=z["bloom"], seeds=seeds)


def load_kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


def flat(reads):
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    joined = b"".join(bs)
    bases = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(1, np.uint8)
    return bases, offs


def random_reads(rng, n, lo=15, hi=320, genome_len=4000, unknown=True):
    g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=genome_len)
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        s = int(rng.integers(0, genome_len - L))
        r = g[s:s + L].copy()
        if rng.random() < 0.5:
            r = comp[r[::-1]]
        if rng.random() < 0.3:
            r[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8))
        if unknown and rng.random() < 0.2:
            r[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"NnacgtRY", dtype=np.uint8))
        out.append(r.tobytes())
    return out
