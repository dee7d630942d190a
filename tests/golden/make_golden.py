#!/usr/bin/env python3
"""Generates tests/golden/*.npz|json from the REAL reference (oracle/_ref/libp3ref.so).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The fixtures are DATA: inputs (read strings) and the reference's outputs on them -- sorted
(canonical 21-mer, count) table, Bloom words, seed strings, primitive known answers.  No reference
source text is stored.  The reference ships no tests/fixtures of its own.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.p3_oracle import RefLib, nwords  # noqa: E402
import platanus3_amd.synth as synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KAT_SEQ = "GATTACAGATTACACATTAGGCCTAGCATCGATCGGGATATCCGCGTTAAACCGGTTTGACCACTGA"


def reads_to_list(bases, offs):
    return [bases[int(offs[i]):int(offs[i + 1])].tobytes().decode() for i in range(offs.size - 1)]


def ragged_reads(rng, n, genome_len=3000):
    g = rng.choice(list("ACGT"), size=genome_len)
    out = []
    for _ in range(n):
        L = int(rng.integers(15, 320))
        s = int(rng.integers(0, genome_len - L))
        r = g[s:s + L].copy()
        if rng.random() < 0.5:
            comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
            r = np.array([comp[c] for c in r[::-1]])
        # sprinkle unknown characters: N, lower case, IUPAC
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, L))] = rng.choice(list("NnacgtRY"))
        out.append("".join(r))
    return out


def lowcomplex_reads():
    reads = ["A" * 200, "A" * 150, "T" * 180, "AC" * 90, "GT" * 80, "ACGT" * 40, "ACGT" * 41,
             "GATC" * 45, "ACG" * 60, "CGT" * 55, "AATT" * 50, "AAAAC" * 40, "GTTTT" * 37]
    reads += ["ACGT" * 16, "ACGT" * 16]          # 64-mers equal to their own reverse complement
    reads += ["AATT" * 16 + "C", "G" + "AATT" * 16]
    return reads


def main():
    ref = RefLib()
    rng = np.random.default_rng(12345)

    # ---- primitive known answers -------------------------------------------------------------
    kat = {"seq": KAT_SEQ, "rows": []}
    for k, m, H in [(21, 452894, 19), (25, 1000003, 19), (31, 66855782, 19), (31, 3342789109, 19),
                    (31, 1 << 30, 10), (32, 12345677, 19), (33, 12345677, 19), (41, 8842216, 19),
                    (51, 10998854, 19), (61, 13155492, 19), (63, 9057880166, 19), (64, (1 << 33) + 7, 19)]:
        for start in (0, 3):
            d = ref.kat(KAT_SEQ[start:start + k], m, H)
            kat["rows"].append(dict(k=k, m=m, H=H, start=start,
                                    fwd=[int(x) for x in d["fwd"]], rc=[int(x) for x in d["rc"]],
                                    comp=[int(x) for x in d["comp"]], canon=[int(x) for x in d["canon"]],
                                    std_hash=d["std_hash"], h1=d["h1"], h2=d["h2"],
                                    pos=[int(x) for x in d["pos"]]))
    kat["hash_words"] = []
    for k in (21, 22, 24, 25, 28, 29, 31, 32, 33, 36, 37, 40, 41, 44, 45, 48, 51, 52, 56, 60, 61, 63, 64):
        nw = nwords(k)
        words = rng.integers(0, 1 << 63, size=(16, nw), dtype=np.uint64) * np.uint64(2) + \
            rng.integers(0, 2, size=(16, nw), dtype=np.uint64)
        bits_top = 2 * k - 64 * (nw - 1)
        if bits_top < 64:
            words[:, nw - 1] &= np.uint64((1 << bits_top) - 1)
        words[0, :] = 0
        sh, h1, h2 = ref.hash_words(words, k)
        kat["hash_words"].append(dict(k=k, words=[[int(x) for x in w] for w in words],
                                      std_hash=[int(x) for x in sh], h1=[int(x) for x in h1],
                                      h2=[int(x) for x in h2]))
    f, r, a, b = ref.encode21("ACGNNacgtACGTACGTACGTA")
    kat["unknown_chars"] = dict(s="ACGNNacgtACGTACGTACGTA", fwd=f, rc=r, dec_fwd=a, dec_rc=b)
    v = [5, 3, 4, 1, 7, 8, 2, 9, 9, 9]
    kat["rmq"] = [dict(v=v, x=x, out=[int(t) for t in ref.rmq(v, x)]) for x in (1, 2, 3, 5, 10)]
    kat["rmq"].append(dict(v=[(1 << 33) + 5, 7, 1 << 31], x=1,
                           out=[int(t) for t in ref.rmq([(1 << 33) + 5, 7, 1 << 31], 1)]))
    kat["estimate_bloom"] = []
    for ab, k in [(1500000, 21), (150000000, 31), (7500000000, 31), (10000000000, 63), (1500000000, 41),
                  (45000, 31), (30000, 25)]:
        m, H = ref.estimate_bloom(ab, k)
        kat["estimate_bloom"].append(dict(all_bases=ab, k=k, m=m, H=H))
    with open(os.path.join(OUT, "kat.json"), "w") as fjson:
        json.dump(kat, fjson, indent=0)

    # ---- stage fixtures ----------------------------------------------------------------------
    cases = []

    def add(name, reads, k, m=None, H=None):
        all_bases = sum(len(x) for x in reads if len(x) >= k)
        if m is None:
            m, H = ref.estimate_bloom(all_bases, k)
        out = ref.run(reads, k, m, H)
        np.savez_compressed(os.path.join(OUT, name + ".npz"),
                            reads=np.array(reads, dtype=object).astype("S"),
                            k=k, m=np.uint64(m), H=H,
                            keys=out["keys"], counts=out["counts"], bloom=out["bloom"],
                            seeds=np.array(out["seeds"], dtype="S") if out["seeds"] else np.zeros(0, "S1"))
        cases.append(dict(name=name, k=k, m=int(m), H=int(H), n_reads=len(reads), n_keys=int(out["keys"].size),
                          n_seeds=len(out["seeds"])))
        print(name, cases[-1])

    b, o = synth.reads(seed=1, total_reads=300, n_reads=300, read_len=150)
    r300 = reads_to_list(b, o)
    add("synth300_k21", r300, 21)
    add("synth300_k31", r300, 31)
    add("synth300_k31_m20_h10", r300, 31, 1 << 20, 10)
    add("synth300_k25", r300, 25)
    for k in (32, 33, 41, 51, 61, 64):
        add("synth200_k%d" % k, r300[:200], k)
    b, o = synth.reads(seed=4, total_reads=24, n_reads=24, read_len=2000, sub_rate_ppm=10000)
    add("long24_k63", reads_to_list(b, o), 63)
    rg = ragged_reads(rng, 250)
    add("ragged_k25", rg, 25)
    add("ragged_k31", rg, 31)
    add("ragged_k47", rg, 47)
    lc = lowcomplex_reads()
    add("lowcomplex_k21", lc, 21, 50021, 7)
    add("lowcomplex_k32", lc, 32, 50021, 19)
    add("lowcomplex_k64", lc, 64, 77777, 3)
    add("single_read_k21", ["ACGTTGCATGCATGCCCGATAGCTAGCTAGGATCGAT"], 21, 1009, 5)
    add("dup_read_k21", ["ACGTTGCATGCATGCCCGATAGCTAGCTAGGATCGAT"] * 2 + ["ACGT"], 21, 1009, 5)
    with open(os.path.join(OUT, "cases.json"), "w") as fjson:
        json.dump(cases, fjson, indent=1)


if __name__ == "__main__":
    main()
