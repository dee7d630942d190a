"""ID-free canonical form of a platanus3 GFA.

Nodes: (kind, min(seq, revcomp(seq)), KC).  Edges: rewritten in terms of canonical node sequences, the orientation
flag flipped when the node's sequence was reverse-complemented, each edge normalised to the smaller of itself and
its reverse-complement reading.  Both are MULTISETS (the reference emits junction-junction links from both ends)."""
import collections

_COMP = str.maketrans("ACGT", "TGCA")


def revcomp(s):
    return s.translate(_COMP)[::-1]


def canonical(gfa_text):
    nodes = collections.Counter()
    byname = {}
    edges = collections.Counter()
    lines = [l for l in gfa_text.split("\n") if l]
    for l in lines:
        f = l.split("\t")
        if f[0] == "S":
            name, seq, kc = f[1], f[2], f[3]
            rc = revcomp(seq)
            cs, flipped = (seq, False) if seq <= rc else (rc, True)
            kind = name.split("_")[0]
            byname[name] = (kind, cs, flipped)
            nodes[(kind, cs, kc)] += 1
    for l in lines:
        f = l.split("\t")
        if f[0] == "L":
            ka, sa, fa = byname[f[1]]
            kb, sb, fb = byname[f[3]]
            oa = (f[2] == "+") != fa
            ob = (f[4] == "+") != fb
            e = ((ka, sa, oa), (kb, sb, ob), f[5])
            er = ((kb, sb, not ob), (ka, sa, not oa), f[5])
            edges[min(e, er)] += 1
    return nodes, edges
