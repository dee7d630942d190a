"""Loader / option fixtures for tests/test_host_io.py.  Writes the tricky input files under tests/golden/io/ (they
are data authored here) and records what the REAL reference's ReadFile(options) + LoadFile() (src/Load.cpp:23-103)
and Options::EstimateBloomfilter (src/Options.cpp:50-60) make of them -> tests/golden/io/expected.json.
Run in the build container (needs oracle/_ref/libp3ref.so):  python tests/golden/make_golden_io.py"""
import base64, json, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.p3_oracle import RefLib  # noqa: E402

IO = os.path.join(HERE, "io")
A = "ACGTTGCATGCCGATAGCTAGGCTAACGTTAGC"  # 33 bases
B = "TTGACCAGTAGGCATCGATCGGATCGATTTACGATCGACTAGCTAGCATG"  # 50 bases
FILES = {
    "simple.fasta": ">r0\n%s\n>r1\n%s\n" % (A, B),
    "multiline_dups.fasta": ">r0 first\n%s\n%s\n>short\nACGT\n>r0 first\n%s\n\n>tail no newline\n%s%s" % (A[:20], A[20:], B, B, A),
    "crlf.fasta": ">r0\r\n%s\r\n>r1\r\n%s\r\n%s\r\n" % (A, B[:25], B[25:]),
    "lower_n.fasta": ">mixed\n%sNNnnacgt%s\n>x\n%s\n" % (A, B, A.lower()),
    "simple.fastq": "@q0\n%s\n+\n%s\n@q1 desc\n%s\n+q1\n%s\n" % (A, "I" * len(A), B, "#" * len(B)),
    "tricky.fastq": "@q0\n%s\n+\n@%s\n@q0\n%s\n+\n%s\n\n%s\n+\n%s\n@last\n%s" % (A, "I" * (len(A) - 1), B, "I" * len(B), A, "I" * len(A), B),
    "fasta_named.fastq": ">r0\n%s\n>r1\n%s\n" % (A, B),
    "fastq_named.fasta": "@q0\n%s\n+\n%s\n" % (B, "I" * len(B)),
    "empty.fasta": "",
    "garbage.fasta": "hello\n>r0\n%s\n" % A,
    "reads.fa": ">r0\n%s\n" % B,
    "blank_lines.fasta": ">r0\n\n%s\n\n\n>r1\n%s\n\n" % (A, B),
    "only_header.fasta": ">r0\n",
}


def main():
    os.makedirs(IO, exist_ok=True)
    ref = RefLib()
    out = {"loads": [], "estimates": []}
    for name, text in FILES.items():
        path = os.path.join(IO, name)
        with open(path, "wb") as f:
            f.write(text.encode())
        for k in (21, 31, 40):
            r = ref.load_file(path, k)
            out["loads"].append(dict(file=name, k=k, all_bases=r["all_bases"], suffix_ok=r["suffix_ok"],
                                     entries=[[base64.b64encode(n).decode(), base64.b64encode(s).decode()] for n, s in r["entries"]]))
    for bases in (60000, 1500000, 150000000, 7500000000, 10000000000, 12345678):
        for k in (21, 25, 31, 41, 51, 61, 63):
            for m_in in (0, 1 << 30):
                m, H = ref.estimate_bloom(bases, k, m_in)
                out["estimates"].append(dict(all_bases=bases, k=k, m_in=m_in, m=m, H=H if m_in == 0 else 10))
    with open(os.path.join(IO, "expected.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", len(out["loads"]), "load cases,", len(out["estimates"]), "estimates")


if __name__ == "__main__":
    main()
