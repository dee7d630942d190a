"""ctypes loaders for the CHECKERS -- test infrastructure only.

  Oracle  : our CPU restatement (oracle/libp3oracle.so, built by oracle/Makefile)
  RefLib  : the real reference compiled from /root/reference (oracle/_ref/libp3ref.so), if present

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (platanus3_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


MAXW = 96  # words of the widest key (k = 3001 takes 94)


def nwords(k):
    return (2 * k + 63) // 64


def build(force=False):
    """Compile libp3oracle.so (always) and _ref/libp3ref.so (only where /root/reference exists)."""
    so = os.path.join(_HERE, "libp3oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("p3_oracle.cpp", "p3_oracle_dbg.cpp", "p3_oracle_stream.cpp", "p3_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libp3oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src"):
        ref = os.path.join(_HERE, "_ref", "libp3ref.so")
        hs = os.path.join(_HERE, "ref_harness.cpp")
        if force or not os.path.exists(ref) or os.path.getmtime(ref) < os.path.getmtime(hs):
            subprocess.check_call(["make", "-C", _HERE, "_ref/libp3ref.so"], stdout=subprocess.DEVNULL)


def _flat(reads):
    """list[bytes|str] or (bases,offsets) -> (uint8 bases, uint64 offsets)"""
    if isinstance(reads, tuple):
        b, o = reads
        return np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(o, dtype=np.uint64)
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if bases.size == 0:
        bases = np.zeros(1, np.uint8)
    return bases, offs


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "libp3oracle.so"))
        L = self.lib
        L.p3o_std_hash.restype = C.c_uint64
        L.p3o_std_hash.argtypes = [_u64p, C.c_int]
        L.p3o_double_hash.argtypes = [_u64p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.p3o_probe_positions.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _u64p]
        L.p3o_rmq.restype = C.c_uint64
        L.p3o_rmq.argtypes = [_u64p, C.c_uint64, C.c_int, _u64p]
        L.p3o_estimate_bloom.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.p3o_count_short.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint32, _u64p, _u64p, C.c_uint64,
                                      C.POINTER(C.c_uint64)]
        L.p3o_list_canon21.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint32, _u64p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.p3o_make_bf.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint32, _u64p, _u64p, C.c_uint64, C.c_int,
                                  C.c_uint64, C.c_int, _u64p, _u8p, C.c_uint64, C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.p3o_gate_flags.restype = C.c_uint64
        L.p3o_gate_flags.argtypes = [_u8p, C.c_uint64, _u64p, _u64p, C.c_uint64, C.c_int, _u8p]
        L.p3o_bloom_query.argtypes = [_u64p, C.c_uint64, C.c_int, _u64p, C.c_uint64, C.c_int, _u8p]
        L.p3o_refshaped_pass.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64, C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                         C.c_void_p, C.POINTER(C.c_uint64)]
        L.p3o_dbg.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint32, C.c_int, _u64p, C.c_uint64, C.c_int, _u64p, C.c_uint64,
                              C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.p3o_dbg_joints.argtypes = [_u8p, _u64p, C.c_uint64, C.c_uint32, C.c_int, _u64p, C.c_uint64, C.c_int, _u64p, C.c_uint64,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_uint64)]
        L.p3o_free.argtypes = [C.c_void_p]
        for f in ("p3o_encode_fwd", "p3o_encode_rc"):
            getattr(L, f).argtypes = [C.c_char_p, C.c_int, _u64p]
        L.p3o_complement.argtypes = [_u64p, C.c_int, _u64p]
        L.p3o_canonical.argtypes = [_u64p, _u64p, C.c_int, _u64p]
        L.p3o_decode.argtypes = [_u64p, C.c_int, C.c_char_p]

    # ---- primitives -------------------------------------------------------------------------
    def encode_fwd(self, s, k=None):
        s = s.encode() if isinstance(s, str) else s
        k = k or len(s)
        out = np.zeros(MAXW, np.uint64)
        self.lib.p3o_encode_fwd(s, k, out)
        return out[: nwords(k)].copy()

    def encode_rc(self, s, k=None):
        s = s.encode() if isinstance(s, str) else s
        k = k or len(s)
        out = np.zeros(MAXW, np.uint64)
        self.lib.p3o_encode_rc(s, k, out)
        return out[: nwords(k)].copy()

    def _pad(self, key):
        w = np.zeros(MAXW, np.uint64)
        key = np.atleast_1d(np.asarray(key, dtype=np.uint64))
        w[: key.size] = key
        return w

    def complement(self, key, k):
        out = np.zeros(MAXW, np.uint64)
        self.lib.p3o_complement(self._pad(key), k, out)
        return out[: nwords(k)].copy()

    def canonical(self, f, r, k):
        out = np.zeros(MAXW, np.uint64)
        self.lib.p3o_canonical(self._pad(f), self._pad(r), k, out)
        return out[: nwords(k)].copy()

    def decode(self, key, k):
        buf = C.create_string_buffer(k)
        self.lib.p3o_decode(self._pad(key), k, buf)
        return buf.raw[:k].decode()

    def std_hash(self, key, k):
        return int(self.lib.p3o_std_hash(self._pad(key), k))

    def double_hash(self, key, k):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.p3o_double_hash(self._pad(key), k, C.byref(a), C.byref(b))
        return a.value, b.value

    def probe_positions(self, h1, h2, m, H):
        out = np.zeros(max(H, 1), np.uint64)
        self.lib.p3o_probe_positions(h1, h2, m, H, out)
        return out[:H].copy()

    def rmq(self, v, x):
        v = np.ascontiguousarray(v, dtype=np.uint64)
        out = np.zeros(max(v.size, 1), np.uint64)
        n = self.lib.p3o_rmq(v if v.size else np.zeros(1, np.uint64), v.size, x, out)
        return out[:n].copy()

    def estimate_bloom(self, all_bases, k, m_in=0):
        m, H = C.c_uint64(), C.c_int()
        self.lib.p3o_estimate_bloom(all_bases, k, m_in, C.byref(m), C.byref(H))
        return m.value, H.value

    # ---- stages -----------------------------------------------------------------------------
    def count_short(self, reads, min_len=21):
        bases, offs = _flat(reads)
        n_reads = offs.size - 1
        cap = max(1, int(sum(max(0, int(offs[i + 1] - offs[i]) - 20) for i in range(n_reads)))
                  if n_reads < 100000 else int(bases.size))
        keys = np.zeros(cap, np.uint64)
        counts = np.zeros(cap, np.uint64)
        n = C.c_uint64()
        rc = self.lib.p3o_count_short(bases, offs, n_reads, min_len, keys, counts, cap, C.byref(n))
        assert rc == 0, rc
        return keys[: n.value].copy(), counts[: n.value].copy()

    def list_canon21(self, reads, min_len=21):
        bases, offs = _flat(reads)
        cap = max(1, int(bases.size))
        out = np.zeros(cap, np.uint64)
        n = C.c_uint64()
        rc = self.lib.p3o_list_canon21(bases, offs, offs.size - 1, min_len, out, cap, C.byref(n))
        assert rc == 0, rc
        return out[: n.value].copy()

    def make_bf(self, reads, keys, counts, k, m, H, min_len=None):
        bases, offs = _flat(reads)
        n_reads = offs.size - 1
        min_len = k if min_len is None else min_len
        words = np.zeros((m + 63) // 64, np.uint64)
        seed_cap = max(1, n_reads)
        seed_chars = np.zeros(seed_cap * k, np.uint8)
        ns, nins, ovf = C.c_uint64(), C.c_uint64(), C.c_int()
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        kk = keys if keys.size else np.zeros(1, np.uint64)
        cc = counts if counts.size else np.zeros(1, np.uint64)
        rc = self.lib.p3o_make_bf(bases, offs, n_reads, min_len, kk, cc, keys.size, k, m, H, words,
                                  seed_chars, seed_cap, C.byref(ns), C.byref(nins), C.byref(ovf))
        assert rc == 0, rc
        seeds = [seed_chars[i * k:(i + 1) * k].tobytes().decode() for i in range(ns.value)]
        return words, seeds, nins.value, ovf.value

    def gate_flags(self, read, keys, counts, k):
        r = np.frombuffer(read.encode() if isinstance(read, str) else read, dtype=np.uint8).copy()
        flags = np.zeros(max(1, r.size), np.uint8)
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        n = self.lib.p3o_gate_flags(r, r.size, keys, counts, keys.size, k, flags)
        return flags[:n].copy()

    def bloom_query(self, words, m, H, keys, k):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
        n = keys.size // nwords(k)
        out = np.zeros(max(1, n), np.uint8)
        self.lib.p3o_bloom_query(np.ascontiguousarray(words, dtype=np.uint64), m, H,
                                 keys if keys.size else np.zeros(1, np.uint64), n, k, out)
        return out[:n].copy()

    def dbg_gfa(self, reads, bloom_words, m, H, seeds, k, min_len=None):
        """G4-G6: GFA text of the de Bruijn stage from (reads, Bloom words, seed strings)."""
        bases, offs = _flat(reads)
        min_len = k if min_len is None else min_len
        nw = nwords(k)
        sk = np.zeros(max(1, len(seeds)) * nw, np.uint64)
        for i, sd in enumerate(seeds):
            sk[i * nw:(i + 1) * nw] = self.encode_fwd(sd)
        txt, ln = C.c_void_p(), C.c_uint64()
        nj, njo, ns = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = self.lib.p3o_dbg(bases, offs, offs.size - 1, min_len, k, np.ascontiguousarray(bloom_words, np.uint64), m, H,
                              sk, len(seeds), C.byref(txt), C.byref(ln), C.byref(nj), C.byref(njo), C.byref(ns))
        assert rc == 0, rc
        s = C.string_at(txt.value, ln.value).decode()
        self.lib.p3o_free(txt)
        return s, dict(junctions=nj.value, joints=njo.value, straights=ns.value)

    def dbg_joints(self, reads, bloom_words, m, H, seeds, k, min_len=None):
        """G5: [(joint k-mer in its smaller orientation, JointInfo::coverage)] sorted (src/DeBruijnGraph.cpp:441-449)"""
        bases, offs = _flat(reads)
        min_len = k if min_len is None else min_len
        nw = nwords(k)
        sk = np.zeros(max(1, len(seeds)) * nw, np.uint64)
        for i, sd in enumerate(seeds):
            sk[i * nw:(i + 1) * nw] = self.encode_fwd(sd)
        txt, ln, jt, jl = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        rc = self.lib.p3o_dbg_joints(bases, offs, offs.size - 1, min_len, k, np.ascontiguousarray(bloom_words, np.uint64), m, H,
                                     sk, len(seeds), C.byref(txt), C.byref(ln), C.byref(jt), C.byref(jl))
        assert rc == 0, rc
        s = C.string_at(jt.value, jl.value).decode()
        self.lib.p3o_free(txt)
        self.lib.p3o_free(jt)
        return [(a, int(b)) for a, b in (line.split("\t") for line in s.splitlines())]

    def refshaped_pass(self, reads, k, m, H, min_len=None, want_bloom=False):
        bases, offs = _flat(reads)
        min_len = k if min_len is None else min_len
        tc, tb, nd, ns = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
        words = np.zeros((m + 63) // 64, np.uint64) if want_bloom else None
        rc = self.lib.p3o_refshaped_pass(bases, offs, offs.size - 1, min_len, k, m, H, C.byref(tc), C.byref(tb),
                                         C.byref(nd), words.ctypes.data if want_bloom else None, C.byref(ns))
        assert rc == 0, rc
        return dict(t_count=tc.value, t_bloom=tb.value, n_distinct=nd.value, n_seeds=ns.value, bloom=words)


    STREAM_STATS = ("n_windows", "n_distinct", "n_solid", "max_count", "digest_sum", "digest_xor", "n_windows_k", "n_passing",
                    "n_seed_reads", "n_seeds", "count_overflow")

    def stream_pass(self, reads, k, m, H, min_len=None, part_bits=4, want_seeds=True):
        """count -> gate -> Bloom -> seeds as streaming passes (inputs of the BASELINE configurations' full size):
        statistics incl. an order-independent digest of the (key, count) table, the Bloom words, the seed keys."""
        bases, offs = _flat(reads)
        min_len = k if min_len is None else min_len
        st = np.zeros(16, np.uint64)
        words = np.zeros((m + 63) // 64, np.uint64)
        n = offs.size - 1
        nw = nwords(k)
        seeds = np.zeros((max(1, n) if want_seeds else 1, nw), np.uint64)
        self.lib.p3o_stream_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64, C.c_int, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        rc = self.lib.p3o_stream_pass(bases.ctypes.data, offs.ctypes.data, n, min_len, k, m, H, part_bits, st.ctypes.data,
                                      words.ctypes.data, seeds.ctypes.data if want_seeds else None, seeds.shape[0])
        assert rc == 0, rc
        out = {name: int(st[i]) for i, name in enumerate(self.STREAM_STATS)}
        out["bloom"] = words
        out["seeds"] = seeds[: out["n_seeds"]] if want_seeds else None
        return out

    @staticmethod
    def table_digest(keys, counts):
        """the digest stream_pass reports, from an exported (key, count) table (any order)"""
        k = np.asarray(keys, np.uint64)
        c = np.asarray(counts, np.uint64)
        with np.errstate(over="ignore"):
            x = k ^ (c * np.uint64(0x9E3779B97F4A7C15))
            x = x + np.uint64(0x9E3779B97F4A7C15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
            return int(x.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(x)) if x.size else 0


class RefLib:
    """The real reference (compiled from /root/reference).  available() is False where the .so is absent."""

    PATH = os.path.join(_HERE, "_ref", "libp3ref.so")

    @classmethod
    def available(cls):
        if os.path.isdir("/root/reference/src"):
            try:
                build()
            except Exception:
                pass
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = C.CDLL(self.PATH)
        L = self.lib
        L.p3ref_open.restype = C.c_void_p
        L.p3ref_open.argtypes = [_u8p, _u64p, C.c_uint64, C.c_int]
        L.p3ref_close.argtypes = [C.c_void_p]
        L.p3ref_all_bases.restype = C.c_uint64
        L.p3ref_all_bases.argtypes = [C.c_void_p]
        L.p3ref_n_reads.restype = C.c_uint64
        L.p3ref_n_reads.argtypes = [C.c_void_p]
        L.p3ref_estimate_bloom.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.p3ref_count_short.argtypes = [C.c_void_p, _u64p, _u64p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.p3ref_make_bf.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, _u64p, _u8p, C.c_uint64,
                                    C.POINTER(C.c_uint64)]
        L.p3ref_last_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.p3ref_assemble.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_char_p]
        L.p3ref_joints.restype = C.c_uint64
        L.p3ref_joints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.p3ref_kat.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_int, _u64p, _u64p, _u64p, _u64p,
                                C.POINTER(C.c_uint64), _u64p, _u64p]
        L.p3ref_hash_words.argtypes = [_u64p, C.c_uint64, C.c_int, _u64p, _u64p, _u64p]
        L.p3ref_bloom_query.argtypes = [_u64p, C.c_uint64, C.c_int, _u64p, C.c_uint64, C.c_int, _u8p]
        L.p3ref_encode21.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_char_p]
        L.p3ref_open_file.restype = C.c_void_p
        L.p3ref_open_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.p3ref_dump_reads.restype = C.c_uint64
        L.p3ref_dump_reads.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.p3ref_run_loaded.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_char_p, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_int)]
        L.p3ref_rmq.restype = C.c_uint64
        L.p3ref_rmq.argtypes = [_u64p, C.c_uint64, C.c_int, _u64p]

    def load_file(self, path, k):
        """The real ReadFile(options) + LoadFile() on `path` -> dict(all_bases, suffix_ok, entries=[(name, seq) sorted])."""
        ok = C.c_int()
        h = self.lib.p3ref_open_file(path.encode(), k, C.byref(ok))
        try:
            need = self.lib.p3ref_dump_reads(h, None, 0)
            buf = C.create_string_buffer(max(1, need))
            self.lib.p3ref_dump_reads(h, buf, need)
            lines = buf.raw[:need].split(b"\n")[:-1] if need else []
            entries = [(lines[i], lines[i + 1]) for i in range(0, len(lines), 2)]
            return dict(all_bases=int(self.lib.p3ref_all_bases(h)), suffix_ok=int(ok.value), entries=entries)
        finally:
            self.lib.p3ref_close(h)

    def run_file(self, path, k, workdir, m_in=0, threads=1):
        """main()'s flow on a file (load, EstimateBloomfilter, Assemble<B>) -> (gfa text, m, H)."""
        ok = C.c_int()
        h = self.lib.p3ref_open_file(path.encode(), k, C.byref(ok))
        try:
            m, H = C.c_uint64(), C.c_int()
            rc = self.lib.p3ref_run_loaded(h, k, m_in, threads, workdir.encode(), C.byref(m), C.byref(H))
            assert rc == 0, rc
            return open(os.path.join(workdir, "de_bruijn_graph.gfa")).read(), m.value, H.value
        finally:
            self.lib.p3ref_close(h)

    def estimate_bloom(self, all_bases, k, m_in=0):
        m, H = C.c_uint64(), C.c_int()
        self.lib.p3ref_estimate_bloom(all_bases, k, m_in, C.byref(m), C.byref(H))
        return m.value, H.value

    def kat(self, seq, m, H):
        k = len(seq)
        nw = nwords(k)
        f, r, c, cn = (np.zeros(MAXW, np.uint64) for _ in range(4))
        sh = C.c_uint64()
        h12 = np.zeros(2, np.uint64)
        pos = np.zeros(max(H, 1), np.uint64)
        rc = self.lib.p3ref_kat(seq.encode(), k, m, H, f, r, c, cn, C.byref(sh), h12, pos)
        assert rc == 0, rc
        return dict(fwd=f[:nw].copy(), rc=r[:nw].copy(), comp=c[:nw].copy(), canon=cn[:nw].copy(),
                    std_hash=sh.value, h1=int(h12[0]), h2=int(h12[1]), pos=pos[:H].copy())

    def hash_words(self, words, k):
        words = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
        n = words.size // nwords(k)
        sh, h1, h2 = (np.zeros(max(n, 1), np.uint64) for _ in range(3))
        rc = self.lib.p3ref_hash_words(words, n, k, sh, h1, h2)
        assert rc == 0, rc
        return sh[:n], h1[:n], h2[:n]

    def encode21(self, s):
        f, r = C.c_uint64(), C.c_uint64()
        a, b = C.create_string_buffer(21), C.create_string_buffer(21)
        self.lib.p3ref_encode21(s.encode(), C.byref(f), C.byref(r), a, b)
        return f.value, r.value, a.raw[:21].decode(), b.raw[:21].decode()

    def rmq(self, v, x):
        v = np.ascontiguousarray(v, dtype=np.uint64)
        out = np.zeros(max(v.size, 1), np.uint64)
        n = self.lib.p3ref_rmq(v, v.size, x, out)
        return out[:n].copy()

    def bloom_query(self, words, m, H, keys, k):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
        n = keys.size // nwords(k)
        out = np.zeros(max(1, n), np.uint8)
        self.lib.p3ref_bloom_query(np.ascontiguousarray(words, dtype=np.uint64), m, H, keys, n, k, out)
        return out[:n].copy()

    def run(self, reads, k, m, H, want_bloom=True):
        """CountShortKmer + MakeBF on the real reference. Returns dict(keys,counts,bloom,seeds,t_count,t_bloom)."""
        bases, offs = _flat(reads)
        n_reads = offs.size - 1
        h = self.lib.p3ref_open(bases, offs, n_reads, k)
        try:
            cap = max(1, int(bases.size))
            keys = np.zeros(cap, np.uint64)
            counts = np.zeros(cap, np.uint64)
            n = C.c_uint64()
            rc = self.lib.p3ref_count_short(h, keys, counts, cap, C.byref(n))
            assert rc == 0, rc
            out = dict(keys=keys[: n.value].copy(), counts=counts[: n.value].copy(),
                       all_bases=self.lib.p3ref_all_bases(h), n_reads=self.lib.p3ref_n_reads(h))
            if want_bloom:
                words = np.zeros((m + 63) // 64, np.uint64)
                seed_cap = max(1, n_reads)
                seed_chars = np.zeros(seed_cap * k, np.uint8)
                ns = C.c_uint64()
                rc = self.lib.p3ref_make_bf(h, k, m, H, words, seed_chars, seed_cap, C.byref(ns))
                assert rc == 0, rc
                out["bloom"] = words
                out["seeds"] = [seed_chars[i * k:(i + 1) * k].tobytes().decode() for i in range(ns.value)]
            tc, tb = C.c_double(), C.c_double()
            self.lib.p3ref_last_times(h, C.byref(tc), C.byref(tb))
            out["t_count"], out["t_bloom"] = tc.value, tb.value
            return out
        finally:
            self.lib.p3ref_close(h)

    def assemble_gfa(self, reads, k, m, H, threads, workdir, joints=False):
        """GFA text of the real Assemble<B>; joints=True: also [(joint k-mer in its smaller orientation,
        JointInfo::coverage)] sorted -- a field the reference fills (src/DeBruijnGraph.cpp:441-449) but never prints"""
        bases, offs = _flat(reads)
        h = self.lib.p3ref_open(bases, offs, offs.size - 1, k)
        jl = None
        try:
            rc = self.lib.p3ref_assemble(h, k, m, H, threads, workdir.encode())
            assert rc == 0, rc
            if joints:
                n = self.lib.p3ref_joints(h, k, None, None, 0)
                km = np.zeros(max(1, n * k), np.uint8)
                cv = np.zeros(max(1, n), np.int32)
                self.lib.p3ref_joints(h, k, km.ctypes.data, cv.ctypes.data, n)
                jl = [(km[i * k:(i + 1) * k].tobytes().decode(), int(cv[i])) for i in range(n)]
        finally:
            self.lib.p3ref_close(h)
        with open(os.path.join(workdir, "de_bruijn_graph.gfa")) as f:
            gfa = f.read()
        return (gfa, jl) if joints else gfa
