"""ctypes binding of include/platanus3_amd.h (libplatanus3_amd.so, built in-tree by _build.build_lib).

There is no CPU fallback: loading fails loudly when the library is missing, and every compute call
raises P3Error when no HIP device is usable.
"""
import ctypes as C
import os

import numpy as np

from . import _build

P3_OK = 0
STATUS = {0: "P3_OK", 1: "P3_EBADARG", 2: "P3_EUNSUPPORTED_K", 3: "P3_ECAPACITY", 4: "P3_EHIP", 5: "P3_ECOMM",
          6: "P3_ECOUNT_OVERFLOW", 7: "P3_ESTATE", 8: "P3_ENOMEM"}


class P3Error(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class CountStats(C.Structure):
    _fields_ = [("n_windows", C.c_uint64), ("n_distinct", C.c_uint64), ("n_solid", C.c_uint64),
                ("max_count", C.c_uint64)]


class BloomStats(C.Structure):
    _fields_ = [("n_windows_k", C.c_uint64), ("n_passing", C.c_uint64), ("n_seed_reads", C.c_uint64)]


class GraphStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_nodes", "n_junctions", "n_joints", "n_straights", "straight_bases", "n_edges",
                                           "levels")]


class Timings(C.Structure):
    _fields_ = [("ms_pack", C.c_float), ("ms_count", C.c_float), ("ms_gate", C.c_float), ("ms_bloom", C.c_float),
                ("ms_seeds", C.c_float), ("ms_total", C.c_float)]


class Traffic(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("bytes_pack", "bytes_count", "bytes_gate", "bytes_bloom", "bytes_seeds", "bytes_total")]


P3_MAX_RANKS = 64


class MrSendLayout(C.Structure):
    _fields_ = [("d_part", C.c_void_p), ("d_ghist", C.c_void_p), ("d_segoff", C.c_void_p),
                ("rec_off", C.c_uint64 * (P3_MAX_RANKS + 1)), ("rows_per_rank", C.c_uint32), ("row_u32", C.c_uint32),
                ("rec_bytes", C.c_uint32)]


_lib = None


def load(build=False):
    """Returns the loaded C-ABI library.  Raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if build:
        _build.build_lib()
    if not os.path.exists(_build.LIB):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the product path)" % _build.LIB)
    if os.environ.get("P3_NO_TORCH", "0") != "1":
        try:  # share ONE HIP runtime with torch when both live in this process
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(_build.LIB)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.p3_create.argtypes = [C.POINTER(vp), C.c_int]
    L.p3_destroy.argtypes = [vp]
    L.p3_last_error.restype = C.c_char_p
    L.p3_last_error.argtypes = [vp]
    L.p3_device_count.restype = C.c_int
    L.p3_set_profiling.argtypes = [vp, C.c_int]
    L.p3_set_profile_filter.argtypes = [vp, C.c_char_p]
    L.p3_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.p3_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.p3_load_reads.argtypes = [vp, vp, vp, u64, u32]
    L.p3_load_reads_device.argtypes = [vp, vp, vp, u64, u64, u32]
    L.p3_count_short.argtypes = [vp]
    L.p3_get_count_stats.argtypes = [vp, C.POINTER(CountStats)]
    L.p3_export_counts.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.p3_make_bloom.argtypes = [vp, u32, u64, u32]
    L.p3_export_bloom.argtypes = [vp, vp]
    L.p3_export_seeds.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.p3_get_bloom_stats.argtypes = [vp, C.POINTER(BloomStats)]
    L.p3_bloom_query.argtypes = [vp, vp, u64, vp]
    L.p3_run_pass.argtypes = [vp, u32, u64, u32]
    L.p3_build_graph.argtypes = [vp]
    L.p3_build_graph_from.argtypes = [vp, vp, C.c_uint64]
    L.p3_count_digest.argtypes = [vp, vp]
    L.p3_probe_plan.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, vp]
    L.p3_segment_capacity.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    L.p3_segment_capacity.restype = C.c_uint32
    L.p3_get_graph_stats.argtypes = [vp, C.POINTER(GraphStats)]
    L.p3_export_junctions.argtypes = [vp, vp, vp, vp, vp, u64, C.POINTER(u64)]
    L.p3_export_straights.argtypes = [vp, vp, vp, u64, u64, C.POINTER(u64)]
    L.p3_export_edges.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.p3_export_joints.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.p3_comm_unique_id.argtypes = [vp]
    L.p3_comm_create_rccl.argtypes = [C.POINTER(vp), vp, C.c_uint32, C.c_uint32, C.c_int]
    L.p3_comm_create_callbacks.argtypes = [C.POINTER(vp), vp, C.c_uint32, C.c_uint32]
    L.p3_comm_destroy.argtypes = [vp]
    L.p3_comm_all_reduce_u64.argtypes = [vp, C.POINTER(u64), C.c_uint32, C.c_int]
    L.p3_group_create.argtypes = [C.POINTER(vp), C.c_uint32, vp]
    L.p3_group_destroy.argtypes = [vp]
    L.p3_comm_create_member.argtypes = [C.POINTER(vp), vp, C.c_uint32]
    L.p3_comm_last_error.argtypes = [vp]
    L.p3_comm_last_error.restype = C.c_char_p
    L.p3_mr_run_pass.argtypes = [vp, vp, C.c_uint32, u64, C.c_uint32]
    L.p3_mr_build_graph.argtypes = [vp, vp]
    L.p3_mr_reshard_reads.argtypes = [vp, vp]
    L.p3_group_reads.argtypes = [vp, C.c_uint32]
    L.p3_write_gfa.argtypes = [vp, C.c_char_p]
    L.p3_mr_config.argtypes = [vp, u32, u32]
    L.p3_mr_count_partition.argtypes = [vp]
    L.p3_mr_count_send_layout.argtypes = [vp, C.POINTER(MrSendLayout)]
    L.p3_mr_count_recv_alloc.argtypes = [vp, C.POINTER(u64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.p3_mr_initial_sbits.argtypes = [vp, C.POINTER(u32)]
    L.p3_mr_count_owned.argtypes = [vp, u32, C.POINTER(C.c_int), C.POINTER(u64)]
    L.p3_rs_log2_for.restype = u32
    L.p3_rs_log2_for.argtypes = [u64]
    L.p3_mr_solid_build.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.p3_mr_singles_layout.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.p3_mr_nonsolid_recv_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.p3_mr_solid_bits.argtypes = [vp, u64]
    L.p3_mr_bloom_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.p3_mr_or_reduce.argtypes = [vp, vp, vp, u64, u32, u64]
    L.p3_mr_alloc_scratch.argtypes = [vp, u64, C.POINTER(vp)]
    L.p3_last_timings.argtypes = [vp, C.POINTER(Timings)]
    L.p3_last_traffic.argtypes = [vp, C.POINTER(Traffic)]
    L.p3_kernel_stat_count.argtypes = [vp]
    L.p3_kernel_stat.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(u64), C.POINTER(C.c_double)]
    L.p3_reset_kernel_stats.argtypes = [vp]
    _lib = L
    return L


def exported_symbols():
    """Every entry point include/platanus3_amd.h declares (used by the CPU-side ABI test)."""
    import re
    with open(os.path.join(_build.INCLUDE, "platanus3_amd.h")) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(p3_[a-z0-9_]+)\s*\(", txt)) - {"p3_status"})


def nwords(k):
    return (2 * k + 63) // 64


class Context:
    """One p3_ctx (one HIP device)."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.p3_create(C.byref(h), device)
        if rc != P3_OK:
            raise P3Error(rc, "p3_create(device=%d): no usable HIP device?" % device)
        self.h = h
        self.k = None
        self.n_ranks = 1

    def close(self):
        if getattr(self, "h", None):
            self.L.p3_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != P3_OK:
            raise P3Error(rc, (self.L.p3_last_error(self.h) or b"").decode())

    def set_option(self, name, value):
        self._chk(self.L.p3_set_option(self.h, name.encode(), value))

    def get_info(self, name):
        v = C.c_int64()
        self._chk(self.L.p3_get_info(self.h, name.encode(), C.byref(v)))
        return v.value

    # ---- T1 -----------------------------------------------------------------------------------------
    def load_reads(self, bases, offsets, min_len):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._keep = (bases, offsets)
        self._chk(self.L.p3_load_reads(self.h, bases.ctypes.data, offsets.ctypes.data, offsets.size - 1, min_len))

    def load_reads_device(self, d_bases, d_offsets, n_reads, total_bases, min_len):
        self._chk(self.L.p3_load_reads_device(self.h, d_bases, d_offsets, n_reads, total_bases, min_len))

    # ---- C1 -----------------------------------------------------------------------------------------
    def count_short(self):
        self._chk(self.L.p3_count_short(self.h))
        return self.count_stats()

    def count_stats(self):
        s = CountStats()
        self._chk(self.L.p3_get_count_stats(self.h, C.byref(s)))
        return dict(n_windows=s.n_windows, n_distinct=s.n_distinct, n_solid=s.n_solid, max_count=s.max_count)

    def export_counts(self):
        n = C.c_uint64()
        nd = self.count_stats()["n_distinct"]
        keys = np.zeros(max(1, nd), np.uint64)
        counts = np.zeros(max(1, nd), np.uint64)
        self._chk(self.L.p3_export_counts(self.h, keys.ctypes.data, counts.ctypes.data, keys.size, C.byref(n)))
        return keys[: n.value], counts[: n.value]

    # ---- G1+G2 --------------------------------------------------------------------------------------
    def make_bloom(self, k, m, H):
        self._chk(self.L.p3_make_bloom(self.h, k, m, H))
        self.k, self.m, self.H = k, m, H
        return self.bloom_stats()

    def run_pass(self, k, m, H):
        self._chk(self.L.p3_run_pass(self.h, k, m, H))
        self.k, self.m, self.H = k, m, H

    def bloom_stats(self):
        s = BloomStats()
        self._chk(self.L.p3_get_bloom_stats(self.h, C.byref(s)))
        return dict(n_windows_k=s.n_windows_k, n_passing=s.n_passing, n_seed_reads=s.n_seed_reads)

    def count_digest(self):
        """(sum, xor) digest of the resident count table (order-independent; see include/platanus3_amd.h)"""
        out = (C.c_uint64 * 2)()
        self._chk(self.L.p3_count_digest(self.h, out))
        return int(out[0]), int(out[1])

    def export_bloom(self):
        words = np.zeros((self.m + 63) // 64, np.uint64)
        self._chk(self.L.p3_export_bloom(self.h, words.ctypes.data))
        return words

    def export_seeds(self, cap=None):
        nw = nwords(self.k)
        n = C.c_uint64()
        cap = cap if cap is not None else max(1, self.bloom_stats()["n_seed_reads"])
        out = np.zeros(cap * nw, np.uint64)
        self._chk(self.L.p3_export_seeds(self.h, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value * nw].reshape(n.value, nw)

    def bloom_query(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1)
        n = keys.size // nwords(self.k)
        out = np.zeros(max(1, n), np.uint8)
        self._chk(self.L.p3_bloom_query(self.h, keys.ctypes.data, n, out.ctypes.data))
        return out[:n]

    # ---- G3-G6 ---------------------------------------------------------------------------------------
    def build_graph(self):
        self._chk(self.L.p3_build_graph(self.h))
        return self.graph_stats()

    def build_graph_from(self, seed_keys):
        """MakeDBG over a seed set of the caller's (forward-strand keys, nw words each) instead of the resident one."""
        keys = np.ascontiguousarray(seed_keys, dtype=np.uint64).reshape(-1)
        n = keys.size // nwords(self.k)
        self._chk(self.L.p3_build_graph_from(self.h, keys.ctypes.data if n else None, n))
        return self.graph_stats()

    def graph_stats(self):
        g = GraphStats()
        self._chk(self.L.p3_get_graph_stats(self.h, C.byref(g)))
        return {f: getattr(g, f) for f, _ in GraphStats._fields_}

    def write_gfa(self, path):
        self._chk(self.L.p3_write_gfa(self.h, path.encode()))

    def export_junctions(self):
        g = self.graph_stats()
        nj, nw = g["n_junctions"], nwords(self.k)
        keys = np.zeros(max(1, nj) * nw, np.uint64)
        cov = np.zeros(max(1, nj), np.int32)
        lc = np.zeros(max(1, nj) * 4, np.int32)
        rc = np.zeros(max(1, nj) * 4, np.int32)
        n = C.c_uint64()
        self._chk(self.L.p3_export_junctions(self.h, keys.ctypes.data, cov.ctypes.data, lc.ctypes.data, rc.ctypes.data,
                                             max(1, nj), C.byref(n)))
        return keys[:nj * nw].reshape(nj, nw), cov[:nj], lc[:nj * 4].reshape(nj, 4), rc[:nj * 4].reshape(nj, 4)

    def export_straights(self):
        """straight sequences (G3 StraightInfo::sequence), in id order"""
        g = self.graph_stats()
        n, nb = int(g["n_straights"]), int(g["straight_bases"])
        so = np.zeros(n + 1, np.uint64)
        seq = np.zeros(nb + 1, np.uint8)
        cnt = C.c_uint64()
        self._chk(self.L.p3_export_straights(self.h, so.ctypes.data, seq.ctypes.data, n + 1, nb + 1, C.byref(cnt)))
        return [seq[int(so[i]):int(so[i + 1])].tobytes().decode() for i in range(n)]

    def export_joints(self):
        """JointInfo::coverage, two per straight: [2i] = the joint that starts straight i's sequence, [2i+1] ends it"""
        nj = int(self.graph_stats()["n_joints"])
        cov = np.zeros(max(1, nj), np.int32)
        n = C.c_uint64()
        self._chk(self.L.p3_export_joints(self.h, cov.ctypes.data, max(1, nj), C.byref(n)))
        return cov[:nj]

    def joint_coverage(self):
        """[(joint k-mer in its smaller orientation, coverage)] sorted: the ID-free form the tests compare"""
        comp = bytes.maketrans(b"ACGT", b"TGCA")
        k, cov, out = self.k, self.export_joints(), []
        for i, sq in enumerate(self.export_straights()):
            for kmer, c in ((sq[:k], cov[2 * i]), (sq[-k:], cov[2 * i + 1])):
                rc = kmer.encode().translate(comp)[::-1].decode()
                out.append((min(kmer, rc), int(c)))
        return sorted(out)

    def mr_run_pass(self, comm, k, m, H):
        """one rank's share of the sharded pass with the exchanges INSIDE the library (p3_exchange.hip)"""
        self._chk(self.L.p3_mr_run_pass(self.h, comm.h, k, m, H))
        self.k, self.m, self.H = k, m, H

    def group_reads(self, n_groups=0):
        """one context, large input: put the reads in the order of their shard key's group (once, after load_reads)"""
        self._chk(self.L.p3_group_reads(self.h, n_groups))

    def mr_reshard_reads(self, comm):
        """deal the reads of a sharded job out once more, by genomic neighbourhood (every rank, once, before the passes)"""
        self._chk(self.L.p3_mr_reshard_reads(self.h, comm.h))

    def mr_build_graph(self, comm):
        """the de Bruijn stage of a sharded job (every rank calls it after mr_run_pass); rank 0 ends up with the graph"""
        self._chk(self.L.p3_mr_build_graph(self.h, comm.h))

    # ---- multi-rank protocol (device pointers are returned as plain ints) ----------------------------
    def mr_config(self, rank, n_ranks):
        self._chk(self.L.p3_mr_config(self.h, rank, n_ranks))
        self.n_ranks = n_ranks

    def mr_count_partition(self):
        self._chk(self.L.p3_mr_count_partition(self.h))

    def mr_count_send_layout(self):
        lay = MrSendLayout()
        self._chk(self.L.p3_mr_count_send_layout(self.h, C.byref(lay)))
        n = self.n_ranks
        return dict(d_part=lay.d_part, d_ghist=lay.d_ghist, d_segoff=lay.d_segoff,
                    rec_off=[int(lay.rec_off[i]) for i in range(n + 1)], rows=lay.rows_per_rank, row_u32=lay.row_u32,
                    rec_bytes=lay.rec_bytes)

    def mr_count_recv_alloc(self, recv_records):
        arr = (C.c_uint64 * len(recv_records))(*[int(x) for x in recv_records])
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.p3_mr_count_recv_alloc(self.h, arr, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def mr_initial_sbits(self):
        v = C.c_uint32()
        self._chk(self.L.p3_mr_initial_sbits(self.h, C.byref(v)))
        return v.value

    def mr_count_owned(self, sbits):
        ovf, mx = C.c_int(), C.c_uint64()
        self._chk(self.L.p3_mr_count_owned(self.h, sbits, C.byref(ovf), C.byref(mx)))
        return ovf.value, mx.value

    def rs_log2_for(self, max_solid):
        return int(self.L.p3_rs_log2_for(max_solid))

    def mr_solid_build(self, rs_log2):
        p, off, own, tot = C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.p3_mr_solid_build(self.h, rs_log2, C.byref(p), C.byref(off), C.byref(own), C.byref(tot)))
        return p.value, off.value, own.value, tot.value

    def mr_singles_layout(self):
        p = C.c_void_p()
        offs = (C.c_uint64 * self.n_ranks)()
        cnts = (C.c_uint64 * self.n_ranks)()
        self._chk(self.L.p3_mr_singles_layout(self.h, C.byref(p), offs, cnts))
        return p.value, [int(x) for x in offs], [int(x) for x in cnts]

    def mr_nonsolid_recv_alloc(self, total):
        p = C.c_void_p()
        self._chk(self.L.p3_mr_nonsolid_recv_alloc(self.h, total, C.byref(p)))
        return p.value

    def mr_solid_bits(self, n_positions):
        self._chk(self.L.p3_mr_solid_bits(self.h, n_positions))

    def mr_bloom_buffer(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.L.p3_mr_bloom_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def mr_or_reduce(self, d_dst, d_src, n_words, n_src, stride_words):
        self._chk(self.L.p3_mr_or_reduce(self.h, d_dst, d_src, n_words, n_src, stride_words))

    def mr_alloc_scratch(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.p3_mr_alloc_scratch(self.h, nbytes, C.byref(p)))
        return p.value

    # ---- timings ------------------------------------------------------------------------------------
    def set_profiling(self, on, only=None):
        self._chk(self.L.p3_set_profile_filter(self.h, only.encode() if only else None))
        self._chk(self.L.p3_set_profiling(self.h, 1 if on else 0))

    def timings(self):
        t = Timings()
        self._chk(self.L.p3_last_timings(self.h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Timings._fields_}

    def traffic(self):
        """algorithmic bytes of the last pass per phase (SURVEY.md section 8d accounting)"""
        t = Traffic()
        self._chk(self.L.p3_last_traffic(self.h, C.byref(t)))
        return {f: int(getattr(t, f)) for f, _ in Traffic._fields_}

    def kernel_stats(self):
        out = {}
        for i in range(self.L.p3_kernel_stat_count(self.h)):
            name = C.create_string_buffer(64)
            n, ms = C.c_uint64(), C.c_double()
            self._chk(self.L.p3_kernel_stat(self.h, i, name, C.byref(n), C.byref(ms)))
            out[name.value.decode()] = dict(launches=n.value, total_ms=ms.value)
        return out

    def reset_kernel_stats(self):
        self._chk(self.L.p3_reset_kernel_stats(self.h))


def decode_key(words, k):
    """Key words (low word first) -> ACGT string (reference GetStringKmer, src/BitCalc.cpp:56-65)."""
    v = 0
    for i, w in enumerate(words):
        v |= int(w) << (64 * i)
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


# ---- communicators of the in-library exchange ---------------------------------------------------------------------------
A2AV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p,
                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p)
AG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
AR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_int)


class CommCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_to_all_v", A2AV_FN), ("all_gather", AG_FN), ("all_reduce_u64", AR_FN)]


def comm_unique_id():
    """rank 0: the id every rank needs for Comm.rccl (hand it over by any channel)"""
    buf = (C.c_uint8 * 128)()
    rc = load().p3_comm_unique_id(buf)
    if rc != 0:
        raise P3Error(rc, "p3_comm_unique_id (librccl not loadable?)")
    return bytes(buf)


class Comm:
    """p3_comm: RCCL (Comm.rccl) or caller callbacks (Comm.callbacks: the three functions get DEVICE pointers)"""

    def __init__(self, h, keep=None):
        self.h, self._keep = h, keep

    @classmethod
    def rccl(cls, unique_id, rank, n_ranks, device):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = load().p3_comm_create_rccl(C.byref(h), buf, rank, n_ranks, device)
        if rc != 0:
            raise P3Error(rc, "p3_comm_create_rccl")
        return cls(h)

    @classmethod
    def callbacks(cls, rank, n_ranks, all_to_all_v, all_gather, all_reduce_u64):
        def guard(fn):
            def run(*a):
                try:
                    return int(fn(*a[1:]) or 0)
                except Exception:  # noqa: BLE001  (an exception must not cross the C frames)
                    import traceback
                    traceback.print_exc()
                    return 1
            return run
        cb = CommCallbacks(None, A2AV_FN(guard(all_to_all_v)), AG_FN(guard(all_gather)), AR_FN(guard(all_reduce_u64)))
        h = C.c_void_p()
        rc = load().p3_comm_create_callbacks(C.byref(h), C.byref(cb), rank, n_ranks)
        if rc != 0:
            raise P3Error(rc, "p3_comm_create_callbacks")
        return cls(h, keep=cb)

    @classmethod
    def member(cls, group, local_index):
        """rank `local_index` of a Group (several protocol ranks in this process)"""
        h = C.c_void_p()
        rc = load().p3_comm_create_member(C.byref(h), group.h, local_index)
        if rc != 0:
            raise P3Error(rc, "p3_comm_create_member")
        return cls(h, keep=group)

    def all_reduce(self, values, op="sum"):
        """host integers reduced over the ranks (every rank calls it); op: "max" or "sum" """
        vals = list(values)
        arr = (C.c_uint64 * len(vals))(*vals)
        rc = load().p3_comm_all_reduce_u64(self.h, arr, len(vals), 0 if op == "max" else 1)
        if rc != 0:
            raise P3Error(rc, "p3_comm_all_reduce_u64: " + load().p3_comm_last_error(self.h).decode())
        return [int(x) for x in arr]

    def close(self):
        if self.h:
            load().p3_comm_destroy(self.h)
            self.h = None


class Group:
    """p3_group: `n_local` protocol ranks in this process (one context and one thread each); `outer` (a Comm with one rank
    per process, or None) carries what leaves the process"""

    def __init__(self, n_local, outer=None):
        self.h = C.c_void_p()
        self.outer = outer
        rc = load().p3_group_create(C.byref(self.h), n_local, outer.h if outer is not None else None)
        if rc != 0:
            raise P3Error(rc, "p3_group_create")

    def close(self):
        if self.h:
            load().p3_group_destroy(self.h)
            self.h = None
