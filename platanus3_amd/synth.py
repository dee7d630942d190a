"""Synthetic read generator -- ctypes front for csrc/p3_synth.c (CPU only)."""
import ctypes as C

import numpy as np

from . import _build

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build_synth())
        _lib.p3_synth_reads.restype = C.c_int
        _lib.p3_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.p3_synth_genome_length.restype = C.c_uint64
        _lib.p3_synth_genome_length.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    return _lib


def reads(seed, total_reads, n_reads=None, first_read=0, read_len=150, coverage=30, sub_rate_ppm=5000):
    """Reads [first_read, first_read+n_reads) of a global set of total_reads -> (uint8 bases, uint64 offsets)."""
    n_reads = total_reads if n_reads is None else n_reads
    bases = np.empty(max(1, n_reads * read_len), dtype=np.uint8)
    offs = np.empty(n_reads + 1, dtype=np.uint64)
    rc = _load().p3_synth_reads(seed, total_reads, first_read, n_reads, read_len, coverage, sub_rate_ppm,
                                bases.ctypes.data, offs.ctypes.data)
    if rc != 0:
        raise RuntimeError("p3_synth_reads failed: %d" % rc)
    return bases[: n_reads * read_len], offs
