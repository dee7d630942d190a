"""platanus3_amd -- MI355X-native k-mer count -> coverage gate -> Bloom build -> seed path of platanus3.

The compute lives in csrc/ (hand-written HIP for gfx950 behind the C ABI of include/platanus3_amd.h).
This package is the thin Python host used by tests and bench.py; the C++ host shim that mirrors the
reference's Assemble.cpp call surface is in host/.
"""
from . import synth  # noqa: F401  (CPU-only generator; safe without a GPU)

__all__ = ["synth"]
