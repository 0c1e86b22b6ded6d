"""dbg_assembly_amd -- MI355X-native k-mer counting / de Bruijn graph construction.

The product is the HIP library behind include/dbgk.h (csrc/ -> lib/libdbgk.so) and the C++ host
layer that mirrors the reference's kmerSet / DBGgraph API (host/).  This Python package is only
the ctypes plumbing used by bench.py and the tests.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
