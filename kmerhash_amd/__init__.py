"""kmerhash_amd: MI355X (gfx950) open-addressing k-mer hash tables behind the ParBLiSS/kmerhash table API.

The product is libkmerhash_amd.so (hand-written HIP, C-ABI in include/kmerhash_amd.h); this package is
the thin host-side mirror of the reference's interface for Python callers plus workload generators.
"""
from .table import (hashmap_robinhood_doubling, hashmap_linearprobe_doubling, hash_batch, HASHES,  # noqa: F401
                    KhError, KhLogicError, KhRetry)
from . import workloads  # noqa: F401

__all__ = ["hashmap_robinhood_doubling", "hashmap_linearprobe_doubling", "hash_batch", "HASHES", "KhError",
           "KhLogicError", "KhRetry", "workloads"]
