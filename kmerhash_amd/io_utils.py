"""The reference's binary dump format (io_utils.hpp:57-103 serialize_vector / deserialize_vector):
    size_t element_size ; size_t count ; raw elements
used by `benchmark_hashtables -F file` (BenchmarkHashTables.cpp:241-248) and -DDUMP_DISTRIBUTED_INPUT.
Elements on this path are std::pair<uint64_t,uint32_t> (16 bytes: key @0, value @8, 4 bytes of padding) or bare
uint64_t keys.  Host-side only."""
import numpy as np

PAIR_DTYPE = np.dtype({"names": ["key", "val"], "formats": [np.uint64, np.uint32], "offsets": [0, 8], "itemsize": 16})


def serialize_pairs(keys, vals, filename):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    vals = np.ascontiguousarray(vals, dtype=np.uint32)
    rec = np.zeros(len(keys), dtype=PAIR_DTYPE)
    rec["key"] = keys
    rec["val"] = vals
    with open(filename, "wb") as f:
        np.array([16, len(keys)], dtype=np.uint64).tofile(f)
        rec.tofile(f)


def serialize_keys(keys, filename):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    with open(filename, "wb") as f:
        np.array([8, len(keys)], dtype=np.uint64).tofile(f)
        keys.tofile(f)


def _header(f):
    h = np.fromfile(f, dtype=np.uint64, count=2)
    if len(h) != 2:
        raise ValueError("truncated header")
    return int(h[0]), int(h[1])


def deserialize_pairs(filename):
    """-> (keys u64[n], vals u32[n]); raises like the reference's std::logic_error on an element-size mismatch"""
    with open(filename, "rb") as f:
        el, n = _header(f)
        if el != 16:
            raise ValueError("input element size not as specified ")     # io_utils.hpp:86
        rec = np.fromfile(f, dtype=PAIR_DTYPE, count=n)
    if len(rec) != n:
        raise ValueError("truncated file")
    return rec["key"].copy(), rec["val"].copy()


def deserialize_keys(filename):
    with open(filename, "rb") as f:
        el, n = _header(f)
        if el != 8:
            raise ValueError("input element size not as specified ")
        k = np.fromfile(f, dtype=np.uint64, count=n)
    if len(k) != n:
        raise ValueError("truncated file")
    return k
