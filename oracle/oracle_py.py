"""ctypes bindings for the CPU oracle libraries (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Nothing under kmerhash_amd/ may import it (tests/test_layout.py enforces that).

  OracleTable  -> oracle/_build/libkh_oracle.so  (own C++ restatement, kh_oracle.hpp)
  RefLPTable   -> oracle/_ref/libref_lp.so       (the real reference LP table; built only where
                                                   /root/reference exists, prebuilt file used elsewhere)
  smhasher_*   -> oracle/_build/libsmhasher.so   (scikit-learn's smhasher copy; golden generation)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KIND_RH, KIND_LP = 0, 1
HASH_IDENTITY, HASH_MURMUR3_X86, HASH_MURMUR3_X64, HASH_FARM = 0, 1, 2, 3

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t)


def build(targets=("all",)):
    """(Re)build oracle libraries with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", HERE] + list(targets))


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "_build", "libkh_oracle.so")
        if not os.path.exists(path):
            build(("all",))
        L = C.CDLL(path)
        L.ora_hash_u64.restype = C.c_uint64
        L.ora_hash_u64.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
        L.ora_hash_batch.argtypes = [C.c_int, C.c_uint64, _u64p, C.c_uint64, _u64p]
        L.ora_murmur3_x86_128.argtypes = [C.c_void_p, C.c_int, C.c_uint32, _u32p]
        L.ora_murmur3_x64_128.argtypes = [C.c_void_p, C.c_int, C.c_uint32, _u64p]
        L.ora_next_power_of_2.restype = C.c_uint64
        L.ora_next_power_of_2.argtypes = [C.c_uint64]
        L.ora_load_threshold.restype = C.c_uint64
        L.ora_load_threshold.argtypes = [C.c_uint64, C.c_float]
        L.ora_create.restype = C.c_void_p
        L.ora_set_key_transform.argtypes = [C.c_void_p, C.c_uint32]
        L.ora_set_key_transform.restype = None
        L.ora_pre_transform.argtypes = [C.c_uint64, C.c_uint32]
        L.ora_pre_transform.restype = C.c_uint64
        L.ora_create.argtypes = [C.c_int, C.c_uint64, C.c_float, C.c_float, C.c_int, C.c_uint64]
        L.ora_destroy.argtypes = [C.c_void_p]
        for f in ("ora_size", "ora_capacity", "ora_max_load", "ora_min_load"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.ora_set_min_load_factor.argtypes = [C.c_void_p, C.c_float]
        L.ora_set_max_load_factor.argtypes = [C.c_void_p, C.c_float]
        L.ora_clear.argtypes = [C.c_void_p]
        L.ora_reserve.argtypes = [C.c_void_p, C.c_uint64]
        L.ora_rehash.argtypes = [C.c_void_p, C.c_uint64]
        L.ora_rehash.restype = C.c_int
        L.ora_probe_overflow.argtypes = [C.c_void_p]
        L.ora_insert.restype = C.c_int64
        L.ora_insert.argtypes = [C.c_void_p, _u64p, _u32p, C.c_uint64]
        L.ora_insert_one.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.ora_update_one.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.ora_count.argtypes = [C.c_void_p, _u64p, C.c_uint64, _u8p]
        L.ora_find.argtypes = [C.c_void_p, _u64p, C.c_uint64, _u32p, _u8p]
        L.ora_find_compact.restype = C.c_uint64
        L.ora_find_compact.argtypes = [C.c_void_p, _u64p, C.c_uint64, _u64p, _u32p]
        L.ora_erase.restype = C.c_int64
        L.ora_erase.argtypes = [C.c_void_p, _u64p, C.c_uint64]
        L.ora_erase_one.argtypes = [C.c_void_p, C.c_uint64]
        L.ora_export_info.argtypes = [C.c_void_p, _u8p]
        L.ora_export_slots.argtypes = [C.c_void_p, _u64p, _u32p]
        L.ora_to_vector.restype = C.c_uint64
        L.ora_to_vector.argtypes = [C.c_void_p, _u64p, _u32p]
        L.ora_displacement_histogram.argtypes = [C.c_void_p, _u64p]
        L.ora_timed_insert.restype = C.c_double
        L.ora_timed_insert.argtypes = [C.c_void_p, _u64p, _u32p, C.c_uint64]
        L.ora_timed_find.restype = C.c_double
        L.ora_timed_find.argtypes = [C.c_void_p, _u64p, C.c_uint64, _u64p, _u32p, _u64p]
        _lib = L
    return _lib


def _k(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _v(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def hash_batch(hash_id, seed, keys):
    keys = _k(keys)
    out = np.empty_like(keys)
    lib().ora_hash_batch(hash_id, seed, _p(keys, _u64p), len(keys), _p(out, _u64p))
    return out


def murmur3_x86_128(data: bytes, seed: int):
    out = np.zeros(4, dtype=np.uint32)
    buf = C.create_string_buffer(data, len(data))
    lib().ora_murmur3_x86_128(buf, len(data), seed, _p(out, _u32p))
    return out


def murmur3_x64_128(data: bytes, seed: int):
    out = np.zeros(2, dtype=np.uint64)
    buf = C.create_string_buffer(data, len(data))
    lib().ora_murmur3_x64_128(buf, len(data), seed, _p(out, _u64p))
    return out


class _TableBase:
    """Common surface: mirrors the reference member names (insert/find/count/erase/size/...)."""

    _pfx = None
    _L = None
    h = None

    def _f(self, name):
        return getattr(self._L, self._pfx + name)

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self._f("size")(self.h))

    def capacity(self):
        return int(self._f("capacity")(self.h))

    def max_load(self):
        return int(self._f("max_load")(self.h))

    def min_load(self):
        return int(self._f("min_load")(self.h))

    def set_key_transform(self, k):
        """PreTransform = lex_less on 2-bit packed DNA k-mers of length k (0: identity); only on an empty table"""
        self._L.ora_set_key_transform(self.h, int(k))

    def set_min_load_factor(self, f):
        self._f("set_min_load_factor")(self.h, f)

    def set_max_load_factor(self, f):
        self._f("set_max_load_factor")(self.h, f)

    def clear(self):
        self._f("clear")(self.h)

    def reserve(self, n):
        self._f("reserve")(self.h, n)

    def rehash(self, b):
        if self._f("rehash")(self.h, b) != 0:
            raise RuntimeError("logic_error")

    def insert(self, keys, vals):
        keys, vals = _k(keys), _v(vals)
        r = self._f("insert")(self.h, _p(keys, _u64p), _p(vals, _u32p), len(keys))
        if r < 0:
            raise RuntimeError("logic_error")
        return int(r)

    def insert_one(self, key, val):
        return bool(self._f("insert_one")(self.h, int(key), int(val)))

    def update_one(self, key, val):
        self._f("update_one")(self.h, int(key), int(val))

    def count(self, keys):
        keys = _k(keys)
        out = np.zeros(len(keys), dtype=np.uint8)
        self._f("count")(self.h, _p(keys, _u64p), len(keys), _p(out, _u8p))
        return out

    def find_compact(self, keys):
        keys = _k(keys)
        ok = np.zeros(len(keys), dtype=np.uint64)
        ov = np.zeros(len(keys), dtype=np.uint32)
        m = self._f("find_compact")(self.h, _p(keys, _u64p), len(keys), _p(ok, _u64p), _p(ov, _u32p))
        return ok[:m].copy(), ov[:m].copy()

    def erase(self, keys):
        keys = _k(keys)
        r = self._f("erase")(self.h, _p(keys, _u64p), len(keys))
        if r < 0:
            raise RuntimeError("logic_error")
        return int(r)

    def erase_one(self, key):
        return int(self._f("erase_one")(self.h, int(key)))

    def export_info(self):
        out = np.zeros(self.capacity(), dtype=np.uint8)
        self._f("export_info")(self.h, _p(out, _u8p))
        return out

    def export_slots(self):
        k = np.zeros(self.capacity(), dtype=np.uint64)
        v = np.zeros(self.capacity(), dtype=np.uint32)
        self._f("export_slots")(self.h, _p(k, _u64p), _p(v, _u32p))
        return k, v

    def to_vector(self):
        k = np.zeros(self.size(), dtype=np.uint64)
        v = np.zeros(self.size(), dtype=np.uint32)
        m = self._f("to_vector")(self.h, _p(k, _u64p), _p(v, _u32p))
        assert m == len(k)
        return k, v

    def sorted_items(self):
        k, v = self.to_vector()
        o = np.argsort(k, kind="stable")
        return k[o], v[o]


class OracleTable(_TableBase):
    _pfx = "ora_"

    def __init__(self, kind, capacity=128, min_lf=None, max_lf=None, hash_id=HASH_MURMUR3_X86, seed=43):
        self._L = lib()
        self.kind = kind
        if min_lf is None:
            min_lf = 0.4 if kind == KIND_RH else 0.2   # hashmap_robinhood.hpp:219 / hashmap_linearprobe.hpp:192
        if max_lf is None:
            max_lf = 0.9 if kind == KIND_RH else 0.6
        self.h = self._L.ora_create(kind, capacity, min_lf, max_lf, hash_id, seed)

    def find(self, keys):
        keys = _k(keys)
        ov = np.zeros(len(keys), dtype=np.uint32)
        of = np.zeros(len(keys), dtype=np.uint8)
        self._L.ora_find(self.h, _p(keys, _u64p), len(keys), _p(ov, _u32p), _p(of, _u8p))
        return ov, of

    def probe_overflow(self):
        return bool(self._L.ora_probe_overflow(self.h))

    def displacement_histogram(self):
        out = np.zeros(128, dtype=np.uint64)
        self._L.ora_displacement_histogram(self.h, _p(out, _u64p))
        return out

    def timed_insert(self, keys, vals):
        keys, vals = _k(keys), _v(vals)
        return float(self._L.ora_timed_insert(self.h, _p(keys, _u64p), _p(vals, _u32p), len(keys)))

    def timed_find(self, keys):
        keys = _k(keys)
        ok = np.zeros(len(keys), dtype=np.uint64)
        ov = np.zeros(len(keys), dtype=np.uint32)
        nf = np.zeros(1, dtype=np.uint64)
        t = float(self._L.ora_timed_find(self.h, _p(keys, _u64p), len(keys), _p(ok, _u64p), _p(ov, _u32p), _p(nf, _u64p)))
        return t, int(nf[0])


_ref = None


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_lp.so")) or os.path.isdir("/root/reference/include/kmerhash")


def ref_lib():
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref_lp.so")
        if not os.path.exists(path):
            if not os.path.isdir("/root/reference/include/kmerhash"):
                raise RuntimeError("reference LP library not built and /root/reference absent")
            build(("ref",))
        L = C.CDLL(path)
        L.ref_lp_create.restype = C.c_void_p
        L.ref_lp_create.argtypes = [C.c_uint64, C.c_float, C.c_float, C.c_int, C.c_uint64]
        L.ref_lp_destroy.argtypes = [C.c_void_p]
        for f in ("ref_lp_size", "ref_lp_capacity", "ref_lp_max_load", "ref_lp_min_load"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.ref_lp_set_min_load_factor.argtypes = [C.c_void_p, C.c_float]
        L.ref_lp_set_max_load_factor.argtypes = [C.c_void_p, C.c_float]
        L.ref_lp_clear.argtypes = [C.c_void_p]
        L.ref_lp_reserve.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_lp_rehash.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_lp_insert.restype = C.c_int64
        L.ref_lp_insert.argtypes = [C.c_void_p, _u64p, _u32p, C.c_uint64]
        L.ref_lp_insert_one.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.ref_lp_update_one.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.ref_lp_count.argtypes = [C.c_void_p, _u64p, C.c_uint64, _u8p]
        L.ref_lp_find_compact.restype = C.c_uint64
        L.ref_lp_find_compact.argtypes = [C.c_void_p, _u64p, C.c_uint64, _u64p, _u32p]
        L.ref_lp_erase.restype = C.c_int64
        L.ref_lp_erase.argtypes = [C.c_void_p, _u64p, C.c_uint64]
        L.ref_lp_erase_one.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_lp_export_info.argtypes = [C.c_void_p, _u8p]
        L.ref_lp_export_slots.argtypes = [C.c_void_p, _u64p, _u32p]
        L.ref_lp_to_vector.restype = C.c_uint64
        L.ref_lp_to_vector.argtypes = [C.c_void_p, _u64p, _u32p]
        L.ref_lp_timed_insert.restype = C.c_double
        L.ref_lp_timed_insert.argtypes = [C.c_void_p, _u64p, _u32p, C.c_uint64]
        L.ref_lp_timed_count.restype = C.c_double
        L.ref_lp_timed_count.argtypes = [C.c_void_p, _u64p, C.c_uint64, _u64p]
        _ref = L
    return _ref


class RefLPTable(_TableBase):
    """The real fsc::hashmap_linearprobe_doubling<uint64_t,uint32_t,Hash> from the reference tree."""

    _pfx = "ref_lp_"
    kind = KIND_LP

    def __init__(self, capacity=128, min_lf=0.2, max_lf=0.6, hash_id=HASH_MURMUR3_X86, seed=43):
        self._L = ref_lib()
        self.h = self._L.ref_lp_create(capacity, min_lf, max_lf, hash_id, seed)

    def timed_insert(self, keys, vals):
        keys, vals = _k(keys), _v(vals)
        return float(self._L.ref_lp_timed_insert(self.h, _p(keys, _u64p), _p(vals, _u32p), len(keys)))

    def timed_count(self, keys):
        keys = _k(keys)
        nf = np.zeros(1, dtype=np.uint64)
        t = float(self._L.ref_lp_timed_count(self.h, _p(keys, _u64p), len(keys), _p(nf, _u64p)))
        return t, int(nf[0])


_smh = None


def smhasher():
    global _smh
    if _smh is None:
        path = os.path.join(HERE, "_build", "libsmhasher.so")
        if not os.path.exists(path):
            build(("smhasher",))
        L = C.CDLL(path)
        L.smh_x86_128.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        L.smh_x64_128.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        _smh = L
    return _smh


def smhasher_x86_128(data: bytes, seed: int):
    out = np.zeros(4, dtype=np.uint32)
    buf = C.create_string_buffer(data, len(data))
    smhasher().smh_x86_128(buf, len(data), seed, out.ctypes.data)
    return out


def smhasher_x64_128(data: bytes, seed: int):
    out = np.zeros(2, dtype=np.uint64)
    buf = C.create_string_buffer(data, len(data))
    smhasher().smh_x64_128(buf, len(data), seed, out.ctypes.data)
    return out


# ---- HyperLogLog: oracle restatement and the real reference (oracle/_ref/libref_hll.so) ----------------------
class _HLLBase:
    _pfx = None
    _L = None
    h = None

    def _f(self, n):
        return getattr(self._L, self._pfx + n)

    def update(self, keys):
        keys = _k(keys)
        self._f("update")(self.h, _p(keys, _u64p), len(keys))

    def update_via_hashval(self, hv):
        hv = _k(hv)
        self._f("update_via_hashval")(self.h, _p(hv, _u64p), len(hv))

    def merge(self, other):
        self._f("merge")(self.h, other.h)

    def clear(self):
        self._f("clear")(self.h)

    def estimate(self):
        return float(self._f("estimate")(self.h))

    def registers(self):
        out = np.zeros(1 << self.precision, dtype=np.uint8)
        self._f("registers")(self.h, _p(out, _u8p))
        return out

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _decl_hll(L, pfx, create_args):
    getattr(L, pfx + "create").restype = C.c_void_p
    getattr(L, pfx + "create").argtypes = create_args
    getattr(L, pfx + "destroy").argtypes = [C.c_void_p]
    getattr(L, pfx + "update").argtypes = [C.c_void_p, _u64p, C.c_uint64]
    getattr(L, pfx + "update_via_hashval").argtypes = [C.c_void_p, _u64p, C.c_uint64]
    getattr(L, pfx + "merge").argtypes = [C.c_void_p, C.c_void_p]
    getattr(L, pfx + "clear").argtypes = [C.c_void_p]
    getattr(L, pfx + "estimate").restype = C.c_double
    getattr(L, pfx + "estimate").argtypes = [C.c_void_p]
    getattr(L, pfx + "registers").argtypes = [C.c_void_p, _u8p]


class OracleHLL(_HLLBase):
    _pfx = "ora_hll_"

    def __init__(self, precision=12, ignore_msb=0, hash_id=HASH_MURMUR3_X86, seed=43):
        self._L = lib()
        _decl_hll(self._L, self._pfx, [C.c_uint32, C.c_uint32, C.c_int, C.c_uint64])
        self.precision = precision
        self.h = self._L.ora_hll_create(precision, ignore_msb, hash_id, seed)


_ref_hll = None


def ref_hll_lib():
    global _ref_hll
    if _ref_hll is None:
        path = os.path.join(HERE, "_ref", "libref_hll.so")
        if not os.path.exists(path):
            if not os.path.isdir("/root/reference/include/kmerhash"):
                raise RuntimeError("reference HLL library not built and /root/reference absent")
            build(("ref",))
        L = C.CDLL(path)
        _decl_hll(L, "ref_hll_", [C.c_uint32, C.c_int, C.c_uint64])
        L.ref_serialize_pairs.argtypes = [_u64p, _u32p, C.c_uint64, C.c_char_p]
        L.ref_deserialize_pairs.restype = C.c_int64
        L.ref_deserialize_pairs.argtypes = [C.c_char_p, _u64p, _u32p, C.c_uint64]
        L.ref_serialize_u64.argtypes = [_u64p, C.c_uint64, C.c_char_p]
        _ref_hll = L
    return _ref_hll


def ref_hll_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_hll.so")) or os.path.isdir("/root/reference/include/kmerhash")


class RefHLL(_HLLBase):
    """the real fsc hyperloglog64<uint64_t, Hash, 12> from the reference tree"""
    _pfx = "ref_hll_"
    precision = 12

    def __init__(self, ignore_msb=0, hash_id=HASH_MURMUR3_X86, seed=43):
        self._L = ref_hll_lib()
        self.h = self._L.ref_hll_create(ignore_msb, hash_id, seed)


def ref_serialize_pairs(keys, vals, path):
    keys, vals = _k(keys), _v(vals)
    ref_hll_lib().ref_serialize_pairs(_p(keys, _u64p), _p(vals, _u32p), len(keys), path.encode())


def ref_deserialize_pairs(path, cap):
    k = np.zeros(cap, dtype=np.uint64)
    v = np.zeros(cap, dtype=np.uint32)
    n = ref_hll_lib().ref_deserialize_pairs(path.encode(), _p(k, _u64p), _p(v, _u32p), cap)
    if n < 0:
        raise RuntimeError("logic_error")
    return k[:n], v[:n]


def ref_serialize_u64(keys, path):
    keys = _k(keys)
    ref_hll_lib().ref_serialize_u64(_p(keys, _u64p), len(keys), path.encode())
