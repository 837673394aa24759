"""Host-side mirror of the reference's table interface over the C-ABI (include/kmerhash_amd.h).

Class and member names follow the reference (fsc::hashmap_robinhood_doubling, hashmap_robinhood.hpp:124-126;
fsc::hashmap_linearprobe_doubling, hashmap_linearprobe.hpp:96-98): insert / find / count / erase / update /
size / capacity / reserve / rehash / clear / to_vector / keys / set_{min,max}_load_factor.

Batches are numpy arrays (host memory, copied by the library) or torch CUDA tensors (device memory, zero
copy; int64/uint64 keys, int32/uint32 values).  Results come back in the same kind of container.
torch is plumbing only (device buffers + the current stream); all work happens in libkmerhash_amd.so.
There is no CPU fallback: constructing a table without the HIP library or without a GPU raises.
"""
import ctypes as C

import numpy as np

from . import _capi as K
from ._capi import (KH_HASH_FARM64, KH_HASH_IDENTITY, KH_HASH_MURMUR3_X64_128_H0, KH_HASH_MURMUR3_X86_128_LO64,
                    KH_KIND_LINEARPROBE, KH_KIND_ROBINHOOD, KhError, KhLogicError, KhRetry)

try:  # torch is optional for host-array use
    import torch
except Exception:  # pragma: no cover
    torch = None

HASHES = {
    "identity": KH_HASH_IDENTITY,
    "murmur3avx64": KH_HASH_MURMUR3_X86_128_LO64,   # fsc::hash::murmur3avx64 == murmur_x86
    "murmur_x86": KH_HASH_MURMUR3_X86_128_LO64,
    "murmur": KH_HASH_MURMUR3_X64_128_H0,           # fsc::hash::murmur (x64_128, h[0])
    "farm": KH_HASH_FARM64,
}


def _is_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _hash_id(h):
    return HASHES[h] if isinstance(h, str) else int(h)


class _Buf:
    """pointer + memory kind of one batch argument"""

    def __init__(self, x, np_dtype, itemsize):
        if _is_tensor(x):
            if not x.is_cuda:
                x = x.cpu().numpy()
            else:
                if x.element_size() != itemsize:
                    raise TypeError("tensor element size %d, expected %d" % (x.element_size(), itemsize))
                self.obj = x.contiguous()
                self.ptr = self.obj.data_ptr()
                self.n = self.obj.numel()
                self.where = K.KH_MEM_DEVICE
                self.device = self.obj.device
                return
        a = np.ascontiguousarray(x, dtype=np_dtype)
        self.obj = a
        self.ptr = a.ctypes.data
        self.n = a.size
        self.where = K.KH_MEM_HOST
        self.device = None


class _HashMapBase:
    KIND = None
    DEFAULT_MIN_LF = None
    DEFAULT_MAX_LF = None

    def __init__(self, capacity=128, min_load_factor=None, max_load_factor=None, hash="murmur3avx64", seed=43, device=0):
        self._L = K.lib()
        self._h = C.c_void_p()
        self.device = int(device)
        mn = self.DEFAULT_MIN_LF if min_load_factor is None else min_load_factor
        mx = self.DEFAULT_MAX_LF if max_load_factor is None else max_load_factor
        st = self._L.kh_create(C.byref(self._h), self.KIND, 8, 4, _hash_id(hash), seed, capacity, mn, mx, self.device)
        if st != K.KH_OK:
            self._h = C.c_void_p()
            raise KhError(st, "kh_create failed (is a GPU visible and the HIP library built?)")

    # -- plumbing --------------------------------------------------------------------------------
    def _chk(self, st):
        if st == K.KH_OK:
            return
        msg = self._L.kh_last_error(self._h).decode()
        if st == K.KH_ERR_FULL:
            raise KhLogicError(st, msg)
        if st == K.KH_ERR_RETRY:
            raise KhRetry(st, msg)
        raise KhError(st, msg)

    def _sync_stream(self, *bufs):
        """issue the table's work on torch's current stream when device tensors are involved"""
        if torch is not None and any(b is not None and b.where == K.KH_MEM_DEVICE for b in bufs):
            s = torch.cuda.current_stream(self.device).cuda_stream
            self._L.kh_set_stream(self._h, C.c_void_p(s))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.kh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out(self, like, n, np_dtype, torch_dtype):
        if like.where == K.KH_MEM_DEVICE:
            t = torch.empty(n, dtype=torch_dtype, device=like.device)
            return t, t.data_ptr()
        a = np.zeros(n, dtype=np_dtype)
        return a, a.ctypes.data

    # -- scalar state ------------------------------------------------------------------------------
    def size(self):
        v = C.c_uint64()
        self._chk(self._L.kh_size(self._h, C.byref(v)))
        return v.value

    def __len__(self):
        return self.size()

    def capacity(self):
        v = C.c_uint64()
        self._chk(self._L.kh_capacity(self._h, C.byref(v)))
        return v.value

    def load_thresholds(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self._L.kh_get_load_thresholds(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_min_load_factor(self, f):
        self._chk(self._L.kh_set_min_load_factor(self._h, f))

    def set_max_load_factor(self, f):
        self._chk(self._L.kh_set_max_load_factor(self._h, f))

    def get_load_factor(self):
        c = C.c_float()
        self._chk(self._L.kh_get_load_factors(self._h, None, None, C.byref(c)))
        return c.value

    def get_min_load_factor(self):
        c = C.c_float()
        self._chk(self._L.kh_get_load_factors(self._h, C.byref(c), None, None))
        return c.value

    def get_max_load_factor(self):
        c = C.c_float()
        self._chk(self._L.kh_get_load_factors(self._h, None, C.byref(c), None))
        return c.value

    def set_key_transform(self, k):
        """PreTransform of fsc::TransformedHash / TransformedComparator (hash_new.hpp:387-1134): k = 0 identity, k = 1..32
        bliss::kmer::transform::lex_less on 2-bit packed DNA k-mers -- a k-mer and its reverse complement are one key
        ("bimolecule" tables); the bits stored are those of the first occurrence.  Only on an empty table."""
        self._chk(self._L.kh_set_key_transform(self._h, K.KH_XF_DNA_LEX_LESS if k else K.KH_XF_IDENTITY, int(k)))

    def clear(self):
        self._chk(self._L.kh_clear(self._h))

    def reserve(self, n):
        self._chk(self._L.kh_reserve(self._h, int(n)))

    def rehash(self, b):
        self._chk(self._L.kh_rehash(self._h, int(b)))

    # -- batch operations ------------------------------------------------------------------------------
    def insert(self, keys, vals=None):
        """insert(Iter,Iter) / insert(vector const&): first value wins.  Returns #inserted.
        `keys` may also be an (n,2)-shaped uint64 pair array laid out like std::pair<uint64_t,uint32_t>."""
        if vals is None:
            kb = _Buf(keys, np.uint64, 8)
            self._sync_stream(kb)
            n = kb.n // 2
            out = C.c_uint64()
            self._chk(self._L.kh_insert_pairs(self._h, kb.ptr, n, kb.where, C.byref(out)))
            return out.value
        kb, vb = _Buf(keys, np.uint64, 8), _Buf(vals, np.uint32, 4)
        if kb.n != vb.n or kb.where != vb.where:
            raise ValueError("keys/vals must have equal length and live in the same memory space")
        self._sync_stream(kb, vb)
        out = C.c_uint64()
        self._chk(self._L.kh_insert(self._h, kb.ptr, vb.ptr, kb.n, kb.where, C.byref(out)))
        return out.value

    def insert_one(self, key, val):
        """insert(value_type const&): the single-key form -- no trailing reserve(size()) (hashmap_robinhood.hpp:522-626)."""
        out = C.c_uint64()
        self._chk(self._L.kh_insert_one(self._h, int(key), int(val), C.byref(out)))
        return out.value

    def update(self, keys, vals):
        """update(k,v) applied in batch order: insert, or overwrite with the last value given."""
        kb, vb = _Buf(keys, np.uint64, 8), _Buf(vals, np.uint32, 4)
        if kb.n != vb.n or kb.where != vb.where:
            raise ValueError("keys/vals must have equal length and live in the same memory space")
        self._sync_stream(kb, vb)
        out = C.c_uint64()
        self._chk(self._L.kh_update(self._h, kb.ptr, vb.ptr, kb.n, kb.where, C.byref(out)))
        return out.value

    # -- streamed insert: one insert whose pairs arrive in pieces (multi-GPU exchange) ----------------------------
    def insert_begin(self, n_total, reduce_plus=False, repeatable=False):
        """repeatable: the caller keeps every piece until insert_end has returned and feeds them again (without this flag) if
        insert_end raises KhRetry -- allows the histogram-free partition of the pieces (kh_insert_begin_ex)"""
        flags = (K.KH_INS_REDUCE_PLUS if reduce_plus else 0) | (K.KH_INS_REPEATABLE if repeatable else 0)
        self._chk(self._L.kh_insert_begin_ex(self._h, int(n_total), flags))

    def insert_feed(self, keys, vals=None):
        """partition this piece now (asynchronous for device tensors); the pieces count as one batch in feed order"""
        kb = _Buf(keys, np.uint64, 8)
        vb = _Buf(vals, np.uint32, 4) if vals is not None else None
        if vb is not None and (kb.n != vb.n or kb.where != vb.where):
            raise ValueError("keys/vals must have equal length and live in the same memory space")
        self._sync_stream(kb, vb)
        self._chk(self._L.kh_insert_feed(self._h, kb.ptr, vb.ptr if vb is not None else None, kb.n, kb.where))

    def insert_end(self):
        out = C.c_uint64()
        self._chk(self._L.kh_insert_end(self._h, C.byref(out)))
        return out.value

    def insert_abort(self):
        """gives up a streamed insert: the pieces fed so far are dropped, the table is unchanged and usable again"""
        self._chk(self._L.kh_insert_abort(self._h))

    def insert_reduce_plus(self, keys, vals=None):
        """Reducer = std::plus (k-mer counting when vals is None: every occurrence adds 1).  Returns #new keys.
        reference: hashmap_robinhood_offsets_reduction::insert(keys, T(1)), counting_batched_robinhood_map."""
        kb = _Buf(keys, np.uint64, 8)
        vb = _Buf(vals, np.uint32, 4) if vals is not None else None
        if vb is not None and (kb.n != vb.n or kb.where != vb.where):
            raise ValueError("keys/vals must have equal length and live in the same memory space")
        self._sync_stream(kb, vb)
        out = C.c_uint64()
        self._chk(self._L.kh_insert_reduce_plus(self._h, kb.ptr, vb.ptr if vb is not None else None, kb.n, kb.where, C.byref(out)))
        return out.value

    def count(self, keys, out=None):
        """count(Iter,Iter): 0/1 per query, query order (uint8).  out: a contiguous uint8 CUDA tensor of the queries' length to write
        into (device queries only; nothing is allocated or copied then)."""
        kb = _Buf(keys, np.uint64, 8)
        self._sync_stream(kb)
        if out is not None and kb.where == K.KH_MEM_DEVICE:
            assert out.is_contiguous() and out.numel() == kb.n and out.element_size() == 1
            optr = out.data_ptr()
        else:
            out, optr = self._out(kb, kb.n, np.uint8, torch.uint8 if torch else None)
        self._chk(self._L.kh_count(self._h, kb.ptr, kb.n, kb.where, optr))
        return out

    def find_values(self, keys, out_vals=None, out_found=None, want_total=True):
        """per-query form: (values, found) aligned with the queries (values of misses are 0).  out_vals / out_found: contiguous
        int32 / uint8 CUDA tensors of the queries' length to write into (device queries only; out_vals must hold 0 where a miss is to
        read 0: the library leaves the values of misses untouched).  want_total=False: the call does not wait for the device."""
        kb = _Buf(keys, np.uint64, 8)
        self._sync_stream(kb)
        if kb.where == K.KH_MEM_DEVICE:
            vals = out_vals if out_vals is not None else torch.zeros(kb.n, dtype=torch.int32, device=kb.device)
            found = out_found if out_found is not None else torch.empty(kb.n, dtype=torch.uint8, device=kb.device)
            assert vals.is_contiguous() and found.is_contiguous() and vals.numel() == kb.n and found.numel() == kb.n
            vptr, fptr = vals.data_ptr(), found.data_ptr()
        else:
            vals = np.zeros(kb.n, dtype=np.uint32)
            found = np.zeros(kb.n, dtype=np.uint8)
            vptr, fptr = vals.ctypes.data, found.ctypes.data
        nf = C.c_uint64()
        self._chk(self._L.kh_find(self._h, kb.ptr, kb.n, kb.where, vptr, fptr, C.byref(nf) if want_total else None))
        return vals, found

    def find(self, keys):
        """find(Iter,Iter): the (key, value) pairs of the hits only, in query order -> (keys, vals)."""
        kb = _Buf(keys, np.uint64, 8)
        self._sync_stream(kb)
        if kb.where == K.KH_MEM_DEVICE:
            ok = torch.empty(kb.n, dtype=torch.int64, device=kb.device)
            ov = torch.empty(kb.n, dtype=torch.int32, device=kb.device)
            kptr, vptr = ok.data_ptr(), ov.data_ptr()
        else:
            ok = np.zeros(kb.n, dtype=np.uint64)
            ov = np.zeros(kb.n, dtype=np.uint32)
            kptr, vptr = ok.ctypes.data, ov.ctypes.data
        nf = C.c_uint64()
        self._chk(self._L.kh_find_compact(self._h, kb.ptr, kb.n, kb.where, kptr, vptr, C.byref(nf)))
        return ok[: nf.value], ov[: nf.value]

    def erase(self, keys):
        """erase(Iter,Iter): returns #erased (RH never shrinks here; LP may, as in the reference)."""
        kb = _Buf(keys, np.uint64, 8)
        self._sync_stream(kb)
        out = C.c_uint64()
        self._chk(self._L.kh_erase(self._h, kb.ptr, kb.n, kb.where, C.byref(out)))
        return out.value

    def erase_one(self, key):
        """erase(key): single-key form, halves the table when size < min_load."""
        out = C.c_uint64()
        self._chk(self._L.kh_erase_one(self._h, int(key), C.byref(out)))
        return out.value

    # -- iteration / exports -----------------------------------------------------------------------------
    def to_vector(self):
        n = self.size()
        k = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1), dtype=np.uint32)
        m = C.c_uint64()
        self._chk(self._L.kh_to_vector(self._h, k.ctypes.data, v.ctypes.data, C.byref(m)))
        assert m.value == n, (m.value, n)
        return k[:n], v[:n]

    def keys(self):
        return self.to_vector()[0]

    def sorted_items(self):
        k, v = self.to_vector()
        o = np.argsort(k, kind="stable")
        return k[o], v[o]

    def export_info(self):
        out = np.zeros(self.capacity(), dtype=np.uint8)
        self._chk(self._L.kh_export_info(self._h, out.ctypes.data))
        return out

    def export_slots(self):
        k = np.zeros(self.capacity(), dtype=np.uint64)
        v = np.zeros(self.capacity(), dtype=np.uint32)
        self._chk(self._L.kh_export_slots(self._h, k.ctypes.data, v.ctypes.data))
        return k, v

    def displacement_histogram(self):
        out = np.zeros(128, dtype=np.uint64)
        self._chk(self._L.kh_displacement_histogram(self._h, out.ctypes.data))
        return out

    # -- measurement ---------------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self._L.kh_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self._L.kh_profile_reset(self._h))

    def profile(self):
        """{kernel name: (launches, total_ms)} measured with HIP events on the table's stream"""
        buf = C.create_string_buffer(1 << 16)
        self._chk(self._L.kh_profile_dump(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out


class hashmap_robinhood_doubling(_HashMapBase):
    """fsc::hashmap_robinhood_doubling<uint64_t, uint32_t, Hash> (defaults hashmap_robinhood.hpp:218-220)"""
    KIND = KH_KIND_ROBINHOOD
    DEFAULT_MIN_LF = 0.4
    DEFAULT_MAX_LF = 0.9

    # insert_integrated / insert_sort / insert_shuffled are the same algorithm in the reference
    # (hashmap_robinhood.hpp:721-836,843,1002) minus the trailing reserve(), which is a no-op.
    def insert_integrated(self, keys, vals=None):
        return self.insert(keys, vals)


class hashmap_linearprobe_doubling(_HashMapBase):
    """fsc::hashmap_linearprobe_doubling<uint64_t, uint32_t, Hash> (defaults hashmap_linearprobe.hpp:191-193)"""
    KIND = KH_KIND_LINEARPROBE
    DEFAULT_MIN_LF = 0.2
    DEFAULT_MAX_LF = 0.6


def hash_batch(keys, hash="murmur3avx64", seed=43, device=0, lex_less_k=0):
    """Hash::operator()(Key const*, count, out): batched 64-bit hashing on the GPU; lex_less_k = k: TransformedHash with the
    lex_less pre-transform on 2-bit packed DNA k-mers (hash of min(k-mer, reverse complement))."""
    L = K.lib()
    kb = _Buf(keys, np.uint64, 8)
    if kb.where == K.KH_MEM_DEVICE:
        out = torch.empty(kb.n, dtype=torch.int64, device=kb.device)
        optr = out.data_ptr()
        stream = torch.cuda.current_stream(device).cuda_stream
    else:
        out = np.zeros(kb.n, dtype=np.uint64)
        optr = out.ctypes.data
        stream = None
    st = L.kh_hash_batch_transformed(_hash_id(hash), seed, K.KH_XF_DNA_LEX_LESS if lex_less_k else K.KH_XF_IDENTITY, int(lex_less_k),
                                     kb.ptr, kb.n, kb.where, optr, device, stream)
    if st != K.KH_OK:
        raise KhError(st, "kh_hash_batch_transformed")
    return out
