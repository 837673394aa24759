"""ctypes binding of include/kmerhash_amd.h (libkmerhash_amd.so).  No fallback: if the HIP library is
missing or no GPU is usable, importing works but every table constructor raises."""
import ctypes as C
import os

from .build import LIB

KH_OK, KH_ERR_INVALID, KH_ERR_NOMEM, KH_ERR_FULL, KH_ERR_PROBE_OVERFLOW, KH_ERR_HIP, KH_ERR_UNSUPPORTED, KH_ERR_RETRY = range(8)
KH_INS_REDUCE_PLUS, KH_INS_REPEATABLE = 1, 2
KH_KIND_ROBINHOOD, KH_KIND_LINEARPROBE = 0, 1
KH_HASH_IDENTITY, KH_HASH_MURMUR3_X86_128_LO64, KH_HASH_MURMUR3_X64_128_H0, KH_HASH_FARM64 = 0, 1, 2, 3
KH_MEM_HOST, KH_MEM_DEVICE = 0, 1
KH_XF_IDENTITY, KH_XF_DNA_LEX_LESS = 0, 1

STATUS_NAMES = {0: "KH_OK", 1: "KH_ERR_INVALID", 2: "KH_ERR_NOMEM", 3: "KH_ERR_FULL", 4: "KH_ERR_PROBE_OVERFLOW",
                5: "KH_ERR_HIP", 6: "KH_ERR_UNSUPPORTED", 7: "KH_ERR_RETRY"}

# every symbol include/kmerhash_amd.h declares (tests check the library exports each one)
SYMBOLS = [
    "kh_create", "kh_destroy", "kh_set_stream", "kh_set_key_transform", "kh_get_key_transform", "kh_hash_batch_transformed", "kh_shard_permute_transformed", "kh_last_error", "kh_size", "kh_capacity", "kh_get_load_thresholds",
    "kh_set_min_load_factor", "kh_set_max_load_factor", "kh_get_load_factors", "kh_clear", "kh_reserve", "kh_rehash",
    "kh_insert", "kh_insert_pairs", "kh_insert_one", "kh_update", "kh_insert_reduce_plus", "kh_insert_begin", "kh_insert_begin_ex", "kh_insert_feed", "kh_insert_end", "kh_insert_abort", "kh_count", "kh_find", "kh_find_compact", "kh_find_compact_pairs",
    "kh_erase", "kh_erase_one", "kh_to_vector", "kh_export_info", "kh_export_slots", "kh_export_raw_slots", "kh_displacement_histogram",
    "kh_hash_batch", "kh_shard_permute", "kh_shard_plan_create", "kh_shard_plan_permute", "kh_shard_plan_permute_global", "kh_shard_plan_offsets", "kh_shard_plan_destroy", "kh_profile_enable", "kh_profile_reset", "kh_profile_query", "kh_profile_dump",
    "kh_kmers_from_sequence", "kh_kmers_from_fastq", "kh_hll_create", "kh_hll_destroy", "kh_hll_set_stream", "kh_hll_update", "kh_hll_update_via_hashval",
    "kh_hll_merge", "kh_hll_clear", "kh_hll_registers", "kh_hll_estimate", "kh_hll_estimate_registers", "kh_release_cached_memory", "kh_version",
]

_lib = None
vp, u64, u32, i32, f32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float
pu64 = C.POINTER(C.c_uint64)


class KhError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), msg))
        self.status = status


class KhRetry(KhError):
    """kh_insert_end of a repeatable streamed insert: the speculative partition did not hold; feed the same pieces again"""


class KhLogicError(KhError):
    """mirrors std::logic_error thrown by the reference LP table (hashmap_linearprobe.hpp:408,503)"""


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise RuntimeError("libkmerhash_amd.so is not built (run `python -m kmerhash_amd.build` / __graft_entry__.build()); "
                           "there is no CPU fallback")
    L = C.CDLL(LIB)
    L.kh_version.restype = C.c_char_p
    L.kh_last_error.restype = C.c_char_p
    L.kh_last_error.argtypes = [vp]
    L.kh_create.argtypes = [C.POINTER(vp), i32, u32, u32, i32, u64, u64, f32, f32, i32]
    L.kh_destroy.argtypes = [vp]
    L.kh_set_stream.argtypes = [vp, vp]
    L.kh_set_key_transform.argtypes = [vp, i32, u32]
    L.kh_get_key_transform.argtypes = [vp, C.POINTER(i32), C.POINTER(u32)]
    L.kh_hash_batch_transformed.argtypes = [i32, u64, i32, u32, vp, u64, i32, vp, i32, vp]
    L.kh_shard_permute_transformed.argtypes = [i32, u64, i32, u32, u32, vp, vp, u64, vp, vp, vp, i32, vp]
    L.kh_size.argtypes = [vp, pu64]
    L.kh_capacity.argtypes = [vp, pu64]
    L.kh_get_load_thresholds.argtypes = [vp, pu64, pu64]
    L.kh_set_min_load_factor.argtypes = [vp, f32]
    L.kh_set_max_load_factor.argtypes = [vp, f32]
    L.kh_get_load_factors.argtypes = [vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
    L.kh_clear.argtypes = [vp]
    L.kh_reserve.argtypes = [vp, u64]
    L.kh_rehash.argtypes = [vp, u64]
    L.kh_insert.argtypes = [vp, vp, vp, u64, i32, pu64]
    L.kh_insert_pairs.argtypes = [vp, vp, u64, i32, pu64]
    L.kh_insert_one.argtypes = [vp, u64, u32, pu64]
    L.kh_update.argtypes = [vp, vp, vp, u64, i32, pu64]
    L.kh_insert_reduce_plus.argtypes = [vp, vp, vp, u64, i32, pu64]
    L.kh_insert_begin.argtypes = [vp, u64, i32]
    L.kh_insert_begin_ex.argtypes = [vp, u64, u32]
    L.kh_shard_plan_create.argtypes = [C.POINTER(vp), i32, u64, i32, u32, u32, vp, u64, u32, pu64, pu64, i32, vp]
    L.kh_shard_plan_permute.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    L.kh_shard_plan_permute_global.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    L.kh_shard_plan_offsets.argtypes = [vp, pu64]
    L.kh_shard_plan_destroy.argtypes = [vp]
    L.kh_shard_plan_destroy.restype = None
    L.kh_insert_feed.argtypes = [vp, vp, vp, u64, i32]
    L.kh_insert_end.argtypes = [vp, pu64]
    L.kh_insert_abort.argtypes = [vp]
    L.kh_count.argtypes = [vp, vp, u64, i32, vp]
    L.kh_find.argtypes = [vp, vp, u64, i32, vp, vp, pu64]
    L.kh_find_compact.argtypes = [vp, vp, u64, i32, vp, vp, pu64]
    L.kh_find_compact_pairs.argtypes = [vp, vp, u64, i32, vp, pu64]
    L.kh_erase.argtypes = [vp, vp, u64, i32, pu64]
    L.kh_erase_one.argtypes = [vp, u64, pu64]
    L.kh_to_vector.argtypes = [vp, vp, vp, pu64]
    L.kh_export_info.argtypes = [vp, vp]
    L.kh_export_slots.argtypes = [vp, vp, vp]
    L.kh_export_raw_slots.argtypes = [vp, vp]
    L.kh_displacement_histogram.argtypes = [vp, vp]
    L.kh_hash_batch.argtypes = [i32, u64, vp, u64, i32, vp, i32, vp]
    L.kh_shard_permute.argtypes = [i32, u64, u32, vp, vp, u64, vp, vp, vp, i32, vp]
    L.kh_release_cached_memory.argtypes = [i32]
    L.kh_kmers_from_sequence.argtypes = [vp, u64, u32, i32, i32, vp, pu64, i32, vp]
    L.kh_kmers_from_fastq.argtypes = [vp, u64, u32, i32, i32, vp, pu64, i32, vp]
    L.kh_hll_create.argtypes = [C.POINTER(vp), u32, u32, i32, u64, i32]
    L.kh_hll_destroy.argtypes = [vp]
    L.kh_hll_set_stream.argtypes = [vp, vp]
    L.kh_hll_update.argtypes = [vp, vp, u64, i32]
    L.kh_hll_update_via_hashval.argtypes = [vp, vp, u64, i32]
    L.kh_hll_merge.argtypes = [vp, vp]
    L.kh_hll_clear.argtypes = [vp]
    L.kh_hll_registers.argtypes = [vp, vp]
    L.kh_hll_estimate.argtypes = [vp, C.POINTER(C.c_double)]
    L.kh_hll_estimate_registers.argtypes = [vp, u32, C.POINTER(C.c_double)]
    L.kh_profile_enable.argtypes = [vp, i32]
    L.kh_profile_reset.argtypes = [vp]
    L.kh_profile_query.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), pu64]
    L.kh_profile_dump.argtypes = [vp, C.c_char_p, u64]
    for s in SYMBOLS:
        if s not in ("kh_version", "kh_last_error"):
            getattr(L, s).restype = i32
    _lib = L
    return L
