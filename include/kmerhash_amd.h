/* kmerhash_amd.h -- C-ABI of libkmerhash_amd.so: MI355X (gfx950) open-addressing k-mer hash tables.
 *
 * This is the drop-in boundary for ONE hot path of ParBLiSS/kmerhash: the batched
 * insert / find / count / erase of
 *     fsc::hashmap_robinhood_doubling   (reference include/kmerhash/hashmap_robinhood.hpp:124-126)
 *     fsc::hashmap_linearprobe_doubling (reference include/kmerhash/hashmap_linearprobe.hpp:96-98)
 * for 64-bit keys (2-bit packed k-mers, k <= 32) with 32-bit mapped values, and of the 64-bit hash
 * functors they are instantiated with.  Plain pointers and sizes only; no C++/torch types cross it.
 * The C++ template shim (include/kmerhash_amd/hashmap.hpp) and the Python host layer (kmerhash_amd/) are
 * thin callers of exactly these entry points; INTEGRATION.md shows the reference-side binding.
 *
 * Every function returns a kh_status; kh_last_error() gives the text of the last failure on a table.
 * A table is single-writer (like the reference: not thread safe); read-only batches may not overlap
 * a mutating batch.  All device work of a table is issued on its stream (kh_set_stream) and the
 * functions that return scalar results (n_inserted, n_found, ...) synchronise that stream.
 *
 * Pointer arguments marked [h|d] live in host or device memory as told by the kh_mem argument;
 * outputs live in the same space as the inputs of that call.  Device pointers must be valid on the
 * table's device.
 */
#ifndef KMERHASH_AMD_H_
#define KMERHASH_AMD_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kh_table kh_table;

typedef enum {
  KH_OK = 0,
  KH_ERR_INVALID = 1,        /* bad argument / unsupported key or value width */
  KH_ERR_NOMEM = 2,          /* device allocation failed */
  KH_ERR_FULL = 3,           /* LP: no slot left -- mirrors std::logic_error, hashmap_linearprobe.hpp:408,503 */
  KH_ERR_PROBE_OVERFLOW = 4, /* RH probe distance would reach 128: the reference asserts (hashmap_robinhood.hpp:556)
                                or silently corrupts under -DNDEBUG; we refuse and leave the table unchanged */
  KH_ERR_HIP = 5,            /* a HIP runtime call failed (no GPU, launch failure, ...) */
  KH_ERR_UNSUPPORTED = 6,
  KH_ERR_RETRY = 7           /* kh_insert_end of a KH_INS_REPEATABLE streamed insert: its speculative partition did not hold for this
                                batch; nothing was inserted; feed the same pieces again after kh_insert_begin without that flag */
} kh_status;

typedef enum {
  KH_KIND_ROBINHOOD = 0,     /* fsc::hashmap_robinhood_doubling  : info 0x00 empty, 0x80|dist occupied (:137-163) */
  KH_KIND_LINEARPROBE = 1    /* fsc::hashmap_linearprobe_doubling: info 0x40 empty, 0x80 deleted, 0x00 normal (:109-139) */
} kh_kind;

typedef enum {
  KH_HASH_IDENTITY = 0,             /* fsc::hash::identity<T>      hash_new.hpp:135-166 */
  KH_HASH_MURMUR3_X86_128_LO64 = 1, /* fsc::hash::murmur3avx64<T>  murmurhash3_64_avx.hpp:1553-1651 (== murmur_x86, hash_new.hpp:218) */
  KH_HASH_MURMUR3_X64_128_H0 = 2,   /* fsc::hash::murmur<T>        hash_new.hpp:206-235 */
  KH_HASH_FARM64 = 3                /* fsc::hash::farm<T>          hash_new.hpp:309-328 (parity unpinned) */
} kh_hash;

typedef enum { KH_MEM_HOST = 0, KH_MEM_DEVICE = 1 } kh_mem;

/* Key transform in front of the hash and inside key equality: fsc::TransformedHash<Key, Hash, PreTransform> together with
 * fsc::TransformedComparator<Key, std::equal_to, PreTransform> (hash_new.hpp:387-1134; used with
 * bliss::kmer::transform::lex_less for "bimolecule" tables in which a k-mer and its reverse complement are one key,
 * test/unit/test_hashmap_robinhood_doubling.cpp:560-626).  The table stores the key bits of the FIRST occurrence, as the
 * reference does; find / to_vector return the stored bits. */
typedef enum {
  KH_XF_IDENTITY = 0,      /* bliss::transform::identity */
  KH_XF_DNA_LEX_LESS = 1   /* bliss::kmer::transform::lex_less on a 2-bit packed DNA k-mer (first base most significant, A0 C1 G2 T3):
                              the key stands for min(key, reverse complement).  kmerind's own packing is not part of the reference
                              tree: PARITY UNPINNED for the bit layout */
} kh_key_transform;

/* ---- lifetime: ctor (capacity=128, min_lf, max_lf)  hashmap_robinhood.hpp:218-233 / hashmap_linearprobe.hpp:191-206 */
kh_status kh_create(kh_table** out, kh_kind kind, uint32_t key_bytes /*8*/, uint32_t val_bytes /*4*/,
                    kh_hash hash, uint64_t seed /*43*/, uint64_t capacity /*128*/,
                    float min_load_factor, float max_load_factor, int device);
kh_status kh_destroy(kh_table* t);
kh_status kh_set_stream(kh_table* t, void* hip_stream /* hipStream_t; NULL = default stream */);
/* the PreTransform of the table's TransformedHash / TransformedComparator; k = k-mer length (1..32) for KH_XF_DNA_LEX_LESS.
 * Only on an empty table (the reference fixes it at compile time). */
kh_status kh_set_key_transform(kh_table* t, kh_key_transform xf, uint32_t k);
kh_status kh_get_key_transform(const kh_table* t, kh_key_transform* xf, uint32_t* k);
const char* kh_last_error(const kh_table* t);

/* ---- scalar state: size() :406 / capacity() :287 / load factors :261-285 / clear :413 / reserve :421 / rehash :432 */
kh_status kh_size(const kh_table* t, uint64_t* out);
kh_status kh_capacity(const kh_table* t, uint64_t* out);
kh_status kh_get_load_thresholds(const kh_table* t, uint64_t* min_load, uint64_t* max_load);
kh_status kh_set_min_load_factor(kh_table* t, float f);
kh_status kh_set_max_load_factor(kh_table* t, float f);
kh_status kh_get_load_factors(const kh_table* t, float* min_lf, float* max_lf, float* current);
kh_status kh_clear(kh_table* t);
kh_status kh_reserve(kh_table* t, uint64_t n);
kh_status kh_rehash(kh_table* t, uint64_t buckets);

/* ---- batch insert == insert(Iter,Iter) / insert(vector const&)  hashmap_robinhood.hpp:633-717,
 *      hashmap_linearprobe.hpp:521-573.  First value wins; capacity follows the reference's doubling
 *      rule exactly (one doubling per insert call made while size >= max_load, duplicates included). */
kh_status kh_insert(kh_table* t, const void* keys /*[h|d] u64[n]*/, const void* vals /*[h|d] u32[n]*/,
                    uint64_t n, kh_mem where, uint64_t* n_inserted);
/* same, input as the reference's std::pair<uint64_t,uint32_t> array (16 B: key @0, value @8) */
kh_status kh_insert_pairs(kh_table* t, const void* pairs16 /*[h|d]*/, uint64_t n, kh_mem where, uint64_t* n_inserted);
/* insert(value_type const&) / insert(key, val): the single-key form (hashmap_robinhood.hpp:522-626, hashmap_linearprobe.hpp:430-515).
 *      Unlike the batch forms it is NOT followed by reserve(size()): the two differ after set_max_load_factor() lowered the
 *      threshold below the current size. */
kh_status kh_insert_one(kh_table* t, uint64_t key, uint32_t val, uint64_t* n_inserted /* 0 or 1 */);
/* update(k,v) applied in order to a batch: insert, or overwrite the existing value (last one wins)
 *      hashmap_robinhood.hpp:1274-1284 / hashmap_linearprobe.hpp:895-905 */
kh_status kh_update(kh_table* t, const void* keys, const void* vals, uint64_t n, kh_mem where, uint64_t* n_inserted);

/* ---- streamed insert: ONE insert(Iter,Iter) whose pairs arrive in pieces (the multi-GPU exchange delivers them peer by
 *      peer / chunk by chunk: khmxx::ialltoallv_and_modify, incremental_mxx.hpp:3437-3645, calls insert_no_estimate per
 *      block).  kh_insert_begin announces the exact total; every kh_insert_feed radix-partitions its piece at once and
 *      returns without synchronising (device pointers), so that work overlaps the next transfer; kh_insert_end de-duplicates
 *      and builds once.  The result equals kh_insert of the concatenated pieces in feed order (first value wins, same
 *      capacity rule).  reduce_plus != 0: kh_insert_reduce_plus semantics (vals may be NULL).  At most 16 feeds.  Between begin and
 *      end every other call that mutates the table or uses its workspace (insert, update, erase, rehash, reserve, clear, find,
 *      count, to_vector, displacement_histogram) returns KH_ERR_INVALID.  Host buffers (KH_MEM_HOST) may be reused as soon as
 *      kh_insert_feed returns; device buffers must stay valid until the work queued on the table's stream has consumed them
 *      (kh_insert_end synchronises). */
kh_status kh_insert_begin(kh_table* t, uint64_t n_total, int reduce_plus);
/* (same reference interface as kh_insert_begin: insert_no_estimate per received block, incremental_mxx.hpp:3437-3645)
 * flags: KH_INS_REDUCE_PLUS = kh_insert_begin's reduce_plus.  KH_INS_REPEATABLE: the caller keeps every piece it feeds (valid and
 *      unchanged) until kh_insert_end has returned and can feed them again.  The library may then partition the pieces without a
 *      histogram pass into slots they share (and without stream positions when a sample of the first piece shows no duplicate key);
 *      if that does not hold for the batch -- skewed or duplicated keys -- kh_insert_end returns KH_ERR_RETRY with the table
 *      unchanged, and the caller repeats begin (without the flag) / feed / end.  (The multi-GPU layer keeps its receive buffers.) */
#define KH_INS_REDUCE_PLUS 1u
#define KH_INS_REPEATABLE 2u
kh_status kh_insert_begin_ex(kh_table* t, uint64_t n_total, unsigned flags);
kh_status kh_insert_feed(kh_table* t, const void* keys /*[h|d] u64[n]*/, const void* vals /*[h|d] u32[n]*/, uint64_t n, kh_mem where);
kh_status kh_insert_end(kh_table* t, uint64_t* n_inserted);
/* gives up a streamed insert after kh_insert_begin: the pieces fed so far are dropped, nothing is inserted and the table is usable
 * again (the feeds only write workspace).  What the reference does when an exception leaves ialltoallv_and_modify's block loop
 * (incremental_mxx.hpp:3437-3645): the partially received batch is simply not inserted.  Synchronises the table's stream (queued
 * partition kernels may still read the caller's buffers).  No-op without a streamed insert in progress. */
kh_status kh_insert_abort(kh_table* t);

/* ---- reducer insert: the Reducer = std::plus form of the reference's batched table,
 *      hashmap_robinhood_offsets_reduction::insert(keys, T(1)) / insert(pairs) (robinhood_offset_hashmap_ptr.hpp:85-97,
 *      2787-2885) as used by dsc::counting_batched_robinhood_map (distributed_batched_robinhood_map.hpp:2542-2543,2633,2899):
 *      the value of a key becomes the (wrapping 32-bit) sum of the values of all its occurrences; vals == NULL means
 *      every occurrence contributes 1 (k-mer counting).  Results only are specified by the reference here (its own
 *      container sizes itself from a HyperLogLog estimate); capacity follows this table's doubling rule. */
kh_status kh_insert_reduce_plus(kh_table* t, const void* keys /*[h|d] u64[n]*/, const void* vals /*[h|d] u32[n] or NULL*/,
                                uint64_t n, kh_mem where, uint64_t* n_inserted);

/* ---- count(Iter,Iter): 0/1 per query in query order  hashmap_robinhood.hpp:1111-1160 / hashmap_linearprobe.hpp:639-688
 *      (the reference returns vector<size_t>; one byte per query here) */
kh_status kh_count(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint8_t* out01 /*[h|d] u8[n]*/);

/* ---- find: per-query form (value + found flag) and the reference's compacted form
 *      find(Iter,Iter) -> vector<pair> of hits in query order  hashmap_robinhood.hpp:1194-1268 / hashmap_linearprobe.hpp:816-889 */
kh_status kh_find(kh_table* t, const void* keys, uint64_t n, kh_mem where,
                  uint32_t* out_vals /*[h|d] u32[n], untouched on miss*/, uint8_t* out_found /*[h|d] u8[n]*/, uint64_t* n_found);
kh_status kh_find_compact(kh_table* t, const void* keys, uint64_t n, kh_mem where,
                          uint64_t* out_keys /*[h|d] u64[n]*/, uint32_t* out_vals /*[h|d] u32[n]*/, uint64_t* n_found);
kh_status kh_find_compact_pairs(kh_table* t, const void* keys, uint64_t n, kh_mem where,
                                void* out_pairs16 /*[h|d] 16 B x n*/, uint64_t* n_found);

/* ---- erase(Iter,Iter)  hashmap_robinhood.hpp:1430-1440 (never shrinks) / hashmap_linearprobe.hpp:1042-1051 (may shrink) */
kh_status kh_erase(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint64_t* n_erased);
/* erase(key) single-key form: also halves the table when size < min_load (:1421-1428 / :1032-1039) */
kh_status kh_erase_one(kh_table* t, uint64_t key, uint64_t* n_erased);

/* ---- iteration / parity exports (host buffers) */
kh_status kh_to_vector(kh_table* t, uint64_t* keys_host, uint32_t* vals_host, uint64_t* n_out); /* to_vector() :388, slot order */
kh_status kh_export_info(kh_table* t, uint8_t* out_host /* capacity bytes, reference encoding of the table's kind */);
kh_status kh_export_slots(kh_table* t, uint64_t* keys_host, uint32_t* vals_host /* capacity entries; empty slots unspecified */);
/* the table as it lies: capacity entries of 16 bytes {u64 key, u32 value, u32 info}; the low byte of `info` is the reference's
 * info byte of the table's kind (hashmap_robinhood.hpp:137-163 / hashmap_linearprobe.hpp:109-139), key and value of an empty slot are
 * unspecified.  What the reference exposes as `container` + `info_container` to its iterators (hashmap_robinhood.hpp:295-309); the
 * C++ shim probes such a snapshot on the host for loops of single-key const calls (find(key) / count(key), :1102,:1165). */
kh_status kh_export_raw_slots(kh_table* t, void* out_host /* capacity x 16 B */);
kh_status kh_displacement_histogram(kh_table* t, uint64_t out[128]); /* RH only: #slots per probe distance (REPROBE_STAT) */

/* ---- batched hashing: Hash::operator()(Key const*, count, out)  murmurhash3_64_avx.hpp:1584-1597, hash_new.hpp:1035-1056 */
kh_status kh_hash_batch(kh_hash hash, uint64_t seed, const void* keys, uint64_t n, kh_mem where,
                        uint64_t* out /*[h|d]*/, int device, void* hip_stream);
/* TransformedHash::operator()(Key const*, count, out)  hash_new.hpp:1035-1056: out[i] = hash(pre_transform(keys[i])) */
kh_status kh_hash_batch_transformed(kh_hash hash, uint64_t seed, kh_key_transform xf, uint32_t k, const void* keys, uint64_t n,
                                    kh_mem where, uint64_t* out /*[h|d]*/, int device, void* hip_stream);

/* ---- key-space sharding for the multi-GPU layer: rank = hash(key, seed) & (p-1) (p power of two) or % p
 *      (distributed_batched_robinhood_map.hpp:513-534,632-741 assign_count_permute).  Device buffers only.
 *      out_* receive the pairs grouped by destination rank (rank 0 first, input order kept inside a rank);
 *      counts_host[p] receives the per-rank element counts.  out_keys_dev == NULL: count only (nothing is permuted). */
kh_status kh_shard_permute(kh_hash hash, uint64_t seed, uint32_t nranks,
                           const uint64_t* keys_dev, const uint32_t* vals_dev /* may be NULL */, uint64_t n,
                           uint64_t* out_keys_dev /* may be NULL */, uint32_t* out_vals_dev /* may be NULL */,
                           uint64_t* counts_host, int device, void* hip_stream);
/* the same with the distributed map's TransformedHash (rank = hash(pre_transform(key)) mod p: both strands of a k-mer go to one rank) */
kh_status kh_shard_permute_transformed(kh_hash hash, uint64_t seed, kh_key_transform xf, uint32_t k, uint32_t nranks,
                                       const uint64_t* keys_dev, const uint32_t* vals_dev, uint64_t n,
                                       uint64_t* out_keys_dev, uint32_t* out_vals_dev, uint64_t* counts_host, int device, void* hip_stream);

/* ---- a batch that will be exchanged in `pieces` pieces (the pipelined insert, khmxx::ialltoallv_and_modify incremental_mxx.hpp:3437-3645):
 *      the counting half of assign_count_permute (distributed_batched_robinhood_map.hpp:632-741) done ONCE -- one count sweep + scan +
 *      host synchronisation for the whole batch.  bounds_host[pieces+1] receives the piece boundaries (multiples of 4096 pairs, the last one = n),
 *      counts_host[pieces][nranks] the destination counts of every piece; kh_shard_plan_permute then permutes piece i (pairs
 *      [bounds[i], bounds[i+1]) of the SAME keys/vals arrays, unchanged since the plan was made) into out_* grouped by rank, input
 *      order kept, without counting again and without synchronising.  nranks <= 8.  Same result as kh_shard_permute on the piece. */
typedef struct kh_shard_plan kh_shard_plan;
kh_status kh_shard_plan_create(kh_shard_plan** out, kh_hash hash, uint64_t seed, kh_key_transform xf, uint32_t k, uint32_t nranks,
                               const uint64_t* keys_dev, uint64_t n, uint32_t pieces, uint64_t* counts_host, uint64_t* bounds_host,
                               int device, void* hip_stream);
kh_status kh_shard_plan_permute(kh_shard_plan* plan, uint32_t piece, const uint64_t* keys_dev, const uint32_t* vals_dev /* may be NULL */,
                                uint64_t* out_keys_dev, uint32_t* out_vals_dev, void* hip_stream);
/* (same reference interface: the permutation half of assign_count_permute, distributed_batched_robinhood_map.hpp:632-741, as the
 * pipelined queries khmxx::ialltoallv_and_query_one_to_one use it, incremental_mxx.hpp:4403-4669)  piece i written to ITS PLACE in the
 * layout of the whole batch grouped by rank -- out_* have room for all n pairs and end up, once every piece has been permuted, equal to
 * kh_shard_permute's output; the part of rank r that piece i contributes is contiguous: it starts at offsets[r][i] and ends at
 * offsets[r][i + 1] (kh_shard_plan_offsets: [nranks][pieces + 1] positions in that layout). */
kh_status kh_shard_plan_permute_global(kh_shard_plan* plan, uint32_t piece, const uint64_t* keys_dev, const uint32_t* vals_dev /* may be NULL */,
                                       uint64_t* out_keys_dev, uint32_t* out_vals_dev, void* hip_stream);
kh_status kh_shard_plan_offsets(const kh_shard_plan* plan, uint64_t* offsets_host /* [nranks][pieces+1] */);
void kh_shard_plan_destroy(kh_shard_plan* plan);

/* ---- k-mer generation front end (BenchmarkKmerCounter.cpp:1655-1706 reads sequences through kmerind's
 *      KmerParser, which is not part of the reference tree: PARITY UNPINNED, the definition below is this library's):
 *      every window of k valid bases (ACGT, either case) of `seq` yields one 2-bit packed k-mer (first base most
 *      significant, A=0 C=1 G=2 T=3), in sequence order; any other byte ends the run (pass read/sequence lines separated
 *      by '\n').  canonical != 0: min(k-mer, reverse complement).  out_kmers needs room for n entries. */
kh_status kh_kmers_from_sequence(const void* seq /*[h|d] u8[n]*/, uint64_t n, uint32_t k /*1..32*/, int canonical, kh_mem where,
                                 uint64_t* out_kmers /*[h|d]*/, uint64_t* n_out, int device, void* hip_stream);
/* the same over raw FASTQ text (BenchmarkKmerCounter.cpp:1476-1560 reads FASTQ through kmerind's FASTQParser, absent: PARITY
 *      UNPINNED): records of 4 lines (@id, sequence, +, quality) starting at byte 0 of `text`; only the sequence lines
 *      (line number 1 mod 4, counted by '\n') yield k-mers, a k-mer never spans two reads.  Pass whole records. */
kh_status kh_kmers_from_fastq(const void* text /*[h|d] u8[n]*/, uint64_t n, uint32_t k /*1..32*/, int canonical, kh_mem where,
                              uint64_t* out_kmers /*[h|d]*/, uint64_t* n_out, int device, void* hip_stream);

/* ---- HyperLogLog cardinality estimator: fsc::hyperloglog64<T, Hash, precision> (hyperloglog64.hpp:142-475),
 *      64-bit hash values: register = top `precision` bits after dropping `ignore_msb` bits, rank = leading zeros + 1
 *      (:175-188); estimate() = harmonic mean with the linear-counting branch below 5m/2 (:201-236).  Registers are
 *      bit-exact with the reference; the estimate is computed on the host in the reference's operation order. */
typedef struct kh_hll kh_hll;
kh_status kh_hll_create(kh_hll** out, uint32_t precision /*12*/, uint32_t ignore_msb /*0*/, kh_hash hash, uint64_t seed, int device);
kh_status kh_hll_destroy(kh_hll* h);
kh_status kh_hll_set_stream(kh_hll* h, void* hip_stream);
kh_status kh_hll_update(kh_hll* h, const void* keys /*[h|d] u64[n]*/, uint64_t n, kh_mem where);               /* update(vals,count) :357 */
kh_status kh_hll_update_via_hashval(kh_hll* h, const void* hashes /*[h|d] u64[n]*/, uint64_t n, kh_mem where); /* :449-455 */
kh_status kh_hll_merge(kh_hll* h, const kh_hll* other);   /* :463 */
kh_status kh_hll_clear(kh_hll* h);                        /* :467 */
kh_status kh_hll_registers(kh_hll* h, uint8_t* out_host /* 2^precision bytes */);
kh_status kh_hll_estimate(kh_hll* h, double* out);       /* :459 */
/* internal_estimate (:201-236) on 2^precision host registers: what estimate_global (:482-484) applies to the registers merged over all
 * ranks (merge_distributed :477-479 = an all-reduce(max) of the registers, done by the caller's communication layer).  Host only. */
kh_status kh_hll_estimate_registers(const uint8_t* registers_host, uint32_t precision, double* out);

/* ---- measurement hooks: per-kernel HIP-event timing on the table's stream (bench.py roofline) */
kh_status kh_profile_enable(kh_table* t, int on);
kh_status kh_profile_reset(kh_table* t);
/* total ms and launch count of kernels whose name starts with `prefix` since the last reset */
kh_status kh_profile_query(kh_table* t, const char* prefix, double* total_ms, uint64_t* launches);
/* writes up to `cap` bytes of "name launches total_ms\n" lines */
kh_status kh_profile_dump(kh_table* t, char* buf, uint64_t cap);

/* freed table buffers and workspaces are cached per device for reuse; this returns them to the driver */
kh_status kh_release_cached_memory(int device);

const char* kh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KMERHASH_AMD_H_ */
