/* kmerhash_amd_dist.h -- C-ABI of libkmerhash_amd_dist.so: the table of kmerhash_amd.h sharded over the GPUs of one node.
 *
 * Replaces the reference's MPI layer for this path, host side in C++ like the reference's:
 *     dsc::batched_robinhood_map_base::insert_p / count_p / find_p / erase_p
 *         (reference include/kmerhash/distributed_batched_robinhood_map.hpp:910-1194, 1258, 1619, 2169)
 *     rank = DistHash(key, seed 9876543) & (p-1)  (or % p)                              (:513-534, :652)
 *     assign_count_permute -> all2all(counts) -> all2allv(payload) -> local batch op     (:632-741, :1024, :1126, :1158)
 *     khmxx::ialltoallv_and_modify (overlapped exchange + insert)                        (io/incremental_mxx.hpp:3437-3645)
 * One process (or thread) per GPU; the exchange is RCCL (ncclAllToAllv of keys and values grouped into ONE launch per
 * piece, every peer pair one xGMI link) on a side stream, overlapped with the radix partition of the piece that landed
 * before (kh_insert_feed) on the table's stream.  Receive order is (source rank 0..p-1, position): first-value-wins across
 * ranks is deterministic.  Results of count / find come back in the PERMUTED input order, next to the permuted keys, as in
 * the reference (:1495).
 *
 * Bootstrap: the caller carries the 128-byte communicator id from rank 0 to every rank with whatever it has (the reference's
 * callers have MPI: MPI_Bcast).  khd_create_local makes all ranks inside ONE process on ONE device
 * (a thread per rank calls the collective entry points): the same sharding code over an in-process transport, used by the
 * tests to run p > 1 on a one-GPU box, where RCCL refuses two ranks on one device.
 *
 * All batch arguments are DEVICE pointers on the map's device.  Every entry point below except khd_unique_id, khd_local,
 * khd_last_error and khd_phase_ms is COLLECTIVE: all ranks call it, in the same order.
 */
#ifndef KMERHASH_AMD_DIST_H_
#define KMERHASH_AMD_DIST_H_

#include "kmerhash_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct khd_map khd_map;

#define KHD_UNIQUE_ID_BYTES 128
#define KHD_DIST_SEED 9876543ull /* distributed_batched_robinhood_map.hpp:513-534 */

/* rank 0: a fresh communicator id (ncclGetUniqueId) to be handed to every rank */
kh_status khd_unique_id(void* id128);

/* one rank of an nranks-way sharded map on `device` (RCCL communicator over the id; nranks == 1 needs no peers).
 * kind/hash/seed/capacity/load factors: the local table (kh_create); dist_hash/dist_seed: the sharding hash. */
kh_status khd_create(khd_map** out, const void* id128, int nranks, int rank, int device, kh_kind kind, kh_hash hash, uint64_t seed,
                     uint64_t capacity, float min_load_factor, float max_load_factor, kh_hash dist_hash, uint64_t dist_seed);
/* all ranks in this process on one device: out[0..nranks) (each used by its own thread) */
kh_status khd_create_local(khd_map** out, int nranks, int device, kh_kind kind, kh_hash hash, uint64_t seed, uint64_t capacity,
                           float min_load_factor, float max_load_factor, kh_hash dist_hash, uint64_t dist_seed);
kh_status khd_destroy(khd_map* m);
kh_status khd_set_stream(khd_map* m, void* hip_stream);
const char* khd_last_error(const khd_map* m);
kh_table* khd_local(khd_map* m); /* this rank's table: kh_size, kh_capacity, kh_to_vector, kh_export_info ... */
int khd_rank(const khd_map* m);
int khd_nranks(const khd_map* m);

/* insert_p (:910-1194).  vals == NULL with reduce_plus != 0: counting insert (std::plus, 1 per occurrence;
 * counting_batched_robinhood_map::insert :2542-2950).  pieces > 1: the batch is cut into `pieces` parts whose exchange overlaps the
 * radix partition of the part before (at most 16); same result as pieces == 1 with the parts concatenated piece-major.  Up to 8
 * ranks the parts are cut at multiples of 4096 pairs (part i = pairs [4096 * (T * i / pieces), 4096 * (T * (i + 1) / pieces)),
 * T = ceil(n / 4096): kh_shard_plan), otherwise at n * i / pieces. */
kh_status khd_insert(khd_map* m, const uint64_t* keys_dev, const uint32_t* vals_dev, uint64_t n, int pieces, int reduce_plus,
                     uint64_t* n_inserted_local);
/* count_p (:1258): out_keys_dev[n] = the keys grouped by owner rank, out01_dev[n] = 0/1 aligned with them */
kh_status khd_count(khd_map* m, const uint64_t* keys_dev, uint64_t n, uint64_t* out_keys_dev, uint8_t* out01_dev);
/* find_p (:1619): values (untouched on a miss) and found flags aligned with the permuted keys */
kh_status khd_find(khd_map* m, const uint64_t* keys_dev, uint64_t n, uint64_t* out_keys_dev, uint32_t* out_vals_dev, uint8_t* out_found_dev);
/* erase_p (:2169): number erased from this rank's local table */
kh_status khd_erase(khd_map* m, const uint64_t* keys_dev, uint64_t n, uint64_t* n_erased_local);
/* sum of the local sizes */
kh_status khd_size(khd_map* m, uint64_t* global_size);

/* per-phase device time of this rank since the last call (HIP events; ms): permute, exchange, feed, build, query.
 * Writes up to `cap` bytes of "name ms\n" lines. */
kh_status khd_phase_ms(khd_map* m, char* buf, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* KMERHASH_AMD_DIST_H_ */
