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
 * khd_last_error, khd_phase_ms, khd_set_stream, khd_set_option, khd_synchronize and khd_debug_fail_next is COLLECTIVE: all ranks
 * call it, in the same order.
 *
 * Failures.  The reference runs under MPI, where an exception on one rank ends the job (MPI_Abort).  Here a rank that fails LOCALLY
 * inside a collective call (allocation, a kh_* call on its table) never leaves its peers waiting: it keeps taking part in the
 * collectives the call still has to run and skips its local work; status votes -- a word that travels with the count exchange, an
 * 8-byte all-reduce before the first payload exchange, for insert / erase one after the last payload exchange (no rank builds or erases
 * when a rank could not send its real pairs: nothing is applied anywhere) and one at the end -- make EVERY rank return a non-OK
 * status (the failing rank its own, the others the worst status seen; khd_last_error says which).  Only a failure in the build / local
 * erase itself (the last vote) leaves the ranks that did not fail with their share of the batch applied.  find / count return without waiting for the device: the status of their local part
 * travels with the last result exchange and is reported by khd_synchronize or by the next collective call.  A failure of the
 * transport itself (an RCCL error, a peer that does not arrive within KHD_OPT_TIMEOUT_MS) aborts the communicator: the call and
 * every later one return KH_ERR_HIP.
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
/* options (set them identically on every rank where noted) */
enum {
  KHD_OPT_FORCE_COLLECTIVES = 1, /* != 0: a map of ONE rank runs the exchange code of the multi-rank case (count exchange, grouped
                                    all-to-all-v per piece, votes) instead of the short cut to its local table; also set by the environment
                                    variable KH_DIST_FORCE_COLLECTIVES=1 at creation.  (Runs every RCCL call of the library on a one-GPU box.) */
  KHD_OPT_QUERY_PIECES = 2,      /* pieces a find / count batch of THIS rank is exchanged in (1..8; 0 = by size: 4 from 2^23 keys, 2 from
                                    2^22): the keys of piece i+1 travel while piece i is looked up and its results return
                                    (khmxx::ialltoallv_and_query_one_to_one, incremental_mxx.hpp:4403-4669).  May differ between ranks. */
  KHD_OPT_TIMEOUT_MS = 3         /* how long a rank waits for its peers inside a collective before it aborts the communicator (default
                                    300000; environment KHD_TIMEOUT_MS) */
};
kh_status khd_set_option(khd_map* m, int option, long long value);
/* waits for everything the map has queued (table stream and exchange stream) and returns the status of the last find / count across
 * ALL ranks (see "Failures" above).  Not collective, but call it on every rank or on none: a rank that has not looked at the
 * status of the last find / count reports it from its NEXT collective call instead of running that call, and so must all others. */
kh_status khd_synchronize(khd_map* m);
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
/* count_p (:1258): out_keys_dev[n] = the keys grouped by owner rank, out01_dev[n] = 0/1 aligned with them.
 * khd_count and khd_find only QUEUE their work: the outputs are complete in the order of the map's stream (khd_set_stream), or after
 * khd_synchronize, which also reports whether every rank's local part succeeded. */
kh_status khd_count(khd_map* m, const uint64_t* keys_dev, uint64_t n, uint64_t* out_keys_dev, uint8_t* out01_dev);
/* find_p (:1619): values (0 on a miss) and found flags aligned with the permuted keys */
kh_status khd_find(khd_map* m, const uint64_t* keys_dev, uint64_t n, uint64_t* out_keys_dev, uint32_t* out_vals_dev, uint8_t* out_found_dev);
/* erase_p (:2169): number erased from this rank's local table */
kh_status khd_erase(khd_map* m, const uint64_t* keys_dev, uint64_t n, uint64_t* n_erased_local);
/* sum of the local sizes */
kh_status khd_size(khd_map* m, uint64_t* global_size);

/* test hook: the next collective call of this rank fails locally at `stage` (1: before the count exchange, 2: on the receive side
 * before any payload, 3: in the local work between the payload exchanges, 4: in the build at the end of an insert / the local erase) as if an
 * allocation had failed there.  Used by tests/cpp/test_dist.cpp to check that no rank hangs and every rank reports. */
kh_status khd_debug_fail_next(khd_map* m, int stage);

/* per-phase device time of this rank since the last call (HIP events; ms): permute, exchange, feed, build, query.
 * Writes up to `cap` bytes of "name ms\n" lines. */
kh_status khd_phase_ms(khd_map* m, char* buf, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* KMERHASH_AMD_DIST_H_ */
