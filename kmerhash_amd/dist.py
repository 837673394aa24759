"""Key-space sharding of the table across the GPUs of one node (one process per GPU, torch.distributed).

Replaces the reference's MPI layer for this path (dsc::batched_robinhood_map_base::insert_p / count_p /
find_p / erase_p, distributed_batched_robinhood_map.hpp:910-1194,1258,1619,2169):
    rank = DistHash(key, seed 9876543) & (p-1)   (or % p)            :513-534,652
    permute the batch into p contiguous segments                      :632-741 (assign_count_permute)
    all2all(counts) ; all2allv(payload)                               :1024,1126 -> RCCL all_to_all_single over xGMI
    local batch op on the received keys                               :1158
    queries: results travel back with the swapped counts              :1495
Every GPU owns an independent local table whose storage hash uses a different seed (43), so local bucket
bits are uncorrelated with the shard bits.  Receive order is fixed (source rank 0..p-1, then position), which
makes first-value-wins across ranks deterministic and replayable by the CPU model in the tests.

`backend` objects supply the two device-specific pieces so that the exchange logic can be exercised on CPU
(gloo, world_size 2) with the oracle in tests; the product backend is GpuBackend.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

import os

DIST_SEED = 9876543   # distributed_batched_robinhood_map.hpp:513-534
# rehearsal switch: run the collectives even when the group has a single rank (exercises the RCCL path on one GPU)
FORCE_COLLECTIVES = os.environ.get("KH_DIST_FORCE_COLLECTIVES", "0") == "1"


class GpuBackend:
    """local table = libkmerhash_amd table on this rank's GPU; sharding = kh_shard_permute (stable)"""

    def __init__(self, device, kind="rh", capacity=128, min_lf=0.35, max_lf=0.8, hash="murmur3avx64", seed=43,
                 dist_hash="murmur3avx64", dist_seed=DIST_SEED):
        import ctypes as C
        from . import _capi as K
        from . import table as T
        self.C, self.K = C, K
        self.device = device
        cls = T.hashmap_robinhood_doubling if kind == "rh" else T.hashmap_linearprobe_doubling
        self.table = cls(capacity, min_lf, max_lf, hash=hash, seed=seed, device=device)
        self.dist_hash = T._hash_id(dist_hash)
        self.dist_seed = dist_seed
        self.torch_device = torch.device("cuda", device)

    def shard(self, keys, vals, p):
        """-> (keys grouped by destination rank, vals grouped, counts[p]) ; stable inside a rank"""
        n = keys.numel()
        ok = torch.empty_like(keys)
        ov = torch.empty_like(vals) if vals is not None else None
        counts = (self.C.c_uint64 * p)()
        st = self.K.lib().kh_shard_permute(self.dist_hash, self.dist_seed, p, keys.data_ptr(),
                                           vals.data_ptr() if vals is not None else None, n, ok.data_ptr(),
                                           ov.data_ptr() if ov is not None else None, counts, self.device,
                                           torch.cuda.current_stream(self.device).cuda_stream)
        if st != self.K.KH_OK:
            raise self.K.KhError(st, "kh_shard_permute")
        return ok, ov, [int(c) for c in counts]

    def shard_counts(self, keys, p):
        """counts[p] of shard() without permuting anything"""
        counts = (self.C.c_uint64 * p)()
        st = self.K.lib().kh_shard_permute(self.dist_hash, self.dist_seed, p, keys.data_ptr(), None, keys.numel(), None, None,
                                           counts, self.device, torch.cuda.current_stream(self.device).cuda_stream)
        if st != self.K.KH_OK:
            raise self.K.KhError(st, "kh_shard_permute (count only)")
        return [int(c) for c in counts]

    def empty(self, n, dtype):
        return torch.empty(n, dtype=dtype, device=self.torch_device)


class ShardedTable:
    """dsc::batched_robinhood_map-style distributed map over torch.distributed (RCCL when backend='nccl')."""

    def __init__(self, backend, group=None):
        self.b = backend
        self.group = group
        self.p = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self._comm = None      # side stream of the pipelined insert, created on first use

    @property
    def local(self):
        return self.b.table

    def _host_staged(self):
        return self.p > 1 and dist.get_backend(self.group) == "gloo" and self.b.torch_device.type == "cuda"

    # ---- exchange helpers ---------------------------------------------------------------------------
    def _exchange_counts(self, send_counts):
        if self.p == 1 and not FORCE_COLLECTIVES:
            return list(send_counts)
        dev = torch.device("cpu") if self._host_staged() else self.b.torch_device
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)       # mxx::all2all(send_counts) :1024
        return [int(x) for x in rc.cpu()]

    def _a2av(self, send, send_counts, recv_counts):
        if self.p == 1 and not FORCE_COLLECTIVES:
            return send
        if self._host_staged():
            # rehearsal on a backend without device collectives (gloo): stage through host memory
            hout = torch.empty(sum(recv_counts), dtype=send.dtype)
            dist.all_to_all_single(hout, send.cpu(), output_split_sizes=recv_counts, input_split_sizes=send_counts, group=self.group)
            return hout.to(send.device)
        out = self.b.empty(sum(recv_counts), send.dtype)
        dist.all_to_all_single(out, send, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                               group=self.group)               # khmxx::distribute_permuted (mxx::all2allv) :1126
        return out

    def _route(self, keys, vals=None):
        ok, ov, sc = self.b.shard(keys, vals, self.p)
        rc = self._exchange_counts(sc)
        rk = self._a2av(ok, sc, rc)
        rv = self._a2av(ov, sc, rc) if ov is not None else None
        return rk, rv, sc, rc

    # ---- batch operations (collective: every rank calls them) -----------------------------------------
    def insert_counts(self, keys, chunks=1):
        """counting_batched_robinhood_map::insert(vector<Key>) (distributed_batched_robinhood_map.hpp:2542-2950): every key
        occurrence adds 1 to its k-mer's count on the owner rank (Reducer = std::plus).  Same exchange as insert()."""
        return self.insert(keys, None, chunks=chunks, reduce_plus=True)

    def insert(self, keys, vals, chunks=1, reduce_plus=False):
        """insert_p :910-1194.  chunks == 1: shard, exchange, one bulk insert.
        chunks > 1: the RCCL analogue of khmxx::ialltoallv_and_modify (incremental_mxx.hpp:3437-3645).  The batch is cut into
        `chunks` pieces.  A count-only pass over every piece and ONE exchange of all the counts tell each rank exactly how
        many pairs it will receive, piece by piece.  Then piece i is permuted on the compute stream, its payload travels on
        the comm stream, and piece i-1 -- already landed -- is radix-partitioned into the local table's streamed insert
        (kh_insert_feed) on the compute stream meanwhile; kh_insert_end de-duplicates and builds once.  xGMI transfers and
        HBM-bound kernels use different resources, so permute + partition hide under the exchange (or the other way round).
        Same result as chunks == 1 with the pieces concatenated piece-major (piece, source rank, position)."""
        n = keys.numel()
        if chunks <= 1 or (self.p == 1 and not FORCE_COLLECTIVES) or not keys.is_cuda:
            rk, rv, _, _ = self._route(keys, vals)
            return self.local.insert_reduce_plus(rk, rv) if reduce_plus else self.local.insert(rk, rv)
        bounds = [n * i // chunks for i in range(chunks + 1)]
        # per-piece destination counts (count-only pass), one exchange for all of them: row = destination rank, column = piece
        sc_piece = [self.b.shard_counts(keys[bounds[i]:bounds[i + 1]], self.p) for i in range(chunks)]
        host = self._host_staged()
        cdev = torch.device("cpu") if host else self.b.torch_device
        sc = torch.tensor([[sc_piece[i][r] for i in range(chunks)] for r in range(self.p)], dtype=torch.int64, device=cdev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc.contiguous(), group=self.group)
        rc = rc.cpu().tolist()                                     # rc[src][piece]
        total = sum(sum(row) for row in rc)
        self.local.insert_begin(total, reduce_plus=reduce_plus)
        cur = torch.cuda.current_stream(self.b.torch_device)
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=self.b.torch_device)
        comm = self._comm
        keep, landed = [], None
        for i in range(chunks):
            a, b = bounds[i], bounds[i + 1]
            ok, ov, scounts = self.b.shard(keys[a:b], vals[a:b] if vals is not None else None, self.p)     # compute stream (stable permutation)
            rcounts = [rc[src][i] for src in range(self.p)]
            comm.wait_stream(cur)
            with torch.cuda.stream(comm):
                rk = self._a2av(ok, scounts, rcounts)
                rv = self._a2av(ov, scounts, rcounts) if ov is not None else None
                for x in (ok, ov):
                    if x is not None:
                        x.record_stream(comm)
                for x in (rk, rv):
                    if x is not None:
                        x.record_stream(cur)
                ev = torch.cuda.Event()
                ev.record(comm)
            if landed is not None:                                 # piece i-1: partition it while piece i travels
                cur.wait_event(landed[0])
                self.local.insert_feed(landed[1], landed[2])
            landed = (ev, rk, rv)
            keep.append((ok, ov, rk, rv))
        cur.wait_event(landed[0])
        self.local.insert_feed(landed[1], landed[2])
        inserted = self.local.insert_end()
        del keep
        return inserted

    def count(self, keys):
        """count_p :1258: results come back in the PERMUTED input order (grouped by owner rank), like the
        reference, together with the permuted keys."""
        ok, _, sc = self.b.shard(keys, None, self.p)
        rc = self._exchange_counts(sc)
        rk = self._a2av(ok, sc, rc)
        res = self.local.count(rk)
        back = self._a2av(res, rc, sc)     # swapped counts :1495
        return ok, back

    def find(self, keys):
        """find_p :1619: (permuted keys, values, found flags) aligned with the permuted keys"""
        ok, _, sc = self.b.shard(keys, None, self.p)
        rc = self._exchange_counts(sc)
        rk = self._a2av(ok, sc, rc)
        vals, found = self.local.find_values(rk)
        return ok, self._a2av(vals, rc, sc), self._a2av(found, rc, sc)

    def erase(self, keys):
        """erase_p :2169: returns the number erased on this rank's local table"""
        rk, _, _, _ = self._route(keys, None)
        return self.local.erase(rk)

    def size(self):
        """global size = sum of local sizes"""
        n = self.local.size()
        if self.p == 1 and not FORCE_COLLECTIVES:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=torch.device("cpu") if self._host_staged() else self.b.torch_device)
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def local_size(self):
        return self.local.size()
