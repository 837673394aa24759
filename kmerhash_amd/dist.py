"""Key-space sharding of the table across the GPUs of one node (one process per GPU, torch.distributed).

Replaces the reference's MPI layer for this path (dsc::batched_robinhood_map_base::insert_p / count_p /
find_p / erase_p, distributed_batched_robinhood_map.hpp:910-1194,1258,1619,2169):
    rank = DistHash(key, seed 9876543) & (p-1)   (or % p)            :513-534,652
    permute the batch into p contiguous segments                      :632-741 (assign_count_permute)
    all2all(counts) ; all2allv(payload)                               :1024,1126 -> RCCL over xGMI
    local batch op on the received keys                               :1158
    queries: results travel back with the swapped counts              :1495
Every GPU owns an independent local table whose storage hash uses a different seed (43), so local bucket
bits are uncorrelated with the shard bits.  Receive order is fixed (source rank 0..p-1, then position), which
makes first-value-wins across ranks deterministic and replayable by the CPU model in the tests.

Collectives per call (p > 1):
    insert(chunks=k) : 1 exchange of ALL per-piece counts, then ONE payload exchange per piece
    count / erase    : 1 count exchange + 1 payload exchange (+ 1 result exchange for count)
    find             : 1 count exchange + 1 key exchange + 1 result exchange (values and found flags together)
A payload exchange is ONE grouped point-to-point launch (ncclGroupStart .. ncclSend/ncclRecv per peer and array ..
ncclGroupEnd through torch's batch_isend_irecv -- what all_to_all_single with split sizes is made of), so keys and
values of a piece travel SoA (12 B per pair, no padding) in a single RCCL kernel; every peer pair is one xGMI link.

`backend` objects supply the two device-specific pieces so that the exchange logic can be exercised on CPU
(gloo, world_size 2/3) with the oracle in tests; the product backend is GpuBackend.
"""
import os

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

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

    def shard_plan(self, keys, p, pieces):
        """ONE count sweep + scan + host synchronisation for a batch that travels in `pieces` pieces (kh_shard_plan_create):
        -> (plan, bounds[pieces+1] (tile-aligned piece boundaries), counts[piece][p]); None when the library cannot plan it"""
        if p > 8:
            return None
        plan = self.C.c_void_p()
        counts = (self.C.c_uint64 * (p * pieces))()
        bounds = (self.C.c_uint64 * (pieces + 1))()
        st = self.K.lib().kh_shard_plan_create(self.C.byref(plan), self.dist_hash, self.dist_seed, 0, 0, p, keys.data_ptr(), keys.numel(), pieces,
                                               counts, bounds, self.device, torch.cuda.current_stream(self.device).cuda_stream)
        if st != self.K.KH_OK:
            raise self.K.KhError(st, "kh_shard_plan_create")
        return _ShardPlan(self, plan, keys), [int(b) for b in bounds], [[int(counts[i * p + r]) for r in range(p)] for i in range(pieces)]


def plan_piece_bounds(n, pieces):
    """where kh_shard_plan cuts a batch of n pairs into `pieces` pieces: at multiples of 4096 pairs"""
    nt = (n + 4095) // 4096
    return [min(n, (nt * i // pieces) * 4096) for i in range(pieces + 1)]


class _ShardPlan:
    def __init__(self, be, handle, keys):
        self.be, self.h, self.keys = be, handle, keys      # (keeps the key tensor alive: the plan refers to it)

    def permute(self, piece, vals, n_piece):
        """piece `piece` of the planned batch grouped by destination rank (no second count, no synchronisation)"""
        be = self.be
        ok = torch.empty(n_piece, dtype=self.keys.dtype, device=self.keys.device)
        ov = torch.empty(n_piece, dtype=vals.dtype, device=vals.device) if vals is not None else None
        st = be.K.lib().kh_shard_plan_permute(self.h, piece, self.keys.data_ptr(), vals.data_ptr() if vals is not None else None, ok.data_ptr(),
                                              ov.data_ptr() if ov is not None else None, torch.cuda.current_stream(be.device).cuda_stream)
        if st != be.K.KH_OK:
            raise be.K.KhError(st, "kh_shard_plan_permute")
        return ok, ov

    def close(self):
        if self.h:
            self.be.K.lib().kh_shard_plan_destroy(self.h)
            self.h = None


class _Phases:
    """per-phase device timings of the sharded operations (HIP events on the stream the phase runs on); off by default"""

    def __init__(self, cuda):
        self.cuda = cuda
        self.pending = []          # (name, start event, end event)
        self.host_ms = {}          # name -> [ms] for CPU backends

    def span(self, name, stream=None):
        return _Span(self, name, stream)

    def collect(self):
        out = {k: list(v) for k, v in self.host_ms.items()}
        if self.cuda and self.pending:
            torch.cuda.synchronize()
            for name, a, b in self.pending:
                out.setdefault(name, []).append(a.elapsed_time(b))
        self.pending, self.host_ms = [], {}
        return {k: float(sum(v)) for k, v in out.items()}


class _Span:
    def __init__(self, ph, name, stream):
        self.ph, self.name, self.stream = ph, name, stream

    def __enter__(self):
        if self.ph is None:
            return self
        if self.ph.cuda:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(self.stream if self.stream is not None else torch.cuda.current_stream())
        else:
            import time
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.ph is None:
            return False
        if self.ph.cuda:
            self.b.record(self.stream if self.stream is not None else torch.cuda.current_stream())
            self.ph.pending.append((self.name, self.a, self.b))
        else:
            import time
            self.ph.host_ms.setdefault(self.name, []).append((time.perf_counter() - self.t0) * 1e3)
        return False


class ShardedTable:
    """dsc::batched_robinhood_map-style distributed map over torch.distributed (RCCL when backend='nccl')."""

    def __init__(self, backend, group=None, timing=False):
        self.b = backend
        self.group = group
        self.p = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist is not None and dist.is_initialized() else 0
        self._comm = None      # side stream of the pipelined insert, created on first use
        self.collectives = {"counts": 0, "payload": 0}     # launches issued by this rank (tests assert the per-call numbers)
        self._ph = _Phases(self.b.torch_device.type == "cuda") if timing else None

    @property
    def local(self):
        return self.b.table

    def timings(self):
        """{phase: total ms since the last call} when constructed with timing=True"""
        return self._ph.collect() if self._ph is not None else {}

    def _span(self, name, stream=None):
        return _Span(self._ph, name, stream)

    def _single(self):
        return self.p == 1 and not FORCE_COLLECTIVES

    def _host_staged(self):
        return self.p > 1 and dist.get_backend(self.group) == "gloo" and self.b.torch_device.type == "cuda"

    # ---- exchange helpers ---------------------------------------------------------------------------
    def _exchange_counts(self, send_counts):
        """send_counts: [p] or [p][k] (row = destination rank) -> what every source sends here, same shape (row = source rank).
        mxx::all2all(send_counts) :1024 -- ONE collective whatever k is."""
        if self._single():
            return [list(r) if isinstance(r, (list, tuple)) else r for r in send_counts]
        dev = torch.device("cpu") if (self._host_staged() or self.b.torch_device.type == "cpu") else self.b.torch_device
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev).contiguous()
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        self.collectives["counts"] += 1
        return rc.cpu().tolist()

    def _exchange(self, sends, send_counts, recv_counts):
        """all-to-all-v of several arrays that share their split sizes (khmxx::distribute_permuted / mxx::all2allv :1126), as
        ONE grouped point-to-point launch.  sends: 1-D tensors grouped by destination rank; returns tensors grouped by
        source rank (rank 0 first, sender's order kept inside a rank).  None entries pass through."""
        live = [s for s in sends if s is not None]
        if self._single() or not live:
            return list(sends)
        host = self._host_staged()
        src = [s.cpu() if host else s for s in live]
        tot = int(sum(recv_counts))
        outs = [torch.empty(tot, dtype=s.dtype) if host else self.b.empty(tot, s.dtype) for s in live]
        soff = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
        roff = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        ops = []
        for d in range(self.p):
            peer = (self.rank + d) % self.p        # every rank starts with itself and walks the ring: pairs line up
            for j, (s, o) in enumerate(zip(src, outs)):
                sseg = s[int(soff[peer]):int(soff[peer + 1])]
                rseg = o[int(roff[peer]):int(roff[peer + 1])]
                if peer == self.rank:
                    rseg.copy_(sseg)
                    continue
                g = dist.get_global_rank(self.group, peer) if self.group is not None else peer
                if sseg.numel():
                    ops.append(dist.P2POp(dist.isend, sseg, g, self.group, tag=j))
                if rseg.numel():
                    ops.append(dist.P2POp(dist.irecv, rseg, g, self.group, tag=j))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self.collectives["payload"] += 1
        if host:
            outs = [o.to(s.device) for o, s in zip(outs, live)]
        it = iter(outs)
        return [next(it) if s is not None else None for s in sends]

    def _route(self, keys, vals=None):
        if self._single():             # one rank owns every key: nothing to permute or exchange
            n = int(keys.numel())
            return keys, keys, vals, [n], [n]
        with self._span("permute"):
            ok, ov, sc = self.b.shard(keys, vals, self.p)
        rc = self._exchange_counts(sc)
        with self._span("exchange"):
            rk, rv = self._exchange([ok, ov], sc, rc)
        return ok, rk, rv, sc, rc

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
        the comm stream (one grouped launch), and piece i-1 -- already landed -- is radix-partitioned into the local table's
        streamed insert (kh_insert_feed) on the compute stream meanwhile; kh_insert_end de-duplicates and builds once.  xGMI
        transfers and HBM-bound kernels use different resources, so permute + partition hide under the exchange (or the
        other way round).  The received pieces are kept until the build has succeeded (a repeatable streamed insert, see below).
        Same result as chunks == 1 with the pieces concatenated piece-major (piece, source rank, position)."""
        n = keys.numel()
        if chunks <= 1 or self._single():
            _, rk, rv, _, _ = self._route(keys, vals)
            with self._span("local_insert"):
                return self.local.insert_reduce_plus(rk, rv) if reduce_plus else self.local.insert(rk, rv)
        cuda = keys.is_cuda            # host tensors (CPU test backends): the same piece loop without streams
        bounds = [n * i // chunks for i in range(chunks + 1)]
        # per-piece destination counts (count-only pass), one exchange for all of them: row = destination rank, column = piece.
        # GPU backend: one sweep + one synchronisation for all pieces (kh_shard_plan), the pieces are then permuted without recounting
        plan = None
        with self._span("count_pass"):
            planned = self.b.shard_plan(keys, self.p, chunks) if hasattr(self.b, "shard_plan") else None
            if planned is not None:
                plan, bounds, sc_piece = planned
            else:
                sc_piece = [self.b.shard_counts(keys[bounds[i]:bounds[i + 1]], self.p) for i in range(chunks)]
        rc = self._exchange_counts([[sc_piece[i][r] for i in range(chunks)] for r in range(self.p)])   # rc[src][piece]
        total = sum(sum(row) for row in rc)
        # the received pieces are kept until the build has succeeded ("repeatable"): the local table may then partition them without
        # a histogram pass into slots the pieces share; if that speculation fails (skewed / duplicated keys: KhRetry) the kept
        # pieces are fed again the exact way
        kept = []
        self.local.insert_begin(total, reduce_plus=reduce_plus, repeatable=True)
        cur = comm = None
        if cuda:
            cur = torch.cuda.current_stream(self.b.torch_device)
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.b.torch_device)
            comm = self._comm
        landed = None
        for i in range(chunks):
            a, b = bounds[i], bounds[i + 1]
            with self._span("permute"):
                if plan is not None:
                    ok, ov = plan.permute(i, vals, b - a)
                    scounts = sc_piece[i]
                else:
                    ok, ov, scounts = self.b.shard(keys[a:b], vals[a:b] if vals is not None else None, self.p)     # compute stream (stable permutation)
            assert list(scounts) == list(sc_piece[i]), "count-only pass and permutation disagree"
            rcounts = [rc[src][i] for src in range(self.p)]
            ev = None
            if cuda:
                comm.wait_stream(cur)
                with torch.cuda.stream(comm):
                    with self._span("exchange", comm):
                        rk, rv = self._exchange([ok, ov], scounts, rcounts)
                    ev = torch.cuda.Event()
                    ev.record(comm)
                for x in (ok, ov):             # allocated on the compute stream, read on the comm stream
                    if x is not None:
                        x.record_stream(comm)
                for x in (rk, rv):             # allocated on the comm stream, read on the compute stream
                    if x is not None:
                        x.record_stream(cur)
            else:
                with self._span("exchange"):
                    rk, rv = self._exchange([ok, ov], scounts, rcounts)
            if landed is not None:                                 # piece i-1: partition it while piece i travels
                if cuda:
                    cur.wait_event(landed[0])
                with self._span("feed"):
                    self.local.insert_feed(landed[1], landed[2])
                kept.append((landed[1], landed[2]))
            landed = (ev, rk, rv)
            del ok, ov, rk, rv
        if cuda:
            cur.wait_event(landed[0])
        with self._span("feed"):
            self.local.insert_feed(landed[1], landed[2])
        kept.append((landed[1], landed[2]))
        landed = None
        with self._span("build"):
            try:
                return self.local.insert_end()
            except Exception as ex:
                if type(ex).__name__ != "KhRetry":
                    raise
            finally:
                if plan is not None:      # (insert_end has synchronised: the last piece's permutation no longer reads the plan)
                    plan.close()
        with self._span("refeed"):
            self.local.insert_begin(total, reduce_plus=reduce_plus)
            for rk, rv in kept:
                self.local.insert_feed(rk, rv)
            return self.local.insert_end()

    def count(self, keys):
        """count_p :1258: results come back in the PERMUTED input order (grouped by owner rank), like the
        reference, together with the permuted keys."""
        ok, rk, _, sc, rc = self._route(keys, None)
        with self._span("local_query"):
            res = self.local.count(rk)
        with self._span("exchange"):
            back, = self._exchange([res], rc, sc)     # swapped counts :1495
        return ok, back

    def find(self, keys):
        """find_p :1619: (permuted keys, values, found flags) aligned with the permuted keys; values and flags return in one
        exchange"""
        ok, rk, _, sc, rc = self._route(keys, None)
        with self._span("local_query"):
            vals, found = self.local.find_values(rk)
        with self._span("exchange"):
            bv, bf = self._exchange([vals, found], rc, sc)
        return ok, bv, bf

    def erase(self, keys):
        """erase_p :2169: returns the number erased on this rank's local table"""
        _, rk, _, _, _ = self._route(keys, None)
        with self._span("local_erase"):
            return self.local.erase(rk)

    def size(self):
        """global size = sum of local sizes"""
        n = self.local.size()
        if self._single():
            return n
        t = torch.tensor([n], dtype=torch.int64, device=torch.device("cpu") if self._host_staged() else self.b.torch_device)
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def local_size(self):
        return self.local.size()
