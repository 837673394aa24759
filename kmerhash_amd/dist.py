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
    insert(chunks=k) : 1 exchange of ALL per-piece counts, ONE payload exchange per piece, 3 votes
    count / find     : 1 count exchange, 1 vote, then per piece 1 key exchange + 1 result exchange (values and flags together)
    erase            : 1 count exchange + 1 payload exchange, 3 votes
A vote is an all-reduce(max) of one status word.  Failure protocol (same as libkmerhash_amd_dist, see kmerhash_amd_dist.cpp): a rank
that fails locally keeps taking part in the collectives the call still has to run, skips its local work, and every rank raises --
the failing rank its own exception, the others ShardPeerError.  find / count do not wait for the device: the status of their local
part travels with the last result exchange and is raised by synchronize() or by the next collective call.
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


class ShardPeerError(RuntimeError):
    """another rank failed inside this collective call (its status word reached this rank through a vote)"""

    def __init__(self, status, where):
        super().__init__("a peer rank failed (status %d) %s" % (status, where))
        self.status = status


MAX_QUERY_PIECES = 8


class _ShardPlan:
    def __init__(self, be, handle, keys):
        self.be, self.h, self.keys = be, handle, keys      # (keeps the key tensor alive: the plan refers to it)

    def offsets(self, p, pieces):
        """[rank][piece] -> where rank's part of that piece starts in the layout of the whole batch grouped by rank (pieces + 1 entries)"""
        out = (self.be.C.c_uint64 * (p * (pieces + 1)))()
        st = self.be.K.lib().kh_shard_plan_offsets(self.h, out)
        if st != self.be.K.KH_OK:
            raise self.be.K.KhError(st, "kh_shard_plan_offsets")
        return [[int(out[r * (pieces + 1) + i]) for i in range(pieces + 1)] for r in range(p)]

    def permute_global(self, piece, out_keys):
        """piece `piece` of the planned batch written to its place in `out_keys` (all n keys grouped by rank)"""
        be = self.be
        st = be.K.lib().kh_shard_plan_permute_global(self.h, piece, self.keys.data_ptr(), None, out_keys.data_ptr(), None,
                                                     torch.cuda.current_stream(be.device).cuda_stream)
        if st != be.K.KH_OK:
            raise be.K.KhError(st, "kh_shard_plan_permute_global")

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
        self._comm = None      # side stream of the pipelined operations, created on first use
        self.collectives = {"counts": 0, "payload": 0, "votes": 0}     # launches issued by this rank (tests assert the per-call numbers)
        self._ph = _Phases(self.b.torch_device.type == "cuda") if timing else None
        self.query_pieces = 0          # pieces of THIS rank's find / count batches (0 = by size; GPU backend only)
        self._late = None              # status words received with the last find / count (tensor [p]), not looked at yet
        self._fail_stage = 0           # test hook: the next collective call fails locally at this stage (1..4)

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
        return (self.p > 1 or FORCE_COLLECTIVES) and dist.get_backend(self.group) == "gloo" and self.b.torch_device.type == "cuda"

    def _ctl_device(self):
        return torch.device("cpu") if (self._host_staged() or self.b.torch_device.type == "cpu") else self.b.torch_device

    # ---- failure protocol ---------------------------------------------------------------------------
    def _inject(self, stage):
        if self._fail_stage == stage:
            self._fail_stage = 0
            raise MemoryError("injected failure (stage %d)" % stage)

    @staticmethod
    def _status_of(ex):
        return int(getattr(ex, "status", 5)) or 5        # KH_ERR_HIP for anything that is not a library status

    def _vote(self, ex, where):
        """all-reduce(max) of this rank's status; raises on EVERY rank if any rank failed (the failing rank its own exception)"""
        t = torch.tensor([self._status_of(ex) if ex is not None else 0], dtype=torch.int64, device=self._ctl_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.collectives["votes"] += 1
        self._raise_if(ex, int(t.item()), where)

    @staticmethod
    def _raise_if(ex, worst, where):
        if ex is not None:
            raise ex
        if worst:
            raise ShardPeerError(worst, where)

    def _late_check(self):
        if self._late is None:
            return
        t, self._late = self._late, None
        worst = int(t.max().item())                      # (waits for the exchange that delivered the words)
        if worst:
            raise ShardPeerError(worst, "in the local part of the previous find / count: its results are invalid")

    def synchronize(self):
        """waits for the queued work of this rank and raises if any rank failed in the local part of the last find / count"""
        if self.b.torch_device.type == "cuda":
            torch.cuda.synchronize(self.b.torch_device)
        self._late_check()

    # ---- exchange helpers ---------------------------------------------------------------------------
    def _exchange_counts(self, send_counts, ex=None):
        """send_counts: [p][k] (row = destination rank) -> ([p][k] what every source sends here (row = source rank), worst status).
        The status word of the caller's local work so far travels with the counts.  mxx::all2all(send_counts) :1024 -- ONE collective."""
        rows = [list(r) + [self._status_of(ex) if ex is not None else 0] for r in send_counts]
        sc = torch.tensor(rows, dtype=torch.int64, device=self._ctl_device()).contiguous()
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        self.collectives["counts"] += 1
        rc = rc.cpu().tolist()
        return [r[:-1] for r in rc], max(r[-1] for r in rc)

    def _exchange(self, arrays):
        """all-to-all-v of several arrays (khmxx::distribute_permuted / mxx::all2allv :1126) as ONE grouped point-to-point launch.
        arrays: (send tensor, send offsets[p], send counts[p], recv tensor, recv offsets[p], recv counts[p]) in elements; the
        receive tensors are filled in place."""
        host = self._host_staged()
        work = []
        for (s, so, sn, r, ro, rn) in arrays:
            work.append((s.cpu() if host else s, so, sn, torch.empty(r.numel(), dtype=r.dtype) if host else r, ro, rn, r))
        ops = []
        self_ops = FORCE_COLLECTIVES and dist.get_backend(self.group) == "nccl"       # the self segment through RCCL too (rehearsal)
        for d in range(self.p):
            peer = (self.rank + d) % self.p        # every rank starts with itself and walks the ring: pairs line up
            for j, (s, so, sn, r, ro, rn, _) in enumerate(work):
                sseg = s[so[peer]:so[peer] + sn[peer]]
                rseg = r[ro[peer]:ro[peer] + rn[peer]]
                if peer == self.rank and not self_ops:
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
        if host:        # only what was received: the rest of a receive tensor belongs to other rounds
            for (_, _, _, r, ro, rn, dst) in work:
                for peer in range(self.p):
                    if rn[peer]:
                        dst[ro[peer]:ro[peer] + rn[peer]].copy_(r[ro[peer]:ro[peer] + rn[peer]])

    @staticmethod
    def _offs(counts):
        out, run = [], 0
        for c in counts:
            out.append(run)
            run += int(c)
        return out

    def _exchange_grouped(self, sends, send_counts, recv_counts):
        """tensors grouped by destination rank -> tensors grouped by source rank (None entries pass through)"""
        tot = int(sum(recv_counts))
        so, ro = self._offs(send_counts), self._offs(recv_counts)
        outs = [self.b.empty(tot, s.dtype) if s is not None else None for s in sends]
        self._exchange([(s, so, send_counts, o, ro, recv_counts) for s, o in zip(sends, outs) if s is not None])
        return outs

    # ---- batch operations (collective: every rank calls them) -----------------------------------------
    def insert_counts(self, keys, chunks=1):
        """counting_batched_robinhood_map::insert(vector<Key>) (distributed_batched_robinhood_map.hpp:2542-2950): every key
        occurrence adds 1 to its k-mer's count on the owner rank (Reducer = std::plus).  Same exchange as insert()."""
        return self.insert(keys, None, chunks=chunks, reduce_plus=True)

    def insert(self, keys, vals, chunks=1, reduce_plus=False):
        """insert_p :910-1194.  The RCCL analogue of khmxx::ialltoallv_and_modify (incremental_mxx.hpp:3437-3645).  The batch is cut
        into `chunks` pieces.  A count-only pass over every piece and ONE exchange of all the counts tell each rank exactly how
        many pairs it will receive, piece by piece.  Then piece i is permuted on the compute stream, its payload travels on
        the comm stream (one grouped launch), and piece i-1 -- already landed -- is radix-partitioned into the local table's
        streamed insert (kh_insert_feed) on the compute stream meanwhile; kh_insert_end de-duplicates and builds once.  xGMI
        transfers and HBM-bound kernels use different resources, so permute + partition hide under the exchange (or the
        other way round).  The received pieces are kept until the build has succeeded (a repeatable streamed insert, see below).
        Same result as ONE insert of the pieces concatenated piece-major (piece, source rank, position)."""
        self._late_check()
        n = keys.numel()
        if self._single():
            with self._span("local_insert"):
                return self.local.insert_reduce_plus(keys, vals) if reduce_plus else self.local.insert(keys, vals)
        chunks = max(1, int(chunks))
        cuda = keys.is_cuda            # host tensors (CPU test backends): the same piece loop without streams
        ex = None                      # the FIRST local failure of this call
        bounds = [n * i // chunks for i in range(chunks + 1)]
        plan = None
        begun = False
        sc_piece = [[0] * self.p for _ in range(chunks)]
        try:
            # ---- stage 1 (local): per-piece destination counts.  GPU backend: one sweep + one synchronisation for all pieces
            #      (kh_shard_plan), the pieces are then permuted without recounting
            try:
                self._inject(1)
                with self._span("count_pass"):
                    planned = self.b.shard_plan(keys, self.p, chunks) if hasattr(self.b, "shard_plan") else None
                    if planned is not None:
                        plan, bounds, sc_piece = planned
                    else:
                        sc_piece = [self.b.shard_counts(keys[bounds[i]:bounds[i + 1]], self.p) for i in range(chunks)]
            except Exception as e:
                ex = e
            # one exchange for the counts of all pieces (row = destination rank, column = piece); the status word of stage 1 with it
            rc, worst = self._exchange_counts([[sc_piece[i][r] for i in range(chunks)] for r in range(self.p)], ex)   # rc[src][piece]
            self._raise_if(ex, worst, "before the count exchange; nothing was exchanged")
            total = sum(sum(row) for row in rc)
            # ---- stage 2 (local): the receive side.  The received pieces are kept until the build has succeeded ("repeatable"):
            #      the local table may then partition them without a histogram pass into slots the pieces share; if that speculation
            #      fails (skewed / duplicated keys: KhRetry) the kept pieces are fed again the exact way
            try:
                self._inject(2)
                self.local.insert_begin(total, reduce_plus=reduce_plus, repeatable=True)
                begun = True
            except Exception as e:
                ex = e
            self._vote(ex, "while preparing to receive; nothing was exchanged")
            # ---- stage 3: the pieces.  A local failure is kept; the rank goes on exchanging and skips its local work
            kept = []
            cur = comm = None
            if cuda:
                cur = torch.cuda.current_stream(self.b.torch_device)
                if self._comm is None:
                    self._comm = torch.cuda.Stream(device=self.b.torch_device)
                comm = self._comm
            landed = None

            def feed(item):
                nonlocal ex
                if ex is not None:
                    return
                try:
                    if cuda:
                        cur.wait_event(item[0])
                    with self._span("feed"):
                        self.local.insert_feed(item[1], item[2])
                    kept.append((item[1], item[2]))
                except Exception as e:
                    ex = e

            for i in range(chunks):
                a, b = bounds[i], bounds[i + 1]
                scounts = sc_piece[i]
                ok = ov = None
                if ex is None:
                    try:
                        if i == 0:
                            self._inject(3)
                        with self._span("permute"):
                            if plan is not None:
                                ok, ov = plan.permute(i, vals, b - a)
                            else:
                                ok, ov, sc2 = self.b.shard(keys[a:b], vals[a:b] if vals is not None else None, self.p)     # stable permutation
                                assert list(sc2) == list(scounts), "count-only pass and permutation disagree"
                    except Exception as e:
                        ex = e
                if ok is None:             # failed: the peers still expect this rank's pairs -- they get a buffer of the right size
                    ok = self.b.empty(b - a, keys.dtype)
                    ov = self.b.empty(b - a, vals.dtype) if vals is not None else None
                rcounts = [rc[src][i] for src in range(self.p)]
                ev = None
                if cuda:
                    comm.wait_stream(cur)
                    with torch.cuda.stream(comm):
                        with self._span("exchange", comm):
                            rk, rv = self._exchange_grouped([ok, ov], scounts, rcounts)
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
                        rk, rv = self._exchange_grouped([ok, ov], scounts, rcounts)
                if landed is not None:                                 # piece i-1: partition it while piece i travels
                    feed(landed)
                landed = (ev, rk, rv)
                del ok, ov, rk, rv
            feed(landed)
            landed = None
            # ---- vote: did every rank send its real pairs and feed what it received?  If not, nobody builds (what a failed rank
            #      sent in place of its pairs must not reach any table)
            if ex is not None and cuda:
                torch.cuda.synchronize(self.b.torch_device)        # the exchanges this rank still took part in have drained
            self._vote(ex, "while the pieces were exchanged; nothing was inserted on any rank")
            # ---- stage 4: the build
            n_new = 0
            if ex is None:
                try:
                    self._inject(4)
                    with self._span("build"):
                        try:
                            n_new = self.local.insert_end()
                            begun = False
                        except Exception as e2:
                            begun = False
                            if type(e2).__name__ != "KhRetry":
                                raise
                            with self._span("refeed"):
                                self.local.insert_begin(total, reduce_plus=reduce_plus)
                                begun = True
                                for rk, rv in kept:
                                    self.local.insert_feed(rk, rv)
                                n_new = self.local.insert_end()
                                begun = False
                except Exception as e:
                    ex = e
            self._vote(ex, "in the build: the ranks that did not fail hold their share of the batch")
            return n_new
        finally:
            if begun and hasattr(self.local, "insert_abort"):          # an open streamed insert never outlives the call
                try:
                    self.local.insert_abort()
                except Exception:
                    pass
            if plan is not None:
                if cuda:
                    torch.cuda.current_stream(self.b.torch_device).synchronize()     # (queued permutations still read the plan)
                plan.close()

    def _my_query_pieces(self, keys, op):
        if op == "erase" or not hasattr(self.b, "shard_plan") or self.p > 8 or not keys.is_cuda:
            return 1
        if self.query_pieces > 0:
            return min(int(self.query_pieces), MAX_QUERY_PIECES)
        n = keys.numel()
        # measured over RCCL, one rank, self-exchange, 10^7 finds (scripts/dist_query_timing.py): 1.08 / 1.33 / 1.75 ms for 1 / 2 / 4 pieces --
        # ~0.22 ms of host work per extra piece in this layer (the C++ layer: 0.045 ms), more than the exchange of 10^7 keys can hide
        return 4 if n >= (1 << 27) else (2 if n >= (1 << 25) else 1)

    def _query(self, keys, op):
        """keys out (grouped by owner, in pieces), the local query of a piece while the next one travels, results back with the
        swapped counts to their place in the permuted order (khmxx::ialltoallv_and_query_one_to_one, incremental_mxx.hpp:4403-4669).
        Every rank chooses the number of pieces of ITS batch; the count exchange always carries MAX_QUERY_PIECES counts per
        destination plus that choice, and all ranks run as many rounds as the rank with the most pieces."""
        self._late_check()
        n = keys.numel()
        if self._single():
            with self._span("local_query"):
                if op == "count":
                    return keys, self.local.count(keys), None
                if op == "find":
                    v, f = self.local.find_values(keys)
                    return keys, v, f
                return None, self.local.erase(keys), None
        p, Q = self.p, MAX_QUERY_PIECES
        cuda = keys.is_cuda
        ex = None
        mine = self._my_query_pieces(keys, op)
        plan = None
        pk = None
        off = [[0] * (mine + 1) for _ in range(p)]
        try:
            # ---- stage 1 (local): the plan (one count sweep + synchronisation) or, without one, a stable permutation that counts
            try:
                self._inject(1)
                with self._span("permute" if mine == 1 else "count_pass"):
                    planned = self.b.shard_plan(keys, p, mine) if (mine > 1 or (cuda and hasattr(self.b, "shard_plan") and p <= 8)) else None
                    if planned is not None:
                        plan = planned[0]
                        off = plan.offsets(p, mine)
                        pk = self.b.empty(n, keys.dtype)
                    else:
                        pk, _, cnt = self.b.shard(keys, None, p)
                        o = self._offs(cnt)
                        off = [[o[r], o[r] + cnt[r]] for r in range(p)]
            except Exception as e:
                ex = e
                pk = self.b.empty(n, keys.dtype)
            sc = [[off[r][i + 1] - off[r][i] if i < mine else 0 for i in range(Q)] + [mine] for r in range(p)]
            rc, worst = self._exchange_counts(sc, ex)                   # rc[src][piece], rc[src][Q] = src's number of pieces
            self._raise_if(ex, worst, "before the count exchange; nothing was exchanged")
            rounds = max(1, max(min(int(r[Q]), Q) for r in rc))
            roff = [0]
            for i in range(rounds):
                roff.append(roff[-1] + sum(rc[s][i] for s in range(p)))
            rtot = roff[-1]
            # ---- stage 2 (local): the receive side
            rkeys = lv = lf = out_v = out_f = None
            try:
                self._inject(2)
                rkeys = self.b.empty(rtot, keys.dtype)
                if op != "erase":
                    lf = self.b.empty(rtot, torch.uint8)
                    out_f = self.b.empty(n, torch.uint8)
                    if op == "find":
                        lv = torch.zeros(rtot, dtype=torch.int32, device=rkeys.device)
                        out_v = torch.zeros(n, dtype=torch.int32, device=rkeys.device)
            except Exception as e:
                ex = e
            self._vote(ex, "while preparing to receive; nothing was exchanged")
            # ---- stage 3: everything is queued; a local failure is kept and the rank goes on exchanging
            cur = comm = None
            if cuda:
                cur = torch.cuda.current_stream(self.b.torch_device)
                if self._comm is None:
                    self._comm = torch.cuda.Stream(device=self.b.torch_device)
                comm = self._comm
            ev_p = []
            for i in range(rounds):
                if plan is not None and i < mine and ex is None:
                    try:
                        if i == 0:
                            self._inject(3)
                        with self._span("permute"):
                            plan.permute_global(i, pk)
                    except Exception as e:
                        ex = e
                elif plan is None and i == 0 and ex is None:
                    try:
                        self._inject(3)
                    except Exception as e:
                        ex = e
                if cuda:
                    e_ = torch.cuda.Event()
                    e_.record(cur)
                    ev_p.append(e_)
            scn = [[sc[r][i] for r in range(p)] for i in range(rounds)]
            sdn = [[off[r][i] if i < mine else 0 for r in range(p)] for i in range(rounds)]
            rcn = [[rc[s][i] for s in range(p)] for i in range(rounds)]
            ev_k = [None] * rounds

            def keys_out(i):
                ro = [roff[i] + x for x in self._offs(rcn[i])]
                if cuda:
                    comm.wait_event(ev_p[i])
                    with torch.cuda.stream(comm):
                        with self._span("exchange", comm):
                            self._exchange([(pk, sdn[i], scn[i], rkeys, ro, rcn[i])])
                        ev_k[i] = torch.cuda.Event()
                        ev_k[i].record(comm)
                else:
                    with self._span("exchange"):
                        self._exchange([(pk, sdn[i], scn[i], rkeys, ro, rcn[i])])

            n_erased = 0
            direct = cuda and hasattr(self.b, "shard_plan")        # the GPU table writes into slices of the result buffers
            keys_out(0)
            status_in = None
            for i in range(rounds):
                if i + 1 < rounds:
                    keys_out(i + 1)                                     # round i+1 travels while round i is looked up
                if cuda:
                    cur.wait_event(ev_k[i])
                a, b = roff[i], roff[i + 1]
                if op == "erase":      # did every rank send its real keys?  (what a failed rank sent instead must not be erased anywhere)
                    if ex is not None and cuda:
                        torch.cuda.synchronize(self.b.torch_device)
                    self._vote(ex, "while the keys were exchanged; nothing was erased on any rank")
                    try:
                        self._inject(4)
                    except Exception as e:
                        ex = e
                if ex is None:
                    try:
                        with self._span("local_query"):
                            if op == "count":
                                if direct:
                                    self.local.count(rkeys[a:b], out=lf[a:b])
                                else:
                                    lf[a:b] = self.local.count(rkeys[a:b])
                            elif op == "find":
                                if direct:      # straight into the result buffers: nothing allocated or copied, no wait for the device
                                    self.local.find_values(rkeys[a:b], out_vals=lv[a:b], out_found=lf[a:b], want_total=False)
                                else:
                                    v, f = self.local.find_values(rkeys[a:b])
                                    lv[a:b] = v
                                    lf[a:b] = f
                            else:
                                n_erased = self.local.erase(rkeys[a:b])
                    except Exception as e:
                        ex = e
                if op == "erase":
                    continue
                last = i == rounds - 1
                so = [a + x for x in self._offs(rcn[i])]
                arrays = []
                if op == "find":
                    arrays.append((lv, so, rcn[i], out_v, sdn[i], scn[i]))
                arrays.append((lf, so, rcn[i], out_f, sdn[i], scn[i]))
                if last:        # the status word of this rank's local work rides with the last result exchange
                    st_out = torch.full((1,), self._status_of(ex) if ex is not None else 0, dtype=torch.int64, device=rkeys.device)
                    status_in = torch.zeros(p, dtype=torch.int64, device=rkeys.device)
                    arrays.append((st_out, [0] * p, [1] * p, status_in, list(range(p)), [1] * p))
                if cuda:
                    comm.wait_stream(cur)
                    with torch.cuda.stream(comm):
                        with self._span("exchange", comm):
                            self._exchange(arrays)
                    for arr in arrays:
                        arr[0].record_stream(comm)
                else:
                    with self._span("exchange"):
                        self._exchange(arrays)
            if op == "erase":
                self._vote(ex, "in the local erase: the ranks that did not fail erased their share")
                return None, n_erased, None
            if cuda:
                cur.wait_stream(comm)       # the results are complete in the order of the caller's stream; nothing was waited for on the host
                for x in (rkeys, lv, lf, pk):
                    if x is not None:
                        x.record_stream(comm)
            # the peers' status words: host exchanges are complete already (every rank raises inside this call); on the GPU they are
            # looked at by synchronize() or by the next collective call -- on EVERY rank, the failing one included (its own word is
            # among them), so that all ranks skip that call together.  Call synchronize() on every rank or on none.
            self._late = status_in
            if not cuda:
                if ex is not None:
                    self._late = None
                    raise ex
                self._late_check()
            elif ex is not None:
                raise ex
            return pk, (out_v if op == "find" else out_f), (out_f if op == "find" else None)
        finally:
            if plan is not None:
                if cuda:
                    torch.cuda.current_stream(self.b.torch_device).synchronize()
                plan.close()

    def count(self, keys):
        """count_p :1258: results come back in the PERMUTED input order (grouped by owner rank), like the
        reference, together with the permuted keys.  Queued only: see synchronize()."""
        pk, flags, _ = self._query(keys, "count")
        return pk, flags

    def find(self, keys):
        """find_p :1619: (permuted keys, values, found flags) aligned with the permuted keys; values and flags return in one
        exchange per piece.  Queued only: see synchronize()."""
        return self._query(keys, "find")

    def erase(self, keys):
        """erase_p :2169: returns the number erased on this rank's local table"""
        return self._query(keys, "erase")[1]

    def size(self):
        """global size = sum of local sizes"""
        self._late_check()
        n = self.local.size()
        if self._single():
            return n
        t = torch.tensor([n], dtype=torch.int64, device=self._ctl_device())
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def local_size(self):
        return self.local.size()
