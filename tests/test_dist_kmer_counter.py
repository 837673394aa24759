"""CPU, gloo world 2: BASELINE configs[4] in miniature -- every rank parses its own FASTQ batches into canonical 31-mers, the
k-mers are sharded by hash over the ranks and counted (std::plus) in the owner's table, which starts at capacity 128 and
doubles under load; then the count / find / erase / count cycle of BenchmarkKmerIndex.cpp:787-843.  The device pieces are
replaced by the oracle (table) and the numpy k-mer statement (oracle/kmers_np.py); the code under test is
kmerhash_amd.kmers.ShardedKmerCounter over kmerhash_amd.dist.ShardedTable, what the GPU ranks run over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_py as O
        from oracle.kmers_np import np_kmers_fastq
        from kmerhash_amd import kmers as KM
        from kmerhash_amd.dist import ShardedTable, DIST_SEED
        from test_dist_gloo import OracleBackend
        k, read_len, n_reads = 31, 100, 600
        fq = KM.synthetic_fastq(n_reads * world, read_len, 9_000, seed=11, n_rate=0.002)      # one file; every rank takes its share of the reads
        recs = fq.split(b"\n")
        per = 4 * n_reads
        mine = b"\n".join(recs[rank * per:(rank + 1) * per]) + b"\n"
        kmer_fn = lambda text: torch.from_numpy(np_kmers_fastq(text, k, True).view(np.int64).copy())
        be = OracleBackend(O, O.KIND_RH)
        be.table.t.close()
        be.table.t = O.OracleTable(O.KIND_RH, 128, 0.35, 0.8, O.HASH_FARM, 43)               # farmhash storage hash, capacity 128: doubles under load
        kc = KM.ShardedKmerCounter(ShardedTable(be), k, True, kmer_fn=kmer_fn, chunks=2)

        # --hll-reserve at p > 1: every rank's registers merged by an all-reduce(max), the GLOBAL estimate / p reserved on every rank
        # (hyperloglog64.hpp:477-489).  The oracle HLL stands in for the GPU one; the estimate formula is the library's (host code).
        class Hll:
            precision = 12
            est_error_rate = 1.04 / 64.0

            def __init__(s):
                s.o = O.OracleHLL(12, 0, O.HASH_FARM, 43)

            def update(s, km):
                s.o.update(km.numpy().view(np.uint64))

            def registers(s):
                return s.o.registers()
        from kmerhash_amd import hll as HL
        be2 = OracleBackend(O, O.KIND_RH)
        be2.table.reserve = lambda n: be2.table.t.reserve(int(n))
        hl = Hll()
        kc2 = KM.ShardedKmerCounter(ShardedTable(be2), k, True, kmer_fn=kmer_fn, chunks=1, reserve_from_estimate=True, hll=hl)
        lines = mine.split(b"\n")
        nb = 3                                                                             # three file batches, cut at record boundaries
        total = 0
        for b in range(nb):
            part = lines[4 * (n_reads * b // nb): 4 * (n_reads * (b + 1) // nb)]
            total += kc.add_fastq(b"\n".join(part) + b"\n")
        assert be.table.t.capacity() > 128
        kc2.add_fastq(mine)
        # the merged registers are those of ONE estimator fed every rank's k-mers; all ranks computed the same global estimate
        allmine = [None] * world
        dist.all_gather_object(allmine, np_kmers_fastq(mine, k, True))
        one = O.OracleHLL(12, 0, O.HASH_FARM, 43)
        for a in allmine:
            one.update(a)
        g_est = HL.estimate_global(hl)
        assert g_est == one.estimate() == HL.estimate_from_registers(one.registers(), 12)
        n_dist = len(np.unique(np.concatenate(allmine)))
        assert abs(g_est - n_dist) < 0.08 * n_dist
        assert be2.table.t.capacity() >= int(g_est / world / 0.8)            # reserved its share before inserting
        # model: counts of all k-mers of the whole file, each on its owner rank
        allk = np_kmers_fastq(fq, k, True)
        tot = torch.tensor([total]); dist.all_reduce(tot)
        assert int(tot.item()) == len(allk)
        uk, cnt = np.unique(allk, return_counts=True)
        owner = (O.hash_batch(O.HASH_MURMUR3_X86, DIST_SEED, uk) % np.uint64(world)).astype(np.int64)
        ck, cv = be.table.t.sorted_items()
        assert np.array_equal(ck, uk[owner == rank]) and np.array_equal(cv, cnt[owner == rank].astype(np.uint32))
        assert kc.size() == len(uk)
        # the query phase: a sample of this rank's own k-mers plus k-mers that do not occur
        myk = np_kmers_fastq(mine, k, True)
        sample = np.concatenate([myk[::7], (np.arange(50, dtype=np.uint64) << np.uint64(40)) | np.uint64(0x155)])
        exp_hits = int(np.isin(sample, uk).sum())
        lut = dict(zip(uk.tolist(), cnt.tolist()))
        exp_occ = sum(lut.get(int(x), 0) for x in sample.tolist())
        res = kc.cycle(torch.from_numpy(sample.view(np.int64).copy()))
        assert res["count_hits"] == res["find_hits"] == exp_hits
        assert res["find_occurrences"] == exp_occ
        assert res["count_hits_after"] == 0
        allq = [None] * world
        dist.all_gather_object(allq, sample)
        gone = np.intersect1d(np.unique(np.concatenate(allq)), uk)
        er = torch.tensor([res["erased_local"]]); dist.all_reduce(er)
        assert int(er.item()) == len(gone) and kc.size() == len(uk) - len(gone)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_kmer_counter_gloo(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res
