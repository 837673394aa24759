"""GPU: every RCCL call of the torch.distributed sharded table executed on the one GPU of the test box.  A process group of ONE
rank over backend "nccl" (= RCCL) with KH_DIST_FORCE_COLLECTIVES=1: ShardedTable then takes the multi-rank code path -- all_to_all_single
of the counts, the grouped batch_isend_irecv payload exchanges (the self segment travels through RCCL too), the all_reduce votes, the
pipelined insert and the pipelined queries on the comm stream -- and must give the plain table's results.  (The C++ layer's RCCL
calls run the same way in tests/cpp/test_dist.cpp.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["KH_DIST_FORCE_COLLECTIVES"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import kmerhash_amd as kh
        from kmerhash_amd import workloads as W
        from kmerhash_amd import dist as khd
        assert khd.FORCE_COLLECTIVES
        n = 4_500_000                                     # capacity 2^23: 2^12 partitions, the histogram-free layout of the pieces
        keys = W.distinct_u64(n, seed=5)
        vals = np.arange(n, dtype=np.uint32)
        keys[n - 4000:] = keys[100:4100]              # duplicates the first piece's sample cannot see: KhRetry inside the sharded insert
        dk = torch.from_numpy(keys.view(np.int64)).cuda()
        dv = torch.from_numpy(vals.view(np.int32)).cuda()
        st = khd.ShardedTable(khd.GpuBackend(0), timing=True)
        assert not st._single()
        plain = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
        assert st.insert(dk, dv, chunks=4) == plain.insert(dk, dv) == n - 4000
        assert st.collectives == {"counts": 1, "payload": 4, "votes": 3}, st.collectives
        assert "refeed" in st.timings()
        assert st.size() == plain.size() and st.local.capacity() == plain.capacity()
        assert np.array_equal(st.local.export_info(), plain.export_info())
        a, b = st.local.sorted_items(), plain.sorted_items()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # pipelined queries (3 pieces) over RCCL: permuted order of ONE rank == input order
        q1 = np.concatenate([keys[:900_000], W.distinct_u64(300_000, seed=77)])
        q1 = q1[W.shuffle_perm(len(q1), 3)]
        dq = torch.from_numpy(q1.view(np.int64)).cuda()
        st.query_pieces = 3
        c0 = dict(st.collectives)
        pk, fv, ff = st.find(dq)
        st.synchronize()
        assert st.collectives == {"counts": c0["counts"] + 1, "payload": c0["payload"] + 6, "votes": c0["votes"] + 1}, st.collectives
        pv, pf = plain.find_values(dq)
        assert torch.equal(pk, dq) and torch.equal(ff, pf) and torch.equal(fv[ff == 1], pv[pf == 1])
        pk2, cnt = st.count(dq)
        st.synchronize()
        assert torch.equal(cnt, pf)
        st.query_pieces = 0
        pk3, cnt3 = st.count(dq[:5000])
        st.synchronize()
        assert torch.equal(cnt3, pf[:5000])
        assert st.erase(dq) == plain.erase(dq)
        assert st.size() == plain.size()
        a, b = st.local.sorted_items(), plain.sorted_items()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # counting insert through the same path
        sc = khd.ShardedTable(khd.GpuBackend(0, hash="farm"))
        pc = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash="farm")
        assert sc.insert_counts(dk, chunks=3) == pc.insert_reduce_plus(dk)
        a, b = sc.local.sorted_items(), pc.sorted_items()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # a local failure at every stage: raised, nothing left half-open, the table usable afterwards
        for op, stages in (("insert", (1, 2, 3, 4)), ("find", (1, 2, 3)), ("erase", (1, 2, 3, 4))):
            for stage in stages:
                st._fail_stage = stage
                with pytest.raises(MemoryError):
                    if op == "insert":
                        st.insert(dk[:200_000], dv[:200_000], chunks=2)
                    elif op == "find":
                        st.find(dq)
                        st.synchronize()
                    else:
                        st.erase(dq[:1000])
                if op == "find" and stage == 3:     # (the words of the failed find are still pending: the next collective call reports them)
                    with pytest.raises(khd.ShardPeerError):
                        st.size()
                assert st.size() == plain.size()
        st.insert(dk[:200_000], dv[:200_000], chunks=2)
        plain.insert(dk[:200_000], dv[:200_000])
        a, b = st.local.sorted_items(), plain.sorted_items()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        q.put("ok")
    except Exception:  # pragma: no cover
        import traceback
        q.put("FAIL: " + traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_table_over_rccl_one_rank_forced_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=500)
    p.join(60)
    assert res == "ok", res
