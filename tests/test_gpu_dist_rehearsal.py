"""GPU: rehearsal of the N>1 path with TWO ranks sharing the one GPU of the test box.  RCCL refuses two ranks on one
device, so the collectives run over gloo (host-staged in ShardedTable); everything else -- kh_shard_permute, split
sizes, the local GPU tables, queries returning with the swapped counts, chunked overlap -- is the code the RCCL ranks
run.  Checked against the single-table CPU model (receive order = source rank, then position)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_py as O
        from kmerhash_amd import workloads as W
        from kmerhash_amd.dist import GpuBackend, ShardedTable, DIST_SEED, plan_piece_bounds
        torch.cuda.set_device(0)
        n = 400_000
        keys, vals = W.w1_benchmark_hashtables(n, seed=100)
        p = W.shuffle_perm(n, 7 + rank)
        keys, vals = keys[p], (vals[p] + np.uint32(rank * 10_000_000))
        dk = torch.from_numpy(keys.view(np.int64).copy()).cuda()
        dv = torch.from_numpy(vals.view(np.int32).copy()).cuda()
        for chunks in (1, 3):
            st = ShardedTable(GpuBackend(0))
            st.insert(dk, dv, chunks=chunks)
            allk, allv = [None] * world, [None] * world
            dist.all_gather_object(allk, keys)
            dist.all_gather_object(allv, vals)
            owner = lambda k: (O.hash_batch(O.HASH_MURMUR3_X86, DIST_SEED, k) % np.uint64(world)).astype(np.int64)
            model = O.OracleTable(O.KIND_RH, 128, 0.35, 0.8, O.HASH_MURMUR3_X86, 43)
            if chunks == 1:
                for r in range(world):
                    m = owner(allk[r]) == rank
                    model.insert(allk[r][m], allv[r][m])
            else:   # piece i of every rank arrives before piece i+1 of any rank
                b = plan_piece_bounds(n, chunks)          # (GPU backend, <= 8 ranks: the pieces are cut at multiples of 4096 pairs)
                for i in range(chunks):
                    for r in range(world):
                        kk, vv = allk[r][b[i]:b[i + 1]], allv[r][b[i]:b[i + 1]]
                        m = owner(kk) == rank
                        model.insert(kk[m], vv[m])
            loc = st.local
            assert (loc.size(), loc.capacity()) == (model.size(), model.capacity())
            assert np.array_equal(loc.export_info(), model.export_info())
            a, bb = loc.sorted_items(), model.sorted_items()
            assert np.array_equal(a[0], bb[0]) and np.array_equal(a[1], bb[1])
            assert st.size() == len(np.unique(np.concatenate(allk)))
            qk = np.concatenate([keys[:20_000], W.distinct_u64(20_000, seed=55 + rank)])
            universe = np.unique(np.concatenate(allk))
            pk, cnt = st.count(torch.from_numpy(qk.view(np.int64).copy()).cuda())
            exp = np.isin(pk.cpu().numpy().view(np.uint64), universe).astype(np.uint8)
            assert np.array_equal(cnt.cpu().numpy(), exp)
            pk2, fv, ff = st.find(torch.from_numpy(qk.view(np.int64).copy()).cuda())
            assert np.array_equal(ff.cpu().numpy(), exp)
            # counting insert through the same (pipelined) exchange: global multiplicities on the owner rank
            sc = ShardedTable(GpuBackend(0))
            sc.insert_counts(dk, chunks=chunks)
            uk, ucnt = np.unique(np.concatenate(allk), return_counts=True)
            mine = owner(uk) == rank
            ck, cv = sc.local.sorted_items()
            assert np.array_equal(ck, uk[mine]) and np.array_equal(cv, ucnt[mine].astype(np.uint32))
            sc.local.close()
            ne = st.erase(dk[:5000])
            tot = torch.tensor([ne])
            dist.all_reduce(tot)
            assert int(tot.item()) == len(np.unique(np.concatenate([x[:5000] for x in allk])))
            loc.close()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_rehearsal():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
