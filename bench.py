#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X (BASELINE.json configs[1] at N=1).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one synthetic batch, inputs already resident in HBM:
    fresh table (capacity 128, min/max load 0.35/0.8, murmur3avx64 seed 43)
    insert  KEYS_PER_GPU random DISTINCT 64-bit k-mers with 32-bit values (table doubles up to 2^27: load 0.745)
    find    QUERIES_PER_GPU keys (all hits), compacted (key,value) result in query order
This is the benchmark_hashmap phase sequence (BenchmarkHashTables.cpp:1037-1186) restricted to the two
rates BASELINE.json's metric names.  N>1 (weak scaling): every rank generates its own KEYS_PER_GPU pairs,
keys are sharded by murmur3(key, seed 9876543) & (N-1), exchanged with RCCL all_to_all_single, and inserted
into the owner's local table; finds travel the same way and results return with the swapped counts.

Rank 0 prints ONE JSON line: metric/value = whole-job k-mer operations (inserts + finds) per second, plus
the two individual rates, the roofline of the dominant kernel (HIP-event timed inside the library on the
table's stream) and a CPU baseline of the same workload on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEYS_PER_GPU = 100_000_000
QUERIES_PER_GPU = 10_000_000
# algorithmic bytes per operation (AoS-equivalent sizes of the reference)
B_INSERT_NEW, B_INSERT_DUP, B_FIND_HIT, B_FIND_MISS = 50, 33, 41, 9
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def gen_inputs(rank, n, nq, workload="w2"):
    from kmerhash_amd import workloads as W
    if workload == "w1":                             # benchmark_hashtables shape: mean multiplicity 5.5 (not the metric's config)
        keys, vals = W.w1_benchmark_hashtables(n, seed=23 + rank)
        return keys, vals, keys[:nq].copy()
    keys = W.distinct_u64(n, seed=1 + rank)          # W2: distinct uniform u64 (bijective splitmix64 of a counter)
    vals = np.arange(n, dtype=np.uint32)
    q = keys[:nq].copy()                             # all hits (BenchmarkHashTables.cpp:1062-1066: first N/Q inputs)
    return keys, vals, q


def _cpu_shard_worker(path, start_at=0.0):
    """one 'rank' of the sharded CPU baseline: private oracle table over its share of the sample (own process, like the
    reference's MPI ranks: page faults of the doubling tables do not contend on one address space).  Prints one JSON line."""
    from oracle import oracle_py as O
    d = np.load(path)
    k, v, qq = d["k"], d["v"], d["q"]
    t = O.OracleTable(O.KIND_RH, 128, 0.35, 0.8, O.HASH_MURMUR3_X86, 43)
    while time.time() < start_at:              # common start time: all ranks run concurrently
        time.sleep(0.001)
    t0 = time.time()
    ti = t.timed_insert(k, v)
    tf = t.timed_find(qq)[0]
    print(json.dumps([ti, tf, t0, time.time()]), flush=True)


def _cpu_sharded(keys, vals, q, P):
    import shutil
    import subprocess
    import tempfile
    from oracle import oracle_py as O
    O.lib()                                    # compiled before the workers start
    r = (O.hash_batch(O.HASH_MURMUR3_X86, 9876543, keys) % np.uint64(P)).astype(np.int32)
    rq = (O.hash_batch(O.HASH_MURMUR3_X86, 9876543, q) % np.uint64(P)).astype(np.int32)
    tmp = tempfile.mkdtemp(prefix="kh_cpu_")
    procs = []
    try:
        for i in range(P):
            path = os.path.join(tmp, "s%d.npz" % i)
            np.savez(path, k=keys[r == i], v=vals[r == i], q=q[rq == i])
        # plain child processes (fresh interpreters), started before this process touches the GPU
        code = "import sys; sys.path.insert(0, %r); import bench; bench._cpu_shard_worker(sys.argv[1], float(sys.argv[2]))" % ROOT
        start_at = time.time() + 4.0
        for i in range(P):
            procs.append(subprocess.Popen([sys.executable, "-c", code, os.path.join(tmp, "s%d.npz" % i), repr(start_at)],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True))
        res = []
        for pr in procs:
            o, _ = pr.communicate(timeout=300)
            if pr.returncode != 0:
                raise RuntimeError("cpu shard worker failed (rc %d)" % pr.returncode)
            res.append(json.loads(o.strip().splitlines()[-1]))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(tmp, ignore_errors=True)
    wi, wf = max(a for a, _, _, _ in res), max(b for _, b, _, _ in res)
    overlap = min(e for _, _, _, e in res) - max(s for _, _, s, _ in res)     # > 0: all ranks ran concurrently
    return {"cores": P, "value": (len(keys) + len(q)) / (wi + wf), "inserts_per_s": len(keys) / wi, "finds_per_s": len(q) / wf,
            "concurrent": bool(overlap > 0),
            "sample": "first %d inserts + %d finds of the same stream split over %d private tables by murmur3(key, 9876543) %% %d, "
                      "one process each, common start (the reference's MPI model without the exchange)" % (len(keys), len(q), P, P)}


def cpu_baseline(keys, vals, q):
    """the CPU oracle (own restatement of the reference RH table: kind 'port') timed on the host cores, on a bounded
    sample of the same stream.  value: ONE thread (the reference is single-threaded per rank: the benchmark_hashtables
    number).  'sharded': the reference's MPI model without the communication: P threads, thread r owns
    the keys with murmur3(key, seed 9876543) % P == r in a private table; aggregate rate over the slowest thread."""
    import threading
    from oracle import oracle_py as O
    n = min(len(keys), 20_000_000)
    nq = min(len(q), 2_000_000)
    t = O.OracleTable(O.KIND_RH, 128, 0.35, 0.8, O.HASH_MURMUR3_X86, 43)
    ti = t.timed_insert(keys[:n], vals[:n])
    tf, hits = t.timed_find(q[:nq])
    assert hits == nq
    out = {"value": (n + nq) / (ti + tf), "unit": "kmer_ops/s", "cores": 1, "kind": "port",
           "sample": "first %d inserts + %d finds of the same stream, oracle RH table (1 thread, g++ -O3)" % (n, nq),
           "inserts_per_s": n / ti, "finds_per_s": nq / tf}
    del t
    # the REAL reference where it could be compiled (oracle/_ref: fsc::hashmap_linearprobe_doubling built from the reference
    # tree as it lies; the Robin Hood header needs an absent kmerind header): same sample, same hash, one thread -- shows that
    # the port's speed is representative of the reference's own tables
    try:
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_lp.so")):
            r = O.RefLPTable(128, 0.35, 0.8, O.HASH_MURMUR3_X86, 43)
            ri = r.timed_insert(keys[:n], vals[:n])
            rc, rhits = r.timed_count(q[:nq])
            assert rhits == nq
            out["reference_lp"] = {"kind": "reference", "cores": 1, "inserts_per_s": n / ri, "counts_per_s": nq / rc,
                                   "sample": "the same sample through the reference's own hashmap_linearprobe_doubling (insert + count)"}
            del r
    except Exception as e:
        out["reference_lp"] = {"error": repr(e)}
    try:
        P = max(1, min(len(os.sched_getaffinity(0)), 16))       # the GPU box's CPU share for one GPU is 16 cores
        if P > 1:
            ns, nqs = min(len(keys), 2 * n), min(len(q), 2 * nq)    # the ranks run in parallel: a larger sample fits the time budget
            out["sharded"] = _cpu_sharded(keys[:ns], vals[:ns], q[:nqs], P)
    except Exception as e:                    # the single-thread figure stands on its own
        out["sharded"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--keys", type=int, default=KEYS_PER_GPU)
    ap.add_argument("--queries", type=int, default=QUERIES_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="w2", choices=["w2", "w1"],
                    help="w2 = configs[1] (default, the metric's workload); w1 = benchmark_hashtables shape (x5.5 multiplicity), informational")
    ap.add_argument("--hash", default="murmur3avx64", choices=["murmur3avx64", "murmur", "farm", "identity"],
                    help="storage hash (default = the metric's: murmur3avx64; the others are informational)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N>1: pieces of the pipelined exchange/insert (permute, xGMI transfer and radix partition of successive pieces "
                         "overlap); 1 = exchange, then one bulk insert; 0 = auto (4 when N>1: the exchange is link-bound at every N)")
    args = ap.parse_args()

    # everything libraries write to stdout (RCCL prints a version banner there at communicator creation) goes to stderr,
    # so that stdout carries exactly ONE line: the JSON result
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("KH_DIST_FORCE_COLLECTIVES", "0") == "1"   # rehearsal of the N>1 path on one GPU
    keys, vals, q = gen_inputs(rank, args.keys, args.queries, args.workload)
    # CPU baseline first (rank 0, N=1 only): its worker processes are started before this process touches the GPU
    cpu = cpu_baseline(keys, vals, q) if (not args.no_cpu_baseline and not distributed) else None

    import torch
    import kmerhash_amd as kh
    from kmerhash_amd import dist as khd
    if args.chunks <= 0:
        args.chunks = 4 if world > 1 else 1
    if distributed and "RANK" not in os.environ:
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    n_distinct = len(np.unique(keys)) if args.workload == "w1" else args.keys
    dk = torch.from_numpy(keys.view(np.int64)).to(dev)
    dv = torch.from_numpy(vals.view(np.int32)).to(dev)
    dq = torch.from_numpy(q.view(np.int64)).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    prof = {}
    ins_ms, find_ms = [], []

    def one_step(timed):
        if distributed:
            be = khd.GpuBackend(local_rank, "rh", 128, 0.35, 0.8, args.hash, 43)
            t = khd.ShardedTable(be)
            table = be.table
        else:
            table = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=args.hash, seed=43, device=local_rank)
            t = None
        if timed:
            table.profile_enable(True)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        if distributed:
            n_ins = t.insert(dk, dv, chunks=args.chunks)
        else:
            n_ins = table.insert(dk, dv)
        e1.record()
        if distributed:
            _, fvals, ffound = t.find(dq)
            n_hit = int(ffound.sum().item())
        else:
            fk, fv = table.find(dq)
            n_hit = fk.numel()
        e2.record()
        torch.cuda.synchronize()
        if timed:
            ins_ms.append(e0.elapsed_time(e1))
            find_ms.append(e1.elapsed_time(e2))
            for k, (n, ms) in table.profile().items():
                a = prof.setdefault(k, [0, 0.0])
                a[0] += n
                a[1] += ms
        state = (n_ins, n_hit, table.size(), table.capacity())
        table.close()
        return state

    for w in range(args.warmup):
        if distributed and w == 0 and args.chunks > 1:
            # safety net for the pipelined exchange (it cannot be rehearsed with real peers on a one-GPU box): if any rank fails
            # its first warm-up step, every rank falls back to exchange-then-insert for the rest of the run
            ok = 1
            try:
                one_step(False)
            except Exception as ex:           # noqa: BLE001
                print("[bench] rank %d: pipelined insert failed (%r), falling back to --chunks 1" % (rank, ex), file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                args.chunks = 1
                one_step(False)
            continue
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    state = None
    for _ in range(args.steps):
        state = one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tot = torch.tensor([state[0], state[1], state[2]], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        g_ins, g_hit, g_size = (int(x) for x in tot.cpu())
    else:
        g_ins, g_hit, g_size = state[0], state[1], state[2]

    # size-independent parity properties at full size (the oracle cannot run 1e8 keys in seconds):
    # every distinct key inserted exactly once, every query found
    if args.workload == "w2":
        assert g_ins == args.keys * world, (g_ins, args.keys * world)
        assert g_size == args.keys * world
    elif not distributed:
        assert g_ins == n_distinct and g_size == n_distinct, (g_ins, n_distinct)
    assert g_hit == args.queries * world, (g_hit, args.queries * world)

    if rank == 0:
        ops_per_step = (args.keys + args.queries) * world
        ms_per_step = elapsed / args.steps * 1e3
        ins_rate = args.keys * world / (np.mean(ins_ms) * 1e-3)
        find_rate = args.queries * world / (np.mean(find_ms) * 1e-3)
        # dominant kernel of the insert path, timed with HIP events inside the library on the table's stream
        dom = max((k for k in prof if k.startswith("k_")), key=lambda k: prof[k][1] / max(prof[k][0], 1) * (prof[k][0] / args.steps))
        launches, total_ms = prof[dom]
        per_launch_ms = total_ms / launches
        units = {"k_find": args.queries, "k_count": args.queries}.get(dom, args.keys)
        bpu = {"k_find": B_FIND_HIT}.get(dom, B_INSERT_NEW)
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes of this same command
        # (profiles/<tag>_pmc_hbm_traffic.json, FETCH_SIZE/WRITE_SIZE collected and corrected as MI355X_MICROARCH.md
        # prescribes); PMC counters cannot be read from inside the timed process, so this is the recorded figure
        traffic = None
        try:
            tag = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()      # written by scripts/summarize_profiles.py
            traffic_src = "%s_pmc_hbm_traffic.json" % tag
            if not distributed:
                traffic = json.load(open(os.path.join(ROOT, "profiles", traffic_src))).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        launches_per_step = launches / args.steps
        alg_bytes = units * bpu / max(launches_per_step, 1.0)          # algorithmic bytes one launch accounts for
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        out = {
            "metric": "kmer_inserts_plus_finds_per_sec",
            "value": ops_per_step / (elapsed / args.steps),
            "unit": "kmer_ops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("(informational, benchmark_hashtables shape x5.5 multiplicity) " if args.workload == "w1" else "") + "configs[1]: Robin Hood table, %d distinct random 64-bit k-mers per GPU (max load 0.8 -> "
                                   "capacity %d, load %.3f), murmur3avx64 seed 43, then %d all-hit finds per GPU%s"
                                   % (args.keys, state[3], state[2] / state[3], args.queries,
                                      "; keys sharded by murmur3(seed 9876543) over RCCL all_to_all" if distributed else ""),
                       "keys_per_gpu": args.keys, "queries_per_gpu": args.queries, "table": "hashmap_robinhood_doubling",
                       "hash": args.hash, "max_load_factor": 0.8, "min_load_factor": 0.35},
            "inserts_per_s": ins_rate, "finds_per_s": find_rate,
            "insert_ms": float(np.mean(ins_ms)), "find_ms": float(np.mean(find_ms)),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": (traffic_src if traffic else None),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": per_launch_ms,
                         "whole_insert_path_frac": ins_rate / world * B_INSERT_NEW / 1e9 / HBM_PEAK_GBS,
                         "whole_find_path_frac": find_rate / world * B_FIND_HIT / 1e9 / HBM_PEAK_GBS},
            "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
