#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X (BASELINE.json configs[1] at N=1).

  python bench.py --gpus N --steps K --warmup W

Launch model.  Run plainly (no WORLD_SIZE in the environment) this process is the LAUNCHER: it never touches a GPU; it
checks that N devices are visible (exit 2 otherwise -- never a silent fall-back to fewer ranks), runs the CPU baseline
legs (N == 1 only), then starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
MASTER_PORT set, one per device), relays rank 0's JSON line and propagates the first non-zero exit code.  Run under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (WORLD_SIZE set) the process IS a rank and
WORLD_SIZE must equal --gpus.  The reference's own multi-rank driver is started by mpirun and partitions by comm.size()
(benchmark/BenchmarkDistHashTables.cpp:908-936, distributed_batched_robinhood_map.hpp:910-1194).

A step = one pass of the hot path over one synthetic batch, inputs already resident in HBM:
    fresh table (capacity 128, min/max load 0.35/0.8, murmur3avx64 seed 43)
    insert  KEYS random DISTINCT 64-bit k-mers with 32-bit values; N == 1: KEYS = 107 374 184 = max_load(2^27, 0.8f), so the
            table ends at capacity 2^27 and load exactly 0.800 (SURVEY 8d W2, N'); N > 1: 10^8 per rank (the shard a rank
            receives is 10^8 +- ~10^4 keys, which must stay under max_load for every rank to do the same work: load 0.745)
    find    QUERIES keys (all hits), compacted (key,value) result in query order
This is the benchmark_hashmap phase sequence (BenchmarkHashTables.cpp:1037-1186) restricted to the two rates
BASELINE.json's metric names.  N > 1 (weak scaling): every rank generates its own pairs, keys are sharded by
murmur3(key, seed 9876543) & (N-1), exchanged over RCCL and inserted into the owner's local table (pipelined: exchange of
piece i overlaps the radix partition of piece i-1); finds travel the same way and results return with the swapped counts.

Rank 0 prints ONE JSON line: metric/value = whole-job k-mer operations (inserts + finds) per second, the two individual
rates, `roofline` (achieved = ops/s x algorithmic bytes/op over 8 TB/s: `frac` = the insert path, inserts/s x 50 B;
the find path, finds/s x 41 B, and the lower of the two are reported next to it) and, at N == 1, `cpu_baseline` (the oracle port and the compiled reference LP table on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# KH_BENCH_REHEARSAL=1: N ranks share GPU 0 and talk over gloo (RCCL refuses two ranks on one device).  Exercises the launcher, the
# barriers, the sharded insert / find and the result reduction of the N > 1 path on a one-GPU box; the JSON line says "rehearsal".
REHEARSAL = os.environ.get("KH_BENCH_REHEARSAL", "0") == "1"
KEYS_LOAD_080 = 107_374_184        # size_t(float(2^27) * 0.8f): hashmap_robinhood.hpp:263; load exactly 0.800
KEYS_PER_GPU_DIST = 100_000_000
QUERIES_PER_GPU = 10_000_000
# algorithmic bytes per operation (AoS-equivalent sizes of the reference) and the sector-granular figures
# (every random touch moves >= 64 B): the realistic random-access bound
B_INSERT_NEW, B_INSERT_DUP, B_FIND_HIT, B_FIND_MISS = 50, 33, 41, 9
B_INSERT_SECTOR, B_FIND_SECTOR = 16 + 3 * 64, 8 + 2 * 64 + 16
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def gen_inputs(rank, n, nq, workload="w2"):
    from kmerhash_amd import workloads as W
    if workload == "w1":                             # benchmark_hashtables shape: mean multiplicity 5.5 (not the metric's config)
        keys, vals = W.w1_benchmark_hashtables(n, seed=23 + rank)
        return keys, vals, keys[:nq].copy()
    keys = W.distinct_u64(n, seed=1 + rank)          # W2: distinct uniform u64 (bijective splitmix64 of a counter)
    vals = np.arange(n, dtype=np.uint32)
    q = keys[:nq].copy()                             # all hits (BenchmarkHashTables.cpp:1062-1066: first N/Q inputs)
    return keys, vals, q


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (launcher process, N == 1 only; test infrastructure from oracle/ used as the thing timed BESIDE the product)
# ---------------------------------------------------------------------------------------------------------------------
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


def host_cores():
    """physical cores of the host (distinct (physical id, core id) pairs of /proc/cpuinfo), logical CPUs, and the CPUs this process may
    run on (SURVEY 8d-ii asks for the physical count next to the sharded figure)"""
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        if pid is not None and cid is not None:
            phys.add((pid, cid))
    except Exception:
        pass
    return {"physical": len(phys) or None, "logical": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}


def cpu_baseline(keys, vals, q):
    """the CPU oracle (own restatement of the reference RH table: kind 'port') timed on the host cores, on a bounded
    sample of the same stream.  value: ONE thread (the reference is single-threaded per rank: the benchmark_hashtables
    number).  'sharded': the reference's MPI model without the communication: P processes, process r owns
    the keys with murmur3(key, seed 9876543) % P == r in a private table; aggregate rate over the slowest process."""
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
    out["host_cores"] = host_cores()
    try:
        P = max(1, min(len(os.sched_getaffinity(0)), 16))       # the GPU box's CPU share for one GPU is 16 cores
        if P > 1:
            ns, nqs = min(len(keys), 2 * n), min(len(q), 2 * nq)    # the ranks run in parallel: a larger sample fits the time budget
            out["sharded"] = _cpu_sharded(keys[:ns], vals[:ns], q[:nqs], P)
            out["sharded"]["host_physical_cores"] = out["host_cores"]["physical"]
            out["sharded"]["cores_note"] = "P = min(CPUs this process may run on, 16: one GPU's share of the box); the host has %s physical cores" % out["host_cores"]["physical"]
    except Exception as e:                    # the single-thread figure stands on its own
        out["sharded"] = {"error": repr(e)}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY 8d timing protocol beyond the headline (N == 1, after the timed region; informational keys of the JSON line)
# ---------------------------------------------------------------------------------------------------------------------
B_COUNT, B_ERASE = 26, 42          # SURVEY 8d algorithmic bytes: count 8 + 17 + 1, erase (hit) 8 + 17 + 17


def _srl(x, s):
    """logical right shift of an int64 tensor"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _splitmix64_t(x):
    """workloads.splitmix64 on an int64 CUDA tensor (wrapping arithmetic; bit-identical to the numpy generator)"""
    import torch
    c = lambda v: torch.tensor(np.array([v], dtype=np.uint64).view(np.int64)[0], dtype=torch.int64, device=x.device)
    z = x + c(0x9E3779B97F4A7C15)
    z = (z ^ _srl(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ _srl(z, 27)) * c(0x94D049BB133111EB)
    return z ^ _srl(z, 31)


def gpu_w1(n_pairs, dev, seed=23, repeats=10):
    """W1 on the device: the benchmark_hashtables shape (BenchmarkHashTables.cpp:192-223): a 62-bit key, then `draw % 10` more copies,
    values = running index, shuffled.  Same rule as workloads.w1_benchmark_hashtables (the permutation is torch's)."""
    import torch
    est = int(n_pairs / ((repeats + 1) / 2.0) * 1.1) + 16
    idx = torch.arange(1, est + 1, dtype=torch.int64, device=dev)
    base = _splitmix64_t(idx + (seed << 40)) & ((1 << 62) - 1)
    freq = (_srl(_splitmix64_t(idx + ((seed + 1) << 40)), 1) % repeats) + 1
    keys = torch.repeat_interleave(base, freq)[:n_pairs]
    assert keys.numel() == n_pairs
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    p = torch.randperm(n_pairs, device=dev, generator=g)
    return keys[p].contiguous(), p.to(torch.int32).contiguous()


def gpu_w3(n_distinct, dev, mult=5, seed=3):
    """W3 on the device: n_distinct distinct 62-bit keys (2-bit packed 31-mers), each exactly `mult` times, shuffled"""
    import torch
    c = lambda v: torch.tensor(np.array([v], dtype=np.uint64).view(np.int64)[0], dtype=torch.int64, device=dev)
    m62 = (1 << 62) - 1
    ctr = torch.arange(n_distinct, dtype=torch.int64, device=dev) + seed * 1000003
    base = (ctr * c(0x9E3779B97F4A7C15)) & m62          # bijections on 62 bits: odd multiplier, xorshift
    base = base ^ (base >> 31)
    base = (base * c(0xBF58476D1CE4E5B9)) & m62
    base = base ^ (base >> 29)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    p = torch.randperm(n_distinct * mult, device=dev, generator=g)
    keys = base.repeat_interleave(mult)[p].contiguous()
    return base, keys, p.to(torch.int32).contiguous()


def _stat(ms, ops, bytes_per_op):
    ms = sorted(ms)
    med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
    rate = ops / (med * 1e-3)
    return {"ms_median": round(med, 4), "ms_min": round(ms[0], 4), "ops": int(ops), "ops_per_s": rate, "bytes_per_op": bytes_per_op,
            "frac": rate * bytes_per_op / 1e9 / HBM_PEAK_GBS}


def five_phases(cls, keys, vals, q_hit, q_miss, n_distinct, host, repeats, hash_name):
    """insert, find, find_miss, count, erase, count2 (BenchmarkHashTables.cpp:1037-1186) on a fresh table per repeat; 1 warm-up.
    host=False: device tensors, HIP events on the stream the library works on; host=True: numpy in / numpy out, wall clock around
    each call (H2D of the inputs and D2H of the results included -- the reference's semantics are host std::vectors)."""
    import torch
    names = ("insert", "find", "find_miss", "count", "erase", "count2")
    rows = {k: [] for k in names}
    nq = len(q_hit)
    for rep in range(repeats + 1):
        t = cls(128, 0.35, 0.8, hash=hash_name, seed=43)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        wall = []

        def mark(i):
            if host:
                torch.cuda.synchronize()
                wall.append(time.perf_counter())
            else:
                ev[i].record()

        mark(0)
        n_ins = t.insert(keys, vals)
        mark(1)
        fk, fv = t.find(q_hit)
        mark(2)
        mk, mv = t.find(q_miss)
        mark(3)
        c = t.count(q_hit)
        mark(4)
        n_er = t.erase(q_hit)
        mark(5)
        c2 = t.count(q_hit)
        mark(6)
        torch.cuda.synchronize()
        # size-independent properties at full size: every distinct key once, every hit found, no miss found, erased keys gone
        assert n_ins == n_distinct, (n_ins, n_distinct)
        assert len(fk) == nq and len(mk) == 0 and int(c.sum()) == nq and int(c2.sum()) == 0 and n_er == nq, (len(fk), len(mk), n_er)
        assert t.size() == n_distinct - n_er
        t.close()
        if rep == 0:
            continue
        for i, k in enumerate(names):
            rows[k].append((wall[i + 1] - wall[i]) * 1e3 if host else ev[i].elapsed_time(ev[i + 1]))
    n = len(keys)
    return {"insert": _stat(rows["insert"], n, B_INSERT_NEW), "find": _stat(rows["find"], nq, B_FIND_HIT),
            "find_miss": _stat(rows["find_miss"], len(q_miss), B_FIND_MISS), "count": _stat(rows["count"], nq, B_COUNT),
            "erase": _stat(rows["erase"], nq, B_ERASE), "count2": _stat(rows["count2"], nq, B_FIND_MISS),
            "repeats": repeats, "timing": "wall clock incl. H2D/D2H" if host else "HIP events, device-resident"}


def run_extras(args, dev, keys, vals, q, dk, dv, dq):
    """phases (RH and LP, device-resident), host_inclusive, w1, second_batch, lp_config2"""
    import torch
    import kmerhash_amd as kh
    from kmerhash_amd import workloads as W
    out = {}
    n, nq = len(keys), len(q)
    miss = W.distinct_u64(nq, seed=977)                       # a counter range the inserted keys do not use
    dmiss = torch.from_numpy(miss.view(np.int64)).to(dev)
    tables = (("robinhood", kh.hashmap_robinhood_doubling), ("linearprobe", kh.hashmap_linearprobe_doubling))
    out["phases"] = {name: five_phases(cls, dk, dv, dq, dmiss, n, False, 5, args.hash) for name, cls in tables}
    out["phases"]["workload"] = "%d distinct keys, %d all-hit queries / %d misses; erase = the %d queried keys; fresh table per repeat, 1 warm-up + 5 repeats" % (n, nq, nq, nq)
    out["phases"]["bytes_per_op"] = "SURVEY 8d: insert 50, find hit 41, find miss 9, count 26, erase 42 (count2 = count of erased keys: 9); frac = ops/s x bytes / 8 TB/s"
    out["host_inclusive"] = {name: five_phases(cls, keys, vals, q, miss, n, True, 2, args.hash) for name, cls in tables}
    out["host_inclusive"]["workload"] = "the same phases with numpy arrays in and out (pageable host memory): H2D of keys/values/queries and D2H of the results are inside the timed calls"
    torch.cuda.empty_cache()
    # ---- W1: the reference benchmark's own input shape (x5.5 mean multiplicity): 10^8 pairs, 10^7 all-hit queries
    n1 = 100_000_000
    k1, v1 = gpu_w1(n1, dev)
    q1 = k1[:nq].contiguous()
    d1 = int(torch.unique(k1).numel())
    ms = {"insert": [], "find": [], "count": [], "erase": []}
    for rep in range(4):
        t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=args.hash, seed=43)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(); ni = t.insert(k1, v1)
        ev[1].record(); fk, fv = t.find(q1)
        ev[2].record(); c = t.count(q1)
        ev[3].record(); ne = t.erase(q1)
        ev[4].record(); torch.cuda.synchronize()
        assert ni == d1 and fk.numel() == nq and int(c.sum().item()) == nq and t.size() == d1 - ne
        cap1 = t.capacity()
        t.close()
        if rep:
            for i, k in enumerate(("insert", "find", "count", "erase")):
                ms[k].append(ev[i].elapsed_time(ev[i + 1]))
    dup = n1 - d1
    out["w1"] = {"workload": "benchmark_hashtables shape (BenchmarkHashTables.cpp:192-223): %d pairs, %d distinct 62-bit keys (mean multiplicity %.2f), capacity %d, %d all-hit queries"
                             % (n1, d1, n1 / d1, cap1, nq),
                 "insert": _stat(ms["insert"], n1, (d1 * B_INSERT_NEW + dup * B_INSERT_DUP) / n1), "find": _stat(ms["find"], nq, B_FIND_HIT),
                 "count": _stat(ms["count"], nq, B_COUNT), "erase": _stat(ms["erase"], nq, B_ERASE)}
    del k1, v1, q1
    torch.cuda.empty_cache()
    # ---- second batch into the loaded table (k_build_fused SRC 2): 2*10^7 new keys + 10^6 repeats of inserted ones
    extra_k = torch.from_numpy(W.distinct_u64(20_000_000, seed=7).view(np.int64)).to(dev)
    b2 = torch.cat([extra_k, dk[:1_000_000]])
    g = torch.Generator(device=dev); g.manual_seed(5)
    b2 = b2[torch.randperm(b2.numel(), device=dev, generator=g)].contiguous()
    v2 = torch.arange(b2.numel(), dtype=torch.int32, device=dev)
    n_first = 80_000_000                                  # 8*10^7 + 2*10^7 = 10^8 <= max_load(2^27): no doubling, the in-capacity form
    ms2, fused = [], 0
    for rep in range(4):
        t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=args.hash, seed=43)
        t.insert(dk[:n_first], dv[:n_first])
        t.profile_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ni = t.insert(b2, v2); e1.record(); torch.cuda.synchronize()
        assert ni == 20_000_000 and t.size() == n_first + 20_000_000, (ni, t.size())
        fused += 1 if "k_insert_fused" in t.profile() else 0
        cap2 = t.capacity()
        t.close()
        if rep:
            ms2.append(e0.elapsed_time(e1))
    out["second_batch"] = dict(_stat(ms2, b2.numel(), (20_000_000 * B_INSERT_NEW + 1_000_000 * B_INSERT_DUP) / b2.numel()),
                               workload="2*10^7 new keys + 10^6 repeats, shuffled, into a table holding 8*10^7 keys (capacity %d stays); "
                                        "existing slots are re-laid out with the batch (32 B per old slot on top of the 8d figure)" % cap2,
                               fused_insert_launches="%d of 4" % fused)
    del extra_k, b2, v2
    torch.cuda.empty_cache()
    # ---- configs[2] / W3: linear-probe table, 2*10^7 31-mers x 5 = 10^8 pairs, insert + count of all distinct + 2*10^6 misses
    base, k3, v3 = gpu_w3(20_000_000, dev)
    miss3 = dmiss[:2_000_000] | (1 << 62)                   # bit 62 set: never a 62-bit k-mer
    q3 = torch.cat([base, miss3]).contiguous()
    ms3 = {"insert": [], "count": []}
    for rep in range(4):
        t = kh.hashmap_linearprobe_doubling(128, 0.35, 0.8, hash=args.hash, seed=43)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record(); ni = t.insert(k3, v3)
        ev[1].record(); c = t.count(q3)
        ev[2].record(); torch.cuda.synchronize()
        assert ni == 20_000_000 and t.size() == 20_000_000
        assert int(c[:20_000_000].sum().item()) == 20_000_000 and int(c[20_000_000:].sum().item()) == 0
        cap3 = t.capacity()
        t.close()
        if rep:
            ms3["insert"].append(ev[0].elapsed_time(ev[1])); ms3["count"].append(ev[1].elapsed_time(ev[2]))
    out["lp_config2"] = {"workload": "configs[2] (W3): hashmap_linearprobe_doubling, 2*10^7 distinct 62-bit 31-mers x 5 = 10^8 shuffled pairs "
                                     "(capacity %d), count of the 2*10^7 distinct keys + 2*10^6 misses; checked: size, every distinct key counted 1, "
                                     "every miss 0" % cap3,
                         "insert": _stat(ms3["insert"], 100_000_000, (20_000_000 * B_INSERT_NEW + 80_000_000 * B_INSERT_DUP) / 100_000_000),
                         "count": _stat(ms3["count"], q3.numel(), (20_000_000 * B_COUNT + 2_000_000 * B_FIND_MISS) / q3.numel())}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def child_env(rank, world, port, base=None):
    """environment of rank `rank` of `world` (the torch.distributed.run contract: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)"""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "KH_BENCH_LAUNCHED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def child_argv(argv):
    """rank processes get the launcher's own arguments; the CPU legs stay with the launcher"""
    out = [sys.executable, os.path.abspath(__file__)] + list(argv)
    if "--no-cpu-baseline" not in out:
        out.append("--no-cpu-baseline")
    return out


def visible_gpus():
    import torch                       # device_count() does not initialise the GPU
    return torch.cuda.device_count()


def launch(args, argv):
    n = args.gpus
    have = visible_gpus()
    if REHEARSAL:              # every rank on device 0 over gloo: a dry run of the N > 1 control flow on a one-GPU box, never a measurement
        have = max(have, n) if have >= 1 else have
    if have < n:
        print("[bench] --gpus %d requested but only %d GPU(s) visible: refusing to run fewer ranks than asked for" % (n, have),
              file=sys.stderr, flush=True)
        return 2
    cpu = None
    if n == 1 and not args.no_cpu_baseline:
        keys, vals, q = gen_inputs(0, args.keys, args.queries, args.workload)
        cpu = cpu_baseline(keys, vals, q)
        del keys, vals, q
    port = _free_port()
    cmd = child_argv(argv)
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=child_env(r, n, port), stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      universal_newlines=True))
    out0 = None
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        for pr in procs:
            pr.wait()
            if pr.returncode != 0 and rc == 0:
                rc = pr.returncode
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    if rc != 0:
        print("[bench] a rank process failed (exit code %d)" % rc, file=sys.stderr, flush=True)
        return rc
    lines = [l for l in (out0 or "").splitlines() if l.strip().startswith("{")]
    if not lines:
        print("[bench] rank 0 printed no result line", file=sys.stderr, flush=True)
        return 3
    res = json.loads(lines[-1])
    if cpu is not None:
        res["cpu_baseline"] = cpu
    print(json.dumps(res), flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    # everything libraries write to stdout (RCCL prints a version banner there at communicator creation) goes to stderr,
    # so that stdout carries exactly ONE line: the JSON result
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("[bench] WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr, flush=True)
        return 2
    force = os.environ.get("KH_DIST_FORCE_COLLECTIVES", "0") == "1"       # rehearsal of the N>1 code path on one GPU
    distributed = world > 1 or force
    keys, vals, q = gen_inputs(rank, args.keys, args.queries, args.workload)

    import torch
    import kmerhash_amd as kh
    from kmerhash_amd import dist as khd
    if REHEARSAL:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        print("[bench] rank %d: device %d not visible" % (rank, local_rank), file=sys.stderr, flush=True)
        return 2
    if args.chunks <= 0:
        args.chunks = 4 if world > 1 else 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (bounded: a collective that does not complete within 3 minutes ends the rank with an error instead of a silent hang)
        import datetime
        if REHEARSAL:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=180))
        if dist.get_world_size() != args.gpus:
            print("[bench] process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus), file=sys.stderr, flush=True)
            return 2
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)                       # every rank contributes a 1 over RCCL
        rccl_ranks = int(one.item())
        if rccl_ranks != args.gpus:
            print("[bench] all-reduce over RCCL saw %d ranks, expected %d" % (rccl_ranks, args.gpus), file=sys.stderr, flush=True)
            return 2

    n_distinct = len(np.unique(keys)) if args.workload == "w1" else args.keys
    dk = torch.from_numpy(keys.view(np.int64)).to(dev)
    dv = torch.from_numpy(vals.view(np.int32)).to(dev)
    dq = torch.from_numpy(q.view(np.int64)).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    prof, phases = {}, {}
    ins_ms, find_ms = [], []

    def one_step(timed):
        if distributed:
            be = khd.GpuBackend(local_rank, "rh", 128, 0.35, 0.8, args.hash, 43)
            t = khd.ShardedTable(be, timing=timed)
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
            t.synchronize()                        # (find only queues its work; this also raises if any rank's local part failed)
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
            if t is not None:
                for k, ms in t.timings().items():
                    phases[k] = phases.get(k, 0.0) + ms
        state = (n_ins, n_hit, table.size(), table.capacity())
        table.close()
        return state

    for _ in range(args.warmup):       # a failure here (or anywhere) ends the rank with a non-zero exit code: no fall-back configuration
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
        im = torch.tensor([float(np.mean(ins_ms)), float(np.mean(find_ms))], dtype=torch.float64, device=dev)
        dist.all_reduce(im, op=dist.ReduceOp.MAX)          # a phase ends when its slowest rank ends
        ins_mean, find_mean = (float(x) for x in im.cpu())
    else:
        g_ins, g_hit, g_size = state[0], state[1], state[2]
        ins_mean, find_mean = float(np.mean(ins_ms)), float(np.mean(find_ms))

    # size-independent parity properties at full size (the oracle cannot run 1e8 keys in seconds):
    # every distinct key inserted exactly once, every query found
    if args.workload == "w2":
        assert g_ins == args.keys * world, (g_ins, args.keys * world)
        assert g_size == args.keys * world
    elif not distributed:
        assert g_ins == n_distinct and g_size == n_distinct, (g_ins, n_distinct)
    assert g_hit == args.queries * world, (g_hit, args.queries * world)

    extras = None
    if not distributed and not args.no_extras and args.workload == "w2":
        try:            # informational legs: a failure here must not take the headline measurement (already complete) with it
            extras = run_extras(args, dev, keys, vals, q, dk, dv, dq)
        except Exception as e:
            import traceback
            traceback.print_exc()
            extras = {"extras_error": repr(e)[:500]}
    if rank == 0:
        ops_per_step = (args.keys + args.queries) * world
        ms_per_step = elapsed / args.steps * 1e3
        ins_rate = args.keys * world / (ins_mean * 1e-3)
        find_rate = args.queries * world / (find_mean * 1e-3)
        # achieved = ops/s x algorithmic bytes per op (per GPU), over the driver-visible phase time
        ins_gbs = ins_rate / world * B_INSERT_NEW / 1e9
        find_gbs = find_rate / world * B_FIND_HIT / 1e9
        low = "insert" if ins_gbs <= find_gbs else "find"
        # the kernel with the largest time per step, HIP-event timed inside the library on the table's stream, and its HBM
        # bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/<tag>_pmc_hbm_traffic.json,
        # FETCH_SIZE/WRITE_SIZE collected and corrected as MI355X_MICROARCH.md prescribes; PMC counters cannot be read from
        # inside the timed process, so this is the recorded figure).  hbm_util = measured traffic / time / peak: how busy the
        # memory system is, NOT the 8d figure.
        dom = max((k for k in prof if k.startswith("k_")), key=lambda k: prof[k][1])
        launches, total_ms = prof[dom]
        per_launch_ms = total_ms / launches
        traffic, traffic_src, path_traffic = None, None, None
        try:
            tag = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()      # written by scripts/summarize_profiles.py
            if not distributed:
                pm = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.json" % tag)))
                traffic = pm.get(dom, {}).get("hbm_bytes_per_launch")
                traffic_src = "%s_pmc_hbm_traffic.json" % tag
                path_traffic = pm.get("_insert_path", {}).get("hbm_bytes_per_batch")
        except Exception:
            traffic = None
        # the find path against the MEASURED random-access rate of this GPU (BASELINE.json's metric: "% HBM random-access roofline"):
        # memory read requests per query (PMC, recorded) x finds/s over the rate at which the device serves random 64-byte sector
        # reads of a table-sized buffer (scripts/random_access_roofline.hip, recorded)
        ra = None
        try:
            rj = json.load(open(os.path.join(ROOT, "profiles", "%s_random_access.json" % tag)))
            touches = find_rate / world * rj["k_find_rdreq_per_query"]
            ra = {"sector_touches_per_query_pmc": rj["k_find_rdreq_per_query"], "achieved_touches_per_s": touches,
                  "measured_peak_touches_per_s": rj["peak_used"], "frac": touches / rj["peak_used"],
                  "source": "profiles/%s_random_access.json" % tag}
        except Exception:
            ra = None
        load = state[2] / state[3]
        out = {
            "metric": "kmer_inserts_plus_finds_per_sec",
            "value": ops_per_step / (elapsed / args.steps),
            "unit": "kmer_ops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("(informational, benchmark_hashtables shape x5.5 multiplicity) " if args.workload == "w1" else "") +
                                   "configs[1]: Robin Hood table, %d distinct random 64-bit k-mers per GPU (max load 0.8 -> capacity %d, "
                                   "load %.3f), murmur3avx64 seed 43, then %d all-hit finds per GPU%s"
                                   % (args.keys, state[3], load, args.queries,
                                      "; keys sharded by murmur3(seed 9876543) over RCCL, exchange pipelined in %d pieces" % args.chunks if distributed else ""),
                       "keys_per_gpu": args.keys, "queries_per_gpu": args.queries, "table": "hashmap_robinhood_doubling",
                       "hash": args.hash, "max_load_factor": 0.8, "min_load_factor": 0.35, "final_load": load,
                       "exchange_pieces": args.chunks if distributed else None},
            "rccl_ranks": rccl_ranks if not REHEARSAL else 0, "rehearsal": REHEARSAL,
            "inserts_per_s": ins_rate, "finds_per_s": find_rate,
            "insert_ms": ins_mean, "find_ms": find_mean,
            "roofline": {"bound": "hbm", "op": "insert", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "achieved": ins_gbs, "frac": ins_gbs / HBM_PEAK_GBS,
                         "formula": "ops/s per GPU x algorithmic bytes/op (SURVEY 8d: insert 50 B, find hit 41 B) / 8 TB/s; achieved / frac = the insert "
                                    "path (inserts_per_s x 50 B), the find path is reported next to it and `lower` names the smaller of the two",
                         "lower": {"op": low, "frac": min(ins_gbs, find_gbs) / HBM_PEAK_GBS},
                         "insert": {"achieved": ins_gbs, "frac": ins_gbs / HBM_PEAK_GBS, "bytes_per_op": B_INSERT_NEW,
                                    "sector_frac": ins_rate / world * B_INSERT_SECTOR / 1e9 / HBM_PEAK_GBS, "sector_bytes_per_op": B_INSERT_SECTOR,
                                    "hbm_bytes_per_batch_pmc": path_traffic},
                         "find": {"achieved": find_gbs, "frac": find_gbs / HBM_PEAK_GBS, "bytes_per_op": B_FIND_HIT,
                                  "sector_frac": find_rate / world * B_FIND_SECTOR / 1e9 / HBM_PEAK_GBS, "sector_bytes_per_op": B_FIND_SECTOR,
                                  "random_access": ra},
                         "traffic": traffic, "traffic_source": traffic_src,
                         "dominant_kernel": {"name": dom, "avg_launch_ms": per_launch_ms, "launches_per_step": launches / args.steps,
                                             "hbm_bytes_per_launch_pmc": traffic,
                                             "hbm_util": (traffic / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}},
            "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
        }
        if distributed:
            out["phases_ms_per_step_rank0"] = {k: round(v / args.steps, 4) for k, v in sorted(phases.items())}
        if extras:
            out.update(extras)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.destroy_process_group()
    return 0


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--keys", type=int, default=0, help="keys per GPU (0 = 107374184 at N=1: load exactly 0.800; 10^8 per rank at N>1)")
    ap.add_argument("--queries", type=int, default=QUERIES_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N == 1: skip the informational legs (five phases for both tables, host-inclusive, W1, second batch, configs[2])")
    ap.add_argument("--workload", default="w2", choices=["w2", "w1"],
                    help="w2 = configs[1] (default, the metric's workload); w1 = benchmark_hashtables shape (x5.5 multiplicity), informational")
    ap.add_argument("--hash", default="murmur3avx64", choices=["murmur3avx64", "murmur", "farm", "identity"],
                    help="storage hash (default = the metric's: murmur3avx64; the others are informational)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N>1: pieces of the pipelined exchange/insert (permute, xGMI transfer and radix partition of successive pieces "
                         "overlap); 1 = exchange, then one bulk insert; 0 = auto (4 when N>1: the exchange is link-bound at every N)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.keys <= 0:
        args.keys = KEYS_LOAD_080 if args.gpus == 1 else KEYS_PER_GPU_DIST
    return args


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if "WORLD_SIZE" in os.environ:
        return run_rank(args)
    return launch(args, argv)


if __name__ == "__main__":
    sys.exit(main())
