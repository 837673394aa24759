#!/usr/bin/env python3
"""BASELINE configs[4] shape: the distributed k-mer counter with its query cycle, on synthetic FASTQ.

Reference: benchmark/BenchmarkKmerCounter.cpp:1476-1787 (read the input in file batches -> canonical 31-mers -> counting
insert into dsc::counting_batched_robinhood_map, farmhash storage hash, the table doubling under load) followed by the
query phase of benchmark/BenchmarkKmerIndex.cpp:787-843 (count, find, erase over a sample of the input).

  python benchmark/kmer_counter.py [--gpus N] [--reads R] [--batches B] [--cycle] ...

Every rank owns R reads of ONE synthetic genome (150-bp error-free reads on both strands: the count of every k-mer is then
predictable from the read positions alone, which is what --verify checks at any scale), parses its FASTQ text on its GPU
(kh_kmers_from_fastq: record structure, 2-bit packing and canonicalisation), and feeds the k-mers to the sharded counting
table (kmerhash_amd.dist.ShardedTable.insert_counts: hash partition, RCCL exchange, local std::plus insert; one rank: no
exchange).  Launch model as bench.py: without WORLD_SIZE this process starts the N ranks itself and never touches a GPU.
Prints ONE JSON line (rank 0).  Informational driver for SURVEY 8f-2 / configs[4]; the contract benchmark is ../bench.py."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--reads", type=int, default=2_000_000, help="reads per rank")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=20_000_000)
    ap.add_argument("--batches", type=int, default=4, help="file batches per rank; 0 = sized from the free HBM the way the reference sizes its file batches "
                                                          "from free memory (BenchmarkKmerCounter.cpp:1508-1590: k-mer estimate x working bytes + 2 x text + table growth <= usable)")
    ap.add_argument("--host-gen", action="store_true", help="generate the reads with numpy on the host (default: on the GPU with torch; neither is timed)")
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--hash", default="farm", choices=["farm", "murmur3avx64", "murmur"])
    ap.add_argument("--chunks", type=int, default=0, help="pieces of the pipelined exchange per batch (0 = 4 when N > 1)")
    ap.add_argument("--cycle", action="store_true", help="run the count / find / erase / count query cycle after the inserts")
    ap.add_argument("--sample-ratio", type=int, default=100, help="queries = every s-th k-mer of the rank's input (BenchmarkKmerIndex -q sampling)")
    ap.add_argument("--hll-reserve", action="store_true", help="pre-size the table from a HyperLogLog estimate per batch instead of doubling under load")
    ap.add_argument("--profile", action="store_true", help="per-kernel HIP-event times of the local table (kh_profile_*) in the JSON line")
    ap.add_argument("--verify", action="store_true", help="check size, total count and a sample of 10^5 k-mer counts against the prediction from the read positions")
    ap.add_argument("--out", default="", help="write this rank's (k-mer, count) tuples (BenchmarkKmerCounter.cpp:1022-1211): 8 + 2 bytes each like the "
                                               "reference, whose CountType is uint16_t (:184) -- the table's 32-bit counts are truncated to 16 bits in the FILE "
                                               "(= the value the reference's wrapping counter would hold); find / count / --verify use the 32-bit counts")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args, argv):
    import torch
    if torch.cuda.device_count() < args.gpus:
        print("[kmer_counter] --gpus %d requested, %d visible" % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def run_rank(args):
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("[kmer_counter] WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        return 2
    import torch
    from kmerhash_amd import kmers as KM
    from kmerhash_amd import dist as khd
    from kmerhash_amd.hll import hyperloglog64
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    chunks = args.chunks if args.chunks > 0 else (4 if world > 1 else 1)

    # ---- input: this rank's reads of the common genome (generation is not timed)
    t0 = time.perf_counter()
    if args.host_gen:
        seq, genome, starts, rev = KM.synthetic_reads(args.reads, args.read_len, args.genome, genome_seed=7, read_seed=100 + rank)   # one genome, own reads
        fq = KM.fastq_from_sequence_lines(seq, args.reads, args.read_len)
        dfq = torch.from_numpy(fq).to(dev)
        del seq, fq
        positions_of = lambda r: torch.from_numpy((starts if r == rank else KM.read_positions(args.reads, args.read_len, args.genome, 100 + r)[0]).astype(np.int64)).to(dev)
    else:
        dfq, dgenome, dstarts = KM.synthetic_fastq_device(args.reads, args.read_len, args.genome, 7, 100 + rank, local)
        genome = None

        def positions_of(r):
            if r == rank:
                return dstarts
            g = torch.Generator(device=dev); g.manual_seed(100 + r)
            return torch.randint(0, args.genome - args.read_len, (args.reads,), dtype=torch.int64, device=dev, generator=g)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    n_text = int(dfq.numel())
    rec = n_text // args.reads
    if args.batches <= 0:
        # the reference adds input files to an iteration while kmer_est x 5 x sizeof(tuple) + 2 x file_size + the table's growth stays
        # under the usable memory (BenchmarkKmerCounter.cpp:1508-1590).  Here, per byte of FASTQ text of a batch: 8 B of k-mer output
        # room + 1 B of masked text (kh_kmers_from_fastq), and per k-mer (0.38 per text byte at 150-bp reads) ~70 B of counting-insert
        # workspace (partition records, lists, exchange buffers); the table (16 B x capacity, twice while it is re-laid out) on top
        free_b, _ = torch.cuda.mem_get_info(dev)
        kmers_per_byte = max(args.read_len - args.k + 1, 1) / float(rec)
        per_byte = 9.0 + kmers_per_byte * 70.0
        table_b = 2 * 16 * (1 << max(7, int(np.ceil(np.log2(max(args.genome, 128) / 0.8)))))
        usable = max(0.6 * free_b - table_b, 0.05 * free_b)
        batch_bytes = min(usable / per_byte, 5.0e8 / kmers_per_byte)           # (a batch stays under 2^29 k-mers: one internal pass of the library,
                                                                               #  whose workspace -- allocated once -- then fits every batch)
        nb = max(1, int(np.ceil(n_text / batch_bytes)))
        args.batches = nb
        batch_note = {"free_hbm": int(free_b), "usable": int(usable), "bytes_per_text_byte": round(per_byte, 1), "batch_text_bytes": int(n_text / nb)}
    else:
        batch_note = None
    cuts = [rec * (args.reads * i // args.batches) for i in range(args.batches + 1)]

    be = khd.GpuBackend(local, "rh", 128, 0.35, 0.8, args.hash, 43)
    st = khd.ShardedTable(be, timing=True)
    hll = hyperloglog64(12, 0, args.hash, 43, local) if args.hll_reserve else None
    kc = KM.ShardedKmerCounter(st, args.k, True, chunks=chunks, reserve_from_estimate=args.hll_reserve, hll=hll)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    caps = []
    if args.profile:
        be.table.profile_enable(True)
    sync()
    t0 = time.perf_counter()
    t_batch = []
    for i in range(args.batches):
        tb = time.perf_counter()
        kc.add_fastq(dfq[cuts[i]:cuts[i + 1]])
        caps.append(be.table.capacity())
        if args.profile:
            torch.cuda.synchronize()
            t_batch.append(round((time.perf_counter() - tb) * 1e3, 2))
    sync()
    t_ins = time.perf_counter() - t0
    total_local = kc.total_kmers
    size_after = kc.size()
    res = {"kmers_local": total_local, "capacity_per_batch_rank0": caps, "distinct_global": size_after,
           "insert_s": t_ins, "phases_ms_rank0": {k: round(v, 3) for k, v in st.timings().items()}, "batch_sizing": batch_note}
    if args.profile:
        res["batch_ms"] = t_batch
        res["insert_kernels_ms"] = {k: round(v[1], 2) for k, v in sorted(be.table.profile().items(), key=lambda kv: -kv[1][1])}
        be.table.profile_reset()

    ok = True
    if args.verify:
        # counts predicted from the read positions of ALL ranks (cov[p] = reads covering the k-mer at genome position p)
        d = torch.zeros(args.genome + 1, dtype=torch.int64, device=dev)
        one = torch.ones(args.reads, dtype=torch.int64, device=dev)
        for r in range(world):
            s_r = positions_of(r)
            d.index_add_(0, s_r, one)
            d.index_add_(0, s_r + (args.read_len - args.k + 1), -one)
        dcov = torch.cumsum(d, 0)[:args.genome]
        exp_total = int(dcov.sum().item())
        exp_distinct = int((dcov > 0).sum().item())                # exact when the genome's k-mers are pairwise distinct (k = 31: they are)
        rng = np.random.default_rng(99 + rank)
        pos = rng.integers(0, args.genome - args.k, 100_000)
        cov_at = dcov[torch.from_numpy(pos).to(dev)].cpu().numpy()
        del d, dcov, one
        if genome is None:
            genome = dgenome.cpu().numpy()
        qk = KM.canonical_kmers_at(genome, pos, args.k)
        pk, vals, found = st.find(torch.from_numpy(qk.view(np.int64)).to(dev))
        got = dict(zip(pk.cpu().numpy().view(np.uint64).tolist(), ((vals.cpu().numpy().view(np.uint32).astype(np.int64)) * found.cpu().numpy()).tolist()))
        exp = {}
        for kk, c in zip(qk.tolist(), cov_at.tolist()):
            exp[kk] = c                                            # (a k-mer sampled twice has the same position-independent count)
        bad = sum(1 for kk, c in exp.items() if got.get(kk, -1) != (c & 0xFFFFFFFF))
        tot = torch.tensor([total_local], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(tot)
        ok = bad == 0 and int(tot.item()) == exp_total and size_after == exp_distinct
        res["verify"] = {"ok": bool(ok), "sample_mismatches": bad, "total_kmers": int(tot.item()), "expected_total": exp_total,
                         "expected_distinct": exp_distinct}

    if args.cycle:
        # queries: every s-th k-mer of this rank's first batch (BenchmarkKmerIndex samples the input file the same way)
        km = KM.kmers_from_fastq(dfq[cuts[0]:cuts[1]], args.k, True, local)
        qs = km[:: args.sample_ratio].contiguous()
        sync()
        t0 = time.perf_counter()
        cyc = kc.cycle(qs)
        sync()
        t_cyc = time.perf_counter() - t0
        nq = int(qs.numel())
        ok = ok and cyc["count_hits"] == nq and cyc["find_hits"] == nq and cyc["count_hits_after"] == 0
        # (ops_per_s: the four operations over their own synchronised times; `seconds` is the wall clock of the whole call, which also
        #  holds torch's result reductions -- their first use loads torch kernels, ~0.1 s once per process)
        res["cycle"] = dict(cyc, queries_local=nq, seconds=t_cyc, ops_per_s=4 * nq * world / (sum(cyc["phase_ms"].values()) * 1e-3), size_after=kc.size(),
                            ok=bool(cyc["count_hits"] == nq and cyc["count_hits_after"] == 0))
    if args.out:
        k_, v_ = be.table.to_vector()
        recs = np.zeros(len(k_), dtype=np.dtype([("kmer", "<u8"), ("count", "<u2")]))
        recs["kmer"] = k_; recs["count"] = v_.astype(np.uint16)
        recs.tofile(args.out + (".%d" % rank if world > 1 else ""))
    if rank == 0:
        res.update({"n_gpus": world, "reads_per_rank": args.reads, "batches": args.batches, "k": args.k, "hash": args.hash,
                    "exchange_pieces": chunks, "fastq_bytes_per_rank": n_text, "bases_per_rank": args.reads * args.read_len, "genome": args.genome,
                    "generation_s": round(t_gen, 2),
                    "kmers_per_s": total_local * world / t_ins, "ok": bool(ok)})
        print(json.dumps(res), flush=True)
    be.table.close()
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if "WORLD_SIZE" in os.environ:
        return run_rank(args)
    if args.gpus == 1:
        os.environ.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
        return run_rank(args)
    return launch(args, argv)


if __name__ == "__main__":
    sys.exit(main())
