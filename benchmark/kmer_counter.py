#!/usr/bin/env python3
"""BenchmarkKmerCounter shape on one GPU (reference benchmark/BenchmarkKmerCounter.cpp:1476-1787: read FASTQ in batches ->
canonical 31-mers -> counting insert -> write (k-mer,count) tuples), on a synthetic FASTQ (random genome, 150-bp reads).
Informational driver for the SURVEY 8f-2 row; the contract benchmark is ../bench.py."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=20_000_000)
    ap.add_argument("--batches", type=int, default=4, help="file batches (the reference reads the input in memory-sized pieces)")
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--out", default="")
    ap.add_argument("--raw-fastq", action="store_true", help="feed raw FASTQ text: the record structure is resolved on the GPU (kh_kmers_from_fastq)")
    a = ap.parse_args()
    import torch
    from kmerhash_amd import kmers as KM
    t0 = time.perf_counter()
    seq = (KM.synthetic_fastq_fixed if a.raw_fastq else KM.synthetic_read_sequences)(a.reads, a.read_len, a.genome, seed=7)
    t_gen = time.perf_counter() - t0
    dseq = torch.from_numpy(seq).cuda()
    kc = KM.KmerCounter(a.k, canonical=True, hash="farm")
    # batches cut at read boundaries
    nl = np.flatnonzero(seq == 10)
    if a.raw_fastq:
        nl = nl[3::4]                            # record ends
    cuts = [0] + [int(nl[len(nl) * i // a.batches - 1]) + 1 for i in range(1, a.batches)] + [len(seq)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = 0
    for i in range(a.batches):
        total += (kc.add_fastq if a.raw_fastq else kc.add_sequences)(dseq[cuts[i]:cuts[i + 1]])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s bytes %d  k-mers %d  distinct %d  capacity %d" % ("FASTQ" if a.raw_fastq else "sequence", len(seq), total, kc.table.size(), kc.table.capacity()))
    print("generate+parse (host) %.2f s ; k-mer generation + counting (device, %d batches) %.4f s = %.3f G k-mers/s"
          % (t_gen, a.batches, dt, total / dt / 1e9))
    if a.out:
        print("wrote %d tuples to %s" % (kc.write(a.out), a.out))
    kc.close()


if __name__ == "__main__":
    main()
