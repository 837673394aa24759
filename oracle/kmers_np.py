"""TEST INFRASTRUCTURE (oracle/): plain numpy statement of this library's k-mer definition (include/kmerhash_amd.h,
kh_kmers_from_sequence / kh_kmers_from_fastq).  The reference takes its parser and k-mer type from kmerind, which is not
part of the reference tree (BenchmarkKmerCounter.cpp:1655-1706): PARITY UNPINNED with respect to the reference; the tests
pin the GPU front end to this statement.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it."""
import numpy as np


def np_kmers(seq, k, canonical):
    """windows of k valid bases (ACGT, either case), first base most significant, A0 C1 G2 T3; any other byte ends a run;
    canonical: min(k-mer, reverse complement)"""
    code = np.full(256, 4, dtype=np.uint8)
    for ch, c in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        code[ch] = c
    c = code[np.asarray(seq, dtype=np.uint8)]
    n = len(c)
    if n < k:
        return np.zeros(0, dtype=np.uint64)
    valid = c < 4
    bad = np.concatenate([[0], np.cumsum(~valid)])
    ok = (bad[k:] - bad[: n - k + 1]) == 0
    fw = np.zeros(n - k + 1, dtype=np.uint64)
    rc = np.zeros(n - k + 1, dtype=np.uint64)
    cc = (c & 3).astype(np.uint64)
    for j in range(k):
        fw = (fw << np.uint64(2)) | cc[j: n - k + 1 + j]
        rc |= (np.uint64(3) - cc[j: n - k + 1 + j]) << np.uint64(2 * j)
    out = np.minimum(fw, rc) if canonical else fw
    return out[ok]


def np_kmers_fastq(text, k, canonical):
    """the same over raw FASTQ text: only the sequence lines (line 1 mod 4) yield k-mers"""
    raw = bytes(text) if isinstance(text, (bytes, bytearray)) else np.asarray(text, dtype=np.uint8).tobytes()
    seqs = raw.split(b"\n")[1::4]
    if not seqs:
        return np.zeros(0, dtype=np.uint64)
    return np_kmers(np.frombuffer(b"\n".join(seqs) + b"\n", dtype=np.uint8), k, canonical)
