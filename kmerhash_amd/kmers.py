"""k-mer counting front end: the BenchmarkKmerCounter shape (BenchmarkKmerCounter.cpp:1476-1787: read
FASTQ/FASTA -> k-mers -> counting insert -> write (k-mer, count) tuples) on the GPU tables.

The reference takes its parser and k-mer type from kmerind (absent), so the k-mer definition here is this library's
(see kh_kmers_from_sequence in include/kmerhash_amd.h): 2-bit packed, first base most significant, A=0 C=1 G=2 T=3,
windows containing any other byte are skipped, canonical = min(k-mer, reverse complement)."""
import ctypes as C

import numpy as np

from . import _capi as K
from .table import _Buf, hashmap_robinhood_doubling, torch


def sequences_from_fastq(buf):
    """FASTQ text (bytes / uint8 array) -> uint8 array holding only the sequence lines, each followed by '\\n'
    (records are 4 lines: @id, sequence, +, quality)."""
    raw = bytes(buf) if isinstance(buf, (bytes, bytearray)) else np.asarray(buf, dtype=np.uint8).tobytes()
    seqs = raw.split(b"\n")[1::4]
    if not seqs:
        return np.zeros(0, dtype=np.uint8)
    return np.frombuffer(b"\n".join(seqs) + b"\n", dtype=np.uint8).copy()


def sequences_from_fasta(buf):
    """FASTA text -> sequence bytes; header lines are replaced by a single '\\n' so that k-mers never span records
    (line breaks inside a record are removed)."""
    a = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else np.asarray(buf, dtype=np.uint8)
    lines = bytes(a).split(b"\n")
    parts = []
    for ln in lines:
        if ln.startswith(b">"):
            parts.append(b"\n")
        else:
            parts.append(ln.strip())
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def kmers_from_fastq(text, k=31, canonical=True, device=0):
    """raw FASTQ text (whole 4-line records; bytes / uint8 array on the host, or a uint8 CUDA tensor) -> packed k-mers of the
    sequence lines: the record structure is resolved on the GPU (kh_kmers_from_fastq), no host-side parsing"""
    return kmers_from_sequence(text, k, canonical, device, _fastq=True)


def kmers_from_sequence(seq, k=31, canonical=True, device=0, _fastq=False):
    """-> packed k-mers (numpy uint64 for host input, torch int64 CUDA tensor for device input), sequence order"""
    L = K.lib()
    if isinstance(seq, (bytes, bytearray)):
        seq = np.frombuffer(seq, dtype=np.uint8)
    b = _Buf(seq, np.uint8, 1)
    n_out = C.c_uint64()
    if b.where == K.KH_MEM_DEVICE:
        out = torch.empty(max(b.n, 1), dtype=torch.int64, device=b.device)
        optr = out.data_ptr()
        stream = torch.cuda.current_stream(device).cuda_stream
    else:
        out = np.zeros(max(b.n, 1), dtype=np.uint64)
        optr = out.ctypes.data
        stream = None
    fn = L.kh_kmers_from_fastq if _fastq else L.kh_kmers_from_sequence
    st = fn(b.ptr, b.n, k, 1 if canonical else 0, b.where, optr, C.byref(n_out), device, stream)
    if st != K.KH_OK:
        raise K.KhError(st, "kh_kmers_from_fastq" if _fastq else "kh_kmers_from_sequence")
    return out[: n_out.value]


class KmerCounter:
    """counting index: k-mer -> number of occurrences (Reducer = std::plus, value 1 per occurrence)"""

    def __init__(self, k=31, canonical=True, hash="farm", min_load_factor=0.35, max_load_factor=0.8, device=0):
        self.k, self.canonical, self.device = k, canonical, device
        self.table = hashmap_robinhood_doubling(128, min_load_factor, max_load_factor, hash=hash, seed=43, device=device)

    def add_sequences(self, seq):
        km = kmers_from_sequence(seq, self.k, self.canonical, self.device)
        if len(km):
            self.table.insert_reduce_plus(km)
        return len(km)

    def add_fastq(self, buf):
        """raw FASTQ text (whole records), host or device: record structure, k-mer generation and counting all run on the GPU"""
        km = kmers_from_fastq(buf, self.k, self.canonical, self.device)
        if len(km):
            self.table.insert_reduce_plus(km)
        return len(km)

    def counts(self):
        return self.table.to_vector()

    def write(self, filename, count_dtype=np.uint16):
        """raw (k-mer, count) tuples, sizeof(KmerType) + sizeof(CountType) bytes each, no padding
        (BenchmarkKmerCounter.cpp:1022-1059 copyToByteArray; CountType = uint16_t there, wrapping like std::plus)"""
        k, v = self.counts()
        rec = np.zeros(len(k), dtype=np.dtype([("kmer", "<u8"), ("count", np.dtype(count_dtype).newbyteorder("<"))]))
        rec["kmer"] = k
        rec["count"] = v.astype(count_dtype)
        rec.tofile(filename)
        return len(k)

    def close(self):
        self.table.close()


def synthetic_fastq(n_reads, read_len=150, genome_len=1_000_000, seed=7, n_rate=0.001):
    """random genome, uniformly sampled reads on both strands, a sprinkle of N's (SURVEY §8d W5 shape, scaled)"""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len, dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    starts = rng.integers(0, genome_len - read_len, n_reads)
    rev = rng.integers(0, 2, n_reads).astype(bool)
    out = []
    for i in range(n_reads):
        r = genome[starts[i]: starts[i] + read_len]
        if rev[i]:
            r = comp[r[::-1]]
        s = lut[r].copy()
        m = rng.random(read_len) < n_rate
        s[m] = ord("N")
        out.append(b"@r%d\n" % i + s.tobytes() + b"\n+\n" + b"I" * read_len + b"\n")
    return b"".join(out)


def synthetic_read_sequences(n_reads, read_len=150, genome_len=1_000_000, seed=7, n_rate=0.001):
    """the sequence lines of synthetic_fastq's reads without the FASTQ text around them (vectorised: for large inputs):
    uint8 array of n_reads lines of read_len bases, each followed by a newline"""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((n_reads, read_len + 1), dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int32)[None, :]
    for c0 in range(0, n_reads, 100_000):                      # chunked: the index matrix stays cache/L2 sized
        c1 = min(n_reads, c0 + 100_000)
        starts = rng.integers(0, genome_len - read_len, c1 - c0, dtype=np.int32)
        rev = rng.integers(0, 2, c1 - c0, dtype=np.uint8).astype(bool)
        r = genome[starts[:, None] + ar]
        r[rev] = (3 - r[rev])[:, ::-1]
        blk = out[c0:c1, :read_len]
        blk[...] = lut[r]
        n_n = rng.binomial((c1 - c0) * read_len, n_rate)
        if n_n:
            blk[rng.integers(0, c1 - c0, n_n), rng.integers(0, read_len, n_n)] = ord("N")
    out[:, read_len] = 10
    return out.reshape(-1)


def synthetic_fastq_fixed(n_reads, read_len=150, genome_len=1_000_000, seed=7, n_rate=0.001):
    """raw FASTQ text of synthetic_read_sequences' reads (vectorised, fixed-width records: '@' + 11-digit id, sequence, '+',
    quality 'I' * read_len), as a uint8 array -- the input shape of BenchmarkKmerCounter, for the GPU FASTQ path"""
    seq = synthetic_read_sequences(n_reads, read_len, genome_len, seed, n_rate).reshape(n_reads, read_len + 1)
    w = 1 + 11 + 1 + (read_len + 1) + 2 + (read_len + 1)
    out = np.empty((n_reads, w), dtype=np.uint8)
    out[:, 0] = ord("@")
    ids = np.arange(n_reads, dtype=np.int64)
    for d in range(11):
        out[:, 11 - d] = ord("0") + (ids % 10)
        ids //= 10
    out[:, 12] = 10
    out[:, 13: 13 + read_len + 1] = seq
    o = 13 + read_len + 1
    out[:, o] = ord("+"); out[:, o + 1] = 10
    out[:, o + 2: o + 2 + read_len] = ord("I")
    out[:, o + 2 + read_len] = 10
    return out.reshape(-1)

