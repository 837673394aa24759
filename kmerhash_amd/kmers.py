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



class ShardedKmerCounter:
    """BASELINE configs[4]: the distributed k-mer counter.  BenchmarkKmerCounter.cpp:1476-1787 reads the input in file batches,
    turns every batch into canonical k-mers, and inserts them into dsc::counting_batched_robinhood_map (Reducer = std::plus,
    distributed_batched_robinhood_map.hpp:2542-2950): every rank parses ITS share of the reads, the k-mers are sharded by
    murmur3(k-mer, seed 9876543) over the ranks (kmerhash_amd.dist.ShardedTable: RCCL exchange, pipelined) and counted in the
    owner's local table, which starts at capacity 128 and doubles under load (or is pre-sized from a HyperLogLog estimate,
    robinhood_offset_hashmap_ptr.hpp:2512-2535, with reserve_from_estimate=True).  cycle() is the query phase of
    BenchmarkKmerIndex.cpp:787-843: count, find, erase over a sample of the input, then count again.

    `sharded` is a kmerhash_amd.dist.ShardedTable (any backend); `kmer_fn(text) -> packed canonical k-mers` is the k-mer
    generator (default: kh_kmers_from_fastq on this rank's GPU)."""

    def __init__(self, sharded, k=31, canonical=True, kmer_fn=None, chunks=1, reserve_from_estimate=False, hll=None):
        self.st, self.k, self.canonical, self.chunks = sharded, k, canonical, chunks
        dev = getattr(sharded.b, "device", 0)
        self.kmer_fn = kmer_fn if kmer_fn is not None else (lambda text: kmers_from_fastq(text, k, canonical, dev))
        self.reserve_from_estimate = reserve_from_estimate
        self.hll = hll
        self.total_kmers = 0

    def add_fastq(self, text):
        """one file batch of this rank: raw FASTQ text (whole records) -> k-mers -> sharded counting insert.  Collective: every
        rank calls it once per batch (an empty batch where a rank has run out of reads)."""
        km = self.kmer_fn(text)
        if self.reserve_from_estimate and self.hll is not None:
            # the reference sizes its counting table from a HyperLogLog estimate of the distinct k-mers seen so far (+ the
            # estimator's standard error 1.04 / sqrt(m)) before it inserts: every rank updates its registers with ITS k-mers, the
            # registers are merged over all ranks (all-reduce(max): hyperloglog64.hpp:477-484 merge_distributed / estimate_global) and
            # every rank reserves its share of the global estimate (estimate_average_per_rank :487-489).  Collective: all ranks, every batch.
            from .hll import estimate_average_per_rank
            if len(km):
                self.hll.update(km)
            est = estimate_average_per_rank(self.hll, self.st.group)
            self.st.local.reserve(int(est * (1.0 + self.hll.est_error_rate)))
        self.st.insert_counts(km, chunks=self.chunks)
        self.total_kmers += len(km)
        return len(km)

    def cycle(self, queries):
        """count -> find -> erase -> count over `queries` (this rank's sample; collective).  Returns per-rank numbers:
        hits of the first count, sum of the counts find returned, keys erased on this rank's local table, hits afterwards."""
        import time
        cuda = getattr(queries, "is_cuda", False)

        def lap(t0):
            if cuda:
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3

        ms = {}
        t0 = time.perf_counter(); _, c1 = self.st.count(queries); ms["count"] = lap(t0)
        t0 = time.perf_counter(); _, vals, found = self.st.find(queries); ms["find"] = lap(t0)
        t0 = time.perf_counter(); erased = self.st.erase(queries); ms["erase"] = lap(t0)
        t0 = time.perf_counter(); _, c2 = self.st.count(queries); ms["count2"] = lap(t0)
        if hasattr(self.st, "synchronize"):
            self.st.synchronize()
        occ = (vals.to(torch.int64) & 0xFFFFFFFF) * found.to(torch.int64)
        return {"count_hits": int(c1.sum()), "find_hits": int(found.sum()), "find_occurrences": int(occ.sum()),
                "erased_local": int(erased), "count_hits_after": int(c2.sum()), "phase_ms": {k: round(v, 3) for k, v in ms.items()}}

    def size(self):
        return self.st.size()


def read_positions(n_reads, read_len, genome_len, read_seed):
    """(starts, strand flags) of the reads synthetic_reads draws for `read_seed`"""
    rng = np.random.default_rng(read_seed)
    starts = rng.integers(0, genome_len - read_len, n_reads, dtype=np.int32)
    rev = rng.integers(0, 2, n_reads, dtype=np.uint8).astype(bool)
    return starts, rev


def synthetic_reads(n_reads, read_len=150, genome_len=1_000_000, genome_seed=7, read_seed=8):
    """error-free reads of a random genome on both strands, with what is needed to PREDICT every k-mer's count: returns
    (sequence lines as synthetic_read_sequences lays them out, genome codes, read starts, strand flags).  Ranks of a
    distributed run share genome_seed and differ in read_seed."""
    genome = np.random.default_rng(genome_seed).integers(0, 4, genome_len, dtype=np.uint8)
    starts, rev = read_positions(n_reads, read_len, genome_len, read_seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((n_reads, read_len + 1), dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int32)[None, :]
    for c0 in range(0, n_reads, 100_000):
        c1 = min(n_reads, c0 + 100_000)
        r = genome[starts[c0:c1, None] + ar]
        rr = rev[c0:c1]
        r[rr] = (3 - r[rr])[:, ::-1]
        out[c0:c1, :read_len] = lut[r]
    out[:, read_len] = 10
    return out.reshape(-1), genome, starts, rev


def fastq_from_sequence_lines(seq_lines, n_reads, read_len):
    """fixed-width FASTQ text around the given sequence lines (the layout of synthetic_fastq_fixed)"""
    seq = seq_lines.reshape(n_reads, read_len + 1)
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


def expected_kmer_coverage(genome_len, starts, read_len, k):
    """cov[p] = number of reads that contain the k-mer starting at genome position p (either strand: the canonical k-mer of a
    window is the same on both): the count the table must hold for that k-mer when the genome's k-mers are all distinct"""
    d = np.zeros(genome_len + 1, dtype=np.int64)
    np.add.at(d, starts, 1)
    np.add.at(d, starts + (read_len - k + 1), -1)
    return np.cumsum(d)[:genome_len]


def canonical_kmers_at(genome, positions, k):
    """packed canonical k-mers of the genome windows starting at `positions` (numpy; A0 C1 G2 T3, first base most significant)"""
    pos = np.asarray(positions, dtype=np.int64)
    fw = np.zeros(len(pos), dtype=np.uint64)
    rc = np.zeros(len(pos), dtype=np.uint64)
    for j in range(k):
        c = genome[pos + j].astype(np.uint64)
        fw = (fw << np.uint64(2)) | c
        rc |= (np.uint64(3) - c) << np.uint64(2 * j)
    return np.minimum(fw, rc)


def synthetic_fastq_device(n_reads, read_len, genome_len, genome_seed, read_seed, device, chunk=1_000_000):
    """the reads of synthetic_reads / fastq_from_sequence_lines generated ON THE GPU (torch): for inputs of BenchmarkKmerCounter's
    size (configs[4]: 6.25 Gbp per rank = 4.2e7 reads, 13 GB of FASTQ text) host generation would take minutes.  Returns
    (fastq text: uint8 CUDA tensor of n_reads fixed-width records, genome codes: uint8 CUDA tensor, read starts: int64 CUDA tensor).
    The generator is torch's (not numpy's): the reads differ from synthetic_reads', the prediction of every count from the read
    positions works the same way."""
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev); g.manual_seed(genome_seed)
    genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev, generator=g)
    g.manual_seed(read_seed)
    starts = torch.randint(0, genome_len - read_len, (n_reads,), dtype=torch.int64, device=dev, generator=g)
    rev = torch.randint(0, 2, (n_reads,), dtype=torch.uint8, device=dev, generator=g).bool()
    w = 1 + 11 + 1 + (read_len + 1) + 2 + (read_len + 1)
    out = torch.empty((n_reads, w), dtype=torch.uint8, device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, dtype=torch.int64, device=dev)[None, :]
    o = 13 + read_len + 1
    for c0 in range(0, n_reads, chunk):
        c1 = min(n_reads, c0 + chunk)
        blk = out[c0:c1]
        blk[:, 0] = ord("@")
        ids = torch.arange(c0, c1, dtype=torch.int64, device=dev)
        for d in range(11):
            blk[:, 11 - d] = (ord("0") + (ids % 10)).to(torch.uint8)
            ids = ids // 10
        blk[:, 12] = 10
        r = genome[starts[c0:c1, None] + ar]
        rr = rev[c0:c1]
        r[rr] = (3 - r[rr]).flip(1)
        blk[:, 13:13 + read_len] = lut[r.long()]
        blk[:, 13 + read_len] = 10
        blk[:, o] = ord("+"); blk[:, o + 1] = 10
        blk[:, o + 2:o + 2 + read_len] = ord("I")
        blk[:, o + 2 + read_len] = 10
        del r
    return out.reshape(-1), genome, starts
