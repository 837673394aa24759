"""GPU: k-mer generation front end + counting (BenchmarkKmerCounter shape).  The reference's parser
(kmerind) is absent, so parity here is against a plain numpy statement of this library's k-mer definition:
PARITY UNPINNED with respect to the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from kmerhash_amd import kmers as KM  # noqa: E402


def np_kmers(seq, k, canonical):
    """numpy statement: windows of k valid bases, first base most significant, A0 C1 G2 T3"""
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


@pytest.mark.parametrize("k", [1, 5, 21, 31, 32])
@pytest.mark.parametrize("canonical", [False, True])
def test_kmers_match_numpy_statement(k, canonical):
    rng = np.random.default_rng(k)
    seq = np.frombuffer(b"ACGTNacgtn\n", dtype=np.uint8)[rng.choice(11, 20_000, p=[.24, .24, .24, .24, .005, .005, .005, .005, .005, .005, .01])]
    exp = np_kmers(seq, k, canonical)
    got = KM.kmers_from_sequence(seq, k, canonical)
    assert np.array_equal(got, exp)
    got_d = KM.kmers_from_sequence(torch.from_numpy(seq.copy()).cuda(), k, canonical)
    assert np.array_equal(got_d.cpu().numpy().view(np.uint64), exp)
    for m in (0, k - 1, k, 63, 64, 65, 64 + k - 1, 64 + k):     # strip boundaries
        if m >= 0:
            assert np.array_equal(KM.kmers_from_sequence(seq[:m], k, canonical), np_kmers(seq[:m], k, canonical))


def test_counter_on_synthetic_fastq(tmp_path):
    fq = KM.synthetic_fastq(3000, 150, 50_000, seed=3)
    seq = KM.sequences_from_fastq(fq)
    lines = fq.split(b"\n")[1::4]
    assert bytes(seq) == b"".join(l + b"\n" for l in lines)
    kc = KM.KmerCounter(31, canonical=True, hash="farm")
    half = len(fq) // 2
    cut = fq.rfind(b"\n@r", 0, half) + 1                      # two file batches, like the reference's batched reading
    n1 = kc.add_fastq(fq[:cut])
    n2 = kc.add_fastq(fq[cut:])
    exp_k, exp_c = np.unique(np_kmers(seq, 31, True), return_counts=True)
    assert n1 + n2 == int(exp_c.sum())
    k, v = kc.counts()
    o = np.argsort(k)
    assert np.array_equal(k[o], exp_k) and np.array_equal(v[o], exp_c.astype(np.uint32))
    # output file: packed (u64 k-mer, u16 count) tuples
    p = str(tmp_path / "counts.bin")
    assert kc.write(p) == len(exp_k)
    raw = np.fromfile(p, dtype=np.uint8)
    assert len(raw) == 10 * len(exp_k)
    rec = np.frombuffer(raw.tobytes(), dtype=np.dtype([("kmer", "<u8"), ("count", "<u2")]))
    oo = np.argsort(rec["kmer"])
    assert np.array_equal(rec["kmer"][oo], exp_k) and np.array_equal(rec["count"][oo], exp_c.astype(np.uint16))
    kc.close()


@pytest.mark.parametrize("k", [3, 31])
def test_fastq_record_structure_is_resolved_on_the_gpu(k):
    """kh_kmers_from_fastq: only the sequence lines (line 1 mod 4) yield k-mers, although ids, '+' lines and quality strings
    are written with the letters ACGT here; reads of varying length cross the 2048-byte tiles of the newline scan; the last
    record has no trailing newline; compared with the host statement (split lines, take [1::4])"""
    rng = np.random.default_rng(k)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    recs = []
    for i in range(4000):
        ln = int(rng.integers(0, 300))
        seq = lut[rng.choice(5, ln, p=[.245, .245, .245, .245, .02])].tobytes()
        qual = lut[rng.integers(0, 4, ln)].tobytes()                 # quality text that looks like bases
        recs.append(b"@ACGTACGTACGTACGTACGTACGTACGTACGTACGT_" + str(i).encode() + b"\n" + seq + b"\n+ACGTACGTACGTACGTACGTACGTACGTACGT\n" + qual + b"\n")
    fq = b"".join(recs)[:-1]
    exp = np_kmers(KM.sequences_from_fastq(fq), k, True)
    got = KM.kmers_from_fastq(fq, k, True)
    assert np.array_equal(got, exp)
    got_d = KM.kmers_from_fastq(torch.from_numpy(np.frombuffer(fq, dtype=np.uint8).copy()).cuda(), k, True)
    assert np.array_equal(got_d.cpu().numpy().view(np.uint64), exp)
    for cut in (0, 1, 5, 2047, 2048, 2049, 4096 + 17):                # truncated inputs: tile edges, mid-record ends
        part = fq[:cut]
        assert np.array_equal(KM.kmers_from_fastq(part, k, True), np_kmers(KM.sequences_from_fastq(part), k, True)), cut


def test_fasta_sequences():
    fa = b">chr1 test\nACGTAC\nGTNNAC\n>chr2\nTTTTGGGGCC\n"
    s = KM.sequences_from_fasta(fa)
    assert bytes(s) == b"\nACGTACGTNNAC\nTTTTGGGGCC"
    assert np.array_equal(KM.kmers_from_sequence(s, 4, False), np_kmers(s, 4, False))
