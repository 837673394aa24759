"""GPU: k-mer generation front end + counting (BenchmarkKmerCounter shape).  The reference's parser
(kmerind) is absent, so parity here is against a plain numpy statement of this library's k-mer definition:
PARITY UNPINNED with respect to the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from kmerhash_amd import kmers as KM  # noqa: E402


from oracle.kmers_np import np_kmers  # noqa: E402  (numpy statement of the k-mer definition: test infrastructure)


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


def _run_counter(*flags):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "benchmark", "kmer_counter.py")] + list(flags), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_kmer_counter_cycle_small():
    """configs[4] shape on one GPU, small: FASTQ batches -> canonical 31-mers (farmhash table, capacity 128, doubling under load) ->
    counting insert; sizes / total / 10^5 sampled counts against the prediction from the read positions; then the count / find /
    erase / count cycle of BenchmarkKmerIndex.cpp:787-843"""
    d = _run_counter("--reads", "60000", "--genome", "400000", "--batches", "5", "--cycle", "--verify", "--sample-ratio", "50")
    assert d["ok"] and d["verify"]["ok"] and d["cycle"]["ok"], d
    caps = d["capacity_per_batch_rank0"]
    assert caps[0] > 128 and caps == sorted(caps) and caps[-1] >= 1 << 19          # grew under load, batch after batch
    assert d["verify"]["total_kmers"] == 60000 * 120 and d["cycle"]["size_after"] < d["distinct_global"]


def test_kmer_counter_hll_reserve():
    d = _run_counter("--reads", "40000", "--genome", "300000", "--batches", "4", "--verify", "--hll-reserve", "--hash", "murmur3avx64")
    assert d["ok"] and d["verify"]["ok"], d


@pytest.mark.timeout(1800)
def test_kmer_counter_cycle_5e8_kmers():
    """one GPU, >= 5e8 k-mers (4.2 M reads of 150 bp, 8 file batches of 160 MB FASTQ text), farmhash, the table
    doubling from 128 buckets under load; counts of 10^5 sampled k-mers, the total and the distinct count are checked against
    the numpy prediction from the read positions; then the find + erase + count cycle over every 100th k-mer of a batch"""
    d = _run_counter("--reads", "4200000", "--genome", "30000000", "--batches", "8", "--cycle", "--verify")
    assert d["ok"] and d["verify"]["ok"] and d["cycle"]["ok"], d
    assert d["verify"]["total_kmers"] == 4_200_000 * 120 >= 500_000_000
    assert d["capacity_per_batch_rank0"][-1] >= 1 << 25
