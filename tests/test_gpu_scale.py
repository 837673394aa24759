"""GPU: BASELINE.json's full sizes through size-independent properties (the oracle cannot run 1e8 keys in
seconds): exact insert counts, every inserted key found with its first value, misses missed, erase ->
count == 0, displacement histogram mass == size, the Robin Hood invariant of the exported info array, and
insertion-order independence of the info array (the reference's canonical-layout property).
A 2^24-key prefix is additionally compared with the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import kmerhash_amd as kh  # noqa: E402
from kmerhash_amd import workloads as W  # noqa: E402


def dev(a):
    return torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32)).cuda()


def check_rh_info_invariants(info, size):
    """valid Robin Hood info array: occupied bytes 0x80|d; d never grows by more than 1 from the previous slot;
    an occupied slot after an empty one has d == 0"""
    occ = info >= 0x80
    assert int(occ.sum()) == size
    assert ((info == 0) | occ).all()
    d = (info & 0x7F).astype(np.int16)
    prev_d = np.roll(d, 1)
    prev_occ = np.roll(occ, 1)
    assert (d[occ & ~prev_occ] == 0).all()
    both = occ & prev_occ
    assert (d[both] <= prev_d[both] + 1).all()


def test_config1_full_size_rh_murmur():
    """configs[1]: 1e8 random 64-bit k-mers, max load 0.8, murmur3, Robin Hood"""
    n, nq = 100_000_000, 10_000_000
    keys = W.distinct_u64(n, seed=1)
    vals = np.arange(n, dtype=np.uint32)
    dk, dv = dev(keys), dev(vals)
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    assert t.insert(dk, dv) == n
    assert t.size() == n and t.capacity() == 1 << 27
    assert t.load_thresholds() == (int(np.float32(1 << 27) * np.float32(0.35)), 107374184)
    hist = t.displacement_histogram()
    assert int(hist.sum()) == n
    q = dev(keys[:nq])
    fk, fv = t.find(q)
    assert fk.numel() == nq
    assert torch.equal(fk, q) and torch.equal(fv, dv[:nq])          # first values round-trip, query order kept
    miss = dev(W.distinct_u64(nq, seed=977))
    assert int(t.count(miss).sum().item()) == 0
    # duplicates only: nothing inserted, table doubles? no: size < max_load, capacity stays
    assert t.insert(dk[:nq], dv[:nq]) == 0 and t.capacity() == 1 << 27
    info = t.export_info()
    check_rh_info_invariants(info, n)
    assert np.array_equal(np.bincount(info[info >= 0x80] & 0x7F, minlength=128).astype(np.uint64), hist)
    assert t.erase(q) == nq
    assert t.size() == n - nq and t.capacity() == 1 << 27
    assert int(t.count(q).sum().item()) == 0
    assert int(t.count(dk[nq:2 * nq]).sum().item()) == nq
    check_rh_info_invariants(t.export_info(), n - nq)
    t.close()


@pytest.mark.parametrize("cls", [kh.hashmap_robinhood_doubling, kh.hashmap_linearprobe_doubling])
def test_config4_per_gpu_size_sliced_histogram(cls):
    """configs[3] puts 1.25e8 keys on every GPU: capacity 2^28 = 2^17 chunks, more partitions than the single-sweep histogram
    has bins (2^16), so the id space is swept in two slices; then a second batch into the non-empty table (merge kernel) and
    an erase.  Size-independent checks, plus the Robin Hood invariants of the exported info array"""
    n, n2, nq = 125_000_000, 20_000_000, 5_000_000
    keys = W.distinct_u64(n + n2, seed=11)
    dk = dev(keys)
    dv = torch.arange(n + n2, dtype=torch.int32, device="cuda")
    t = cls(128, 0.35, 0.8)
    assert t.insert(dk[:n], dv[:n]) == n
    assert t.size() == n and t.capacity() == 1 << 28
    fk, fv = t.find(dk[:nq])
    assert torch.equal(fk, dk[:nq]) and torch.equal(fv, dv[:nq])
    assert int(t.count(dk[n:n + nq]).sum().item()) == 0
    # second batch: 2e7 new keys + 1e6 repeats of keys of the table
    k2 = torch.cat([dk[n:], dk[:1_000_000]]); v2 = torch.cat([dv[n:], dv[:1_000_000] + 7])
    assert t.insert(k2, v2) == n2 and t.size() == n + n2 and t.capacity() == 1 << 28
    fk, fv = t.find(dk[:nq])
    assert torch.equal(fv, dv[:nq])                                  # first values kept
    fk, fv = t.find(dk[n + n2 - nq:])
    assert torch.equal(fk, dk[n + n2 - nq:]) and torch.equal(fv, dv[n + n2 - nq:])
    assert t.erase(dk[100:100 + nq]) == nq and int(t.count(dk[100:100 + nq]).sum().item()) == 0
    assert t.size() == n + n2 - nq
    if cls is kh.hashmap_robinhood_doubling:
        check_rh_info_invariants(t.export_info(), n + n2 - nq)
    t.close()


def test_exact_max_load_capacity_edge_full_size():
    """N' = 107374184 = size_t(float(2^27) * 0.8f): the table ends at load exactly 0.800 in 2^27 buckets only
    because the stream ends with the element that reaches max_load (SURVEY §7 capacity rule)"""
    n = 107_374_184
    keys = W.distinct_u64(n, seed=5)
    dk = dev(keys)
    dv = torch.arange(n, dtype=torch.int32, device="cuda")
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    assert t.insert(dk, dv) == n
    assert t.capacity() == 1 << 27 and t.size() == n
    assert t.insert(dk[:1], dv[:1]) == 0          # one more call (a duplicate) doubles
    assert t.capacity() == 1 << 28
    assert int(t.count(dk[: 1 << 20]).sum().item()) == 1 << 20
    t.close()


@pytest.mark.parametrize("cls,kind", [(kh.hashmap_robinhood_doubling, 0), (kh.hashmap_linearprobe_doubling, 1)])
def test_config_w1_prefix_bit_exact_vs_oracle(oracle, cls, kind):
    """benchmark_hashtables shape (mean multiplicity 5.5): 2^24 pairs compared with the oracle bit for bit"""
    n = 1 << 24
    keys, vals = W.w1_benchmark_hashtables(n, seed=23)
    g = cls(128, 0.35, 0.8)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    assert (g.size(), g.capacity()) == (o.size(), o.capacity())
    if kind == 0:
        assert np.array_equal(g.export_info(), o.export_info())
    a, b = g.sorted_items(), o.sorted_items()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    q = W.queries_hits_and_misses(keys, 1 << 20, 0.5)
    assert np.array_equal(g.count(q), o.count(q))
    fk, fv = g.find(q)
    ok, ov = o.find_compact(q)
    assert np.array_equal(fk, ok) and np.array_equal(fv, ov)
    assert g.erase(q) == o.erase(q)
    assert (g.size(), g.capacity()) == (o.size(), o.capacity())
    if kind == 0:
        assert np.array_equal(g.export_info(), o.export_info())
    g.close()


def test_config3_lp_31mers_5x_full_size():
    """configs[2]: LP table, 2e7 distinct 62-bit 31-mers x5 = 1e8 inserts; insert + count"""
    keys, vals = W.w3_kmers_5x(20_000_000, 5, seed=3)
    t = kh.hashmap_linearprobe_doubling(128, 0.35, 0.8)
    assert t.insert(dev(keys), dev(vals)) == 20_000_000
    assert t.size() == 20_000_000 and t.capacity() == 1 << 25
    # first value wins: a stable argsort keeps stream positions ascending inside a key's 5 copies,
    # so every 5th entry is the key's first occurrence in the stream
    order = np.argsort(keys, kind="stable")
    sk, sv = keys[order[::5]], vals[order[::5]]
    fv, found = t.find_values(dev(sk[: 1 << 20].copy()))
    assert bool(found.all())
    assert np.array_equal(fv.cpu().numpy().view(np.uint32), sv[: 1 << 20])
    assert int(t.count(dev(W.distinct_u64(2_000_000, seed=4242) | np.uint64(1 << 63))).sum().item()) == 0
    t.close()


def test_info_array_is_insertion_order_independent():
    keys, vals = W.w1_benchmark_hashtables(3_000_000, seed=41)
    p = W.shuffle_perm(len(keys), 5)
    a = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    b = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    a.insert(dev(keys), dev(vals))
    for i in range(0, len(keys), 700_001):                 # other order, several batches
        b.insert(dev(keys[p][i:i + 700_001].copy()), dev(vals[p][i:i + 700_001].copy()))
    assert a.capacity() == b.capacity() and a.size() == b.size()
    assert np.array_equal(a.export_info(), b.export_info())
    assert np.array_equal(a.sorted_items()[0], b.sorted_items()[0])
    er = np.unique(keys)[:200_000]
    a.erase(dev(er)); b.erase(dev(er[::-1].copy()))
    assert np.array_equal(a.export_info(), b.export_info())
    a.close(); b.close()
