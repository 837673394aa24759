"""GPU parity tests proper: the HIP tables (through the C-ABI) against the CPU oracle on the same seeded
inputs.  Integer/byte/index work: every comparison is bit-exact.

What is compared: size, capacity, the Robin Hood info array (canonical: it depends only on
the key set and the capacity), the sorted (key, first-wins value) set, per-query count, the compacted find
result in query order, erase counts, post-erase state.  For the LP table the slot layout depends on the
insertion order in the reference, so results - not layout - are compared.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import kmerhash_amd as kh  # noqa: E402
from kmerhash_amd import workloads as W  # noqa: E402

KINDS = [("rh", kh.hashmap_robinhood_doubling, 0), ("lp", kh.hashmap_linearprobe_doubling, 1)]
HASHES = [("murmur3avx64", 1), ("murmur", 2), ("farm", 3), ("identity", 0)]


def dev(a):
    if a.dtype == np.uint64:
        return torch.from_numpy(a.view(np.int64)).cuda()
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).cuda()
    return torch.from_numpy(a).cuda()


def host(t, dtype):
    return t.cpu().numpy().view(dtype)


def check_state(g, o, kind, layout=True):
    assert g.size() == o.size()
    assert g.capacity() == o.capacity()
    assert g.load_thresholds() == (o.min_load(), o.max_load())
    gk, gv = g.sorted_items()
    ok, ov = o.sorted_items()
    assert np.array_equal(gk, ok)
    assert np.array_equal(gv, ov)
    ginfo = g.export_info()
    oinfo = o.export_info()
    if kind == 0 and layout:
        assert np.array_equal(ginfo, oinfo), "Robin Hood info array differs from the oracle"
        assert np.array_equal(g.displacement_histogram(), o.displacement_histogram())
    if kind == 1:
        # same occupied slot set is NOT required for LP (tombstones / order), but the count must agree
        assert int((ginfo < 0x40).sum()) == o.size()
    # every occupied slot holds a key whose probe sequence reaches it (self-consistency through count)
    if g.size():
        k, _ = g.to_vector()
        assert g.count(k).all()


def check_queries(g, o, q):
    assert np.array_equal(g.count(q), o.count(q))
    assert np.array_equal(host(g.count(dev(q)), np.uint8), o.count(q))
    fk, fv = g.find(dev(q))
    ok, ov = o.find_compact(q)
    assert np.array_equal(host(fk, np.uint64), ok)
    assert np.array_equal(host(fv, np.uint32), ov)
    fk2, fv2 = g.find(q)
    assert np.array_equal(fk2, ok) and np.array_equal(fv2, ov)
    vals, found = g.find_values(q)
    ovals, ofound = o.find(q)
    assert np.array_equal(found, ofound)
    assert np.array_equal(vals[found == 1], ovals[ofound == 1])


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("hname,hid", HASHES)
@pytest.mark.parametrize("n", [0, 1, 7, 1000, 100_000])
def test_insert_find_count_erase(oracle, kname, cls, kind, hname, hid, n):
    keys, vals = W.w1_benchmark_hashtables(n, seed=23) if n else (np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    if hname == "identity":
        keys = W.splitmix64(keys)   # identity on clustered k-mers would overflow the 7-bit distance; spread them
    g = cls(128, 0.35, 0.8, hash=hname, seed=43)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8, hid, 43)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    check_state(g, o, kind)
    q = W.queries_hits_and_misses(keys, max(2 * n // 3, 4), 0.5) if n else W.distinct_u64(5)
    check_queries(g, o, q)
    e = q[: len(q) // 2]
    assert g.erase(dev(e)) == o.erase(e)
    check_state(g, o, kind)
    check_queries(g, o, q)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_incremental_batches_host_arrays(oracle, kname, cls, kind):
    """several insert batches with overlaps, host (numpy) inputs, growth across many doublings"""
    g = cls(128, 0.35, 0.8)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    allk, allv = W.w1_benchmark_hashtables(60_000, seed=5)
    cuts = [0, 1, 2, 10, 103, 104, 1000, 5000, 30_000, 60_000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert g.insert(allk[a:b], allv[a:b]) == o.insert(allk[a:b], allv[a:b])
        check_state(g, o, kind)
    # re-insert everything: all duplicates, first values stay
    assert g.insert(allk, allv + 7) == o.insert(allk, allv + 7) == 0
    check_state(g, o, kind)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_capacity_rule_edges(oracle, kname, cls, kind):
    """doubling happens on ANY insert call made while size >= max_load, duplicates included
    (hashmap_robinhood.hpp:530); SURVEY Appendix B: 102 distinct in cap 128 stay, a following duplicate doubles."""
    keys = W.distinct_u64(300, seed=9)
    vals = np.arange(300, dtype=np.uint32)
    for tail in (0, 1):
        g = cls(128, 0.35, 0.8)
        o = oracle.OracleTable(kind, 128, 0.35, 0.8)
        k = np.concatenate([keys[:102], keys[:tail]])
        v = np.concatenate([vals[:102], vals[:tail] + 900])
        assert g.insert(k, v) == o.insert(k, v) == 102
        assert g.capacity() == o.capacity() == (128 if tail == 0 else 256)
        check_state(g, o, kind)
        # next batch of one duplicate doubles a table that sits exactly at max_load
        assert g.insert(keys[:1], vals[:1]) == o.insert(keys[:1], vals[:1]) == 0
        assert g.capacity() == o.capacity() == 256
        check_state(g, o, kind)
        g.close()
    # the new key that reaches max_load is the LAST call vs. is followed by a duplicate
    for order in ("last", "followed"):
        g = cls(128, 0.35, 0.8)
        o = oracle.OracleTable(kind, 128, 0.35, 0.8)
        if order == "last":
            k = np.concatenate([keys[:50], keys[:50], keys[50:102]])
        else:
            k = np.concatenate([keys[:102], keys[5:6]])
        v = np.arange(len(k), dtype=np.uint32)
        assert g.insert(k, v) == o.insert(k, v)
        assert g.capacity() == o.capacity()
        check_state(g, o, kind)
        g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_erase_shrink_rules(oracle, kname, cls, kind):
    """erase 100 of 102 via erase(begin,end): RH stays at cap 256, LP shrinks to cap 2."""
    keys = W.distinct_u64(103, seed=4)
    vals = np.arange(103, dtype=np.uint32)
    g = cls(128, 0.35, 0.8)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    assert g.insert(keys, vals) == o.insert(keys, vals)
    assert g.capacity() == o.capacity() == 256
    assert g.erase(keys[:101]) == o.erase(keys[:101]) == 101
    assert g.capacity() == o.capacity() == (256 if kind == 0 else 2)
    check_state(g, o, kind)
    check_queries(g, o, keys)
    # next insert: the LP table of 2 buckets holding 2 elements doubles on the next call
    assert g.insert(keys[:5], vals[:5]) == o.insert(keys[:5], vals[:5])
    check_state(g, o, kind)
    # single-key erase halves when size < min_load
    for k in keys[:8]:
        assert g.erase_one(int(k)) == o.erase_one(int(k))
        assert g.capacity() == o.capacity()
    check_state(g, o, kind)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_update_reserve_rehash_clear(oracle, kname, cls, kind):
    keys, vals = W.w1_benchmark_hashtables(20_000, seed=77)
    g = cls(128, 0.35, 0.8)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    g.reserve(5000); o.reserve(5000)
    assert g.capacity() == o.capacity()
    g.insert(keys[:10_000], vals[:10_000]); o.insert(keys[:10_000], vals[:10_000])
    # update == insert-or-overwrite in batch order (last value wins)
    g.update(keys[5_000:], vals[5_000:] + 1_000_000)
    for k, v in zip(keys[5_000:], vals[5_000:] + 1_000_000):
        o.update_one(int(k), int(v))
    check_state(g, o, kind)
    g.rehash(1 << 17); o.rehash(1 << 17)
    check_state(g, o, kind)
    g.clear(); o.clear()
    assert g.size() == o.size() == 0 and g.capacity() == o.capacity()
    g.insert(keys, vals); o.insert(keys, vals)
    check_state(g, o, kind)
    g.close()


def test_pairs_layout(oracle):
    """std::pair<uint64_t,uint32_t> arrays (16 B, value at +8) in and out"""
    keys, vals = W.w1_benchmark_hashtables(5000, seed=3)
    pairs = np.zeros((len(keys), 2), dtype=np.uint64)
    pairs[:, 0] = keys
    pairs[:, 1] = vals
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    o = oracle.OracleTable(0, 128, 0.35, 0.8)
    assert g.insert(pairs) == o.insert(keys, vals)
    check_state(g, o, 0)
    g2 = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    assert g2.insert(torch.from_numpy(pairs.view(np.int64)).cuda()) == o.size()
    check_state(g2, o, 0)
    g.close(); g2.close()


def test_probe_overflow_is_refused(oracle):
    """identity hash on dense small keys drives the probe distance past 127: the reference asserts
    (hashmap_robinhood.hpp:556) or silently corrupts under NDEBUG; we refuse and keep the table intact."""
    g = kh.hashmap_robinhood_doubling(1 << 20, 0.35, 0.8, hash="identity")
    good = W.distinct_u64(1000, seed=8)
    g.insert(good, np.arange(1000, dtype=np.uint32))
    bad = (np.arange(200_000, dtype=np.uint64) % np.uint64(1500)) * np.uint64(1 << 20) + np.uint64(77)
    with pytest.raises(kh.KhError) as ei:
        g.insert(bad, np.zeros(len(bad), dtype=np.uint32))
    assert "KH_ERR_PROBE_OVERFLOW" in str(ei.value)
    assert g.size() == 1000 and g.count(good).all()
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_small_batches_are_applied_in_place(oracle, kname, cls, kind):
    """a handful of keys (<= 16, no doubling possible) is applied by one lane with the reference's single-key algorithms
    (k_small_batch) instead of a re-layout of the table: insert / update / std::plus / erase, duplicates inside the batch,
    existing and new keys; state bit-exact after every call"""
    keys = W.distinct_u64(300_000, seed=41)
    vals = np.arange(len(keys), dtype=np.uint32)
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    assert g.insert(dev(keys[:250_000]), dev(vals[:250_000])) == o.insert(keys[:250_000], vals[:250_000])
    g.profile_enable(True)
    rng = np.random.default_rng(8)
    for step in range(60):
        m = int(rng.integers(1, 17))
        ks = np.concatenate([keys[rng.integers(0, 250_000, m // 2)], keys[250_000 + rng.integers(0, 50_000, m - m // 2)]])
        ks = ks[rng.permutation(len(ks))]
        if m > 3:
            ks[1] = ks[0]                                            # a duplicate inside the batch
        vs = rng.integers(0, 2**32, len(ks), dtype=np.uint32)
        op = step % 4
        if op == 0:
            assert g.insert(dev(ks), dev(vs)) == o.insert(ks, vs), step
        elif op == 1:
            assert g.insert(ks, vs) == o.insert(ks, vs), step          # host buffers
        elif op == 2:
            g.update(ks, vs)
            for k, v in zip(ks.tolist(), vs.tolist()):
                o.update_one(k, v)
        else:
            if kind == 0:
                assert g.erase(dev(ks)) == o.erase(ks), step
            else:
                assert g.erase(ks) == o.erase(ks), step
        check_state(g, o, kind)
    prof = g.profile()
    assert "k_small_batch" in prof and "k_chunk_place" not in prof and "k_rebuild_fused" not in prof and "k_insert_fused" not in prof, prof
    # std::plus on existing and new keys (numpy statement)
    ks = np.concatenate([keys[:3], keys[:3], W.distinct_u64(2, seed=99)])
    before = dict(zip(*[x.tolist() for x in g.sorted_items()]))
    assert g.insert_reduce_plus(dev(ks)) == 2
    after = dict(zip(*[x.tolist() for x in g.sorted_items()]))
    for k in keys[:3].tolist():
        assert after[k] == (before[k] + 2) & 0xFFFFFFFF
    for k in W.distinct_u64(2, seed=99).tolist():
        assert after[k] == 1
    g.close()


def test_small_batch_defers_probe_overflow_to_the_general_path(oracle):
    """128 keys with one home bucket occupy distances 0..127; the 129th, inserted alone, would need distance 128: the dry run of
    the in-place path sees it, the general path refuses, the table is unchanged"""
    g = kh.hashmap_robinhood_doubling(1 << 14, 0.35, 0.8, hash="identity")
    same_home = np.uint64(4000) + (np.arange(1, 130, dtype=np.uint64) << np.uint64(32))
    assert g.insert(same_home[:100], np.arange(100, dtype=np.uint32)) == 100
    for i in range(100, 128):                                          # single-key inserts: the in-place path
        assert g.insert(same_home[i:i + 1], np.array([i], dtype=np.uint32)) == 1
    assert int(np.flatnonzero(g.displacement_histogram())[-1]) == 127
    with pytest.raises(kh.KhError) as ei:
        g.insert(same_home[128:129], np.array([128], dtype=np.uint32))
    assert "KH_ERR_PROBE_OVERFLOW" in str(ei.value)
    assert g.size() == 128 and g.count(same_home[:128]).all() and not g.count(same_home[128:129]).any()
    # a second element of the bucket in front of the pile displaces it by one slot: its last element would pass distance 127
    front = np.uint64(3999) + (np.arange(1, 3, dtype=np.uint64) << np.uint64(32))
    assert g.insert(front[:1], np.array([1], dtype=np.uint32)) == 1               # slot 3999 was free
    with pytest.raises(kh.KhError):
        g.insert(front[1:], np.array([2], dtype=np.uint32))
    assert g.size() == 129 and g.count(same_home[:128]).all() and g.count(front).tolist() == [1, 0]
    assert g.erase(front[:1]) == 1 and g.erase(same_home[5:6]) == 1               # backward shifts in place
    assert g.insert(same_home[128:129], np.array([128], dtype=np.uint32)) == 1   # now there is room: distance 127 again
    assert int(np.flatnonzero(g.displacement_histogram())[-1]) == 127 and g.size() == 128
    g.close()


def test_hash_batch_matches_oracle(oracle):
    keys = np.concatenate([np.arange(0, 1003, dtype=np.uint64), W.distinct_u64(10_000, seed=2),
                           np.array([0xFFFFFFFFFFFFFFFF, 1 << 63, 1], dtype=np.uint64)])
    for hname, hid in HASHES:
        for seed in (43, 0, 9876543):
            got = kh.hash_batch(keys, hash=hname, seed=seed)
            assert np.array_equal(got, oracle.hash_batch(hid, seed, keys)), (hname, seed)
            got_d = kh.hash_batch(dev(keys), hash=hname, seed=seed)
            assert np.array_equal(host(got_d, np.uint64), got)


def test_sentinel_key_values(oracle):
    """0 and 0xFFFF...F are ordinary keys (the LDS de-dup set uses all-ones as its empty marker internally)"""
    keys = np.array([0, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 1, 0, 2**63], dtype=np.uint64)
    vals = np.arange(len(keys), dtype=np.uint32)
    for _, cls, kind in KINDS:
        g = cls(128, 0.35, 0.8)
        o = oracle.OracleTable(kind, 128, 0.35, 0.8)
        assert g.insert(keys, vals) == o.insert(keys, vals) == 4
        check_state(g, o, kind)
        check_queries(g, o, np.array([0, 5, 0xFFFFFFFFFFFFFFFF, 2**63, 3], dtype=np.uint64))
        g.close()


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 8, 16])
def test_shard_permute_is_stable_partition_by_rank(oracle, p):
    """rank = murmur3(key, seed 9876543) & (p-1) (power of two) or % p; order kept inside a rank"""
    from kmerhash_amd.dist import GpuBackend, DIST_SEED
    keys, vals = W.w1_benchmark_hashtables(150_001, seed=8)
    be = GpuBackend(0)
    ok, ov, counts = be.shard(dev(keys), dev(vals), p)
    r = (oracle.hash_batch(1, DIST_SEED, keys) % np.uint64(p)).astype(np.int64)
    order = np.argsort(r, kind="stable")
    assert counts == np.bincount(r, minlength=p).tolist()
    assert be.shard_counts(dev(keys), p) == counts            # count-only mode (out_keys == NULL)
    assert np.array_equal(host(ok, np.uint64), keys[order])
    assert np.array_equal(host(ov, np.uint32), vals[order])
    be.table.close()


def test_sharded_table_single_rank(oracle):
    from kmerhash_amd.dist import GpuBackend, ShardedTable
    keys, vals = W.w1_benchmark_hashtables(50_000, seed=12)
    be = GpuBackend(0)
    st = ShardedTable(be)
    o = oracle.OracleTable(0, 128, 0.35, 0.8)
    assert st.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    check_state(be.table, o, 0)
    pk, c = st.count(dev(keys[:1000]))
    assert bool(c.all()) and st.size() == o.size()
    be.table.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_reducer_plus_kmer_counting(kname, cls, kind):
    """Reducer = std::plus (counting_batched_robinhood_map: insert(keys, T(1))): the value of a key is
    the wrapping 32-bit sum over its occurrences; checked against numpy (the reference semantics is a per-key sum)."""
    keys, _ = W.w1_benchmark_hashtables(300_000, seed=61)
    g = cls(128, 0.35, 0.8)
    uk, cnt = np.unique(keys[:200_000], return_counts=True)
    assert g.insert_reduce_plus(dev(keys[:200_000])) == len(uk)          # vals=None: every occurrence adds 1
    sk, sv = g.sorted_items()
    assert np.array_equal(sk, uk) and np.array_equal(sv, cnt.astype(np.uint32))
    # second batch: existing keys are increased in place, new ones appended; explicit (wrapping) values
    k2 = keys[150_000:]
    v2 = (np.arange(len(k2), dtype=np.uint64) * np.uint64(0x9E3779B1) % np.uint64(2**32)).astype(np.uint32)
    n_new = g.insert_reduce_plus(k2, v2)
    exp = dict(zip(uk.tolist(), cnt.astype(np.uint64).tolist()))
    new = 0
    for k, v in zip(k2.tolist(), v2.tolist()):
        if k not in exp:
            exp[k] = 0
            new += 1
        exp[k] = (exp[k] + v) & 0xFFFFFFFF
    assert n_new == new and g.size() == len(exp)
    sk, sv = g.sorted_items()
    ek = np.array(sorted(exp), dtype=np.uint64)
    assert np.array_equal(sk, ek)
    assert np.array_equal(sv, np.array([exp[int(k)] for k in ek], dtype=np.uint32))
    if kind == 0:
        from_scratch = cls(g.capacity(), 0.35, 0.8)
        from_scratch.insert(ek, np.zeros(len(ek), dtype=np.uint32))
        assert np.array_equal(g.export_info(), from_scratch.export_info())   # layout is the canonical one
        from_scratch.close()
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("seed", [1, 2])
def test_random_operation_sequences(oracle, kname, cls, kind, seed):
    """differential fuzz: random mixes of insert / update / reducer / count / find / erase / erase_one / reserve / rehash
    batches of random sizes on one table, state compared with the oracle after every step"""
    rng = np.random.default_rng(1000 * seed + kind)
    g = cls(128, 0.35, 0.8)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    universe = W.splitmix64(np.arange(60_000, dtype=np.uint64) + np.uint64(seed << 24))
    for step in range(60):
        op = int(rng.integers(0, 9))
        m = int(rng.choice([0, 1, 3, 50, 2000, 20_000]))
        ks = universe[rng.integers(0, len(universe), m)]
        vs = rng.integers(0, 2**32, m, dtype=np.uint32)
        if op <= 2:
            assert g.insert(dev(ks), dev(vs)) == o.insert(ks, vs), step
        elif op == 3:
            g.update(ks, vs)
            for k, v in zip(ks.tolist(), vs.tolist()):
                o.update_one(k, v)
        elif op == 4:
            check_queries(g, o, ks) if m else None
        elif op == 5:
            assert g.erase(dev(ks)) == o.erase(ks), step
        elif op == 6:
            for k in ks[:5]:
                assert g.erase_one(int(k)) == o.erase_one(int(k))
        elif op == 7:
            r = int(rng.integers(0, 50_000))
            g.reserve(r); o.reserve(r)
        else:
            b = int(rng.choice([1024, 4096, 1 << 16, 1 << 18]))
            if o.size() <= b * 0.5:
                g.rehash(b); o.rehash(b)
        check_state(g, o, kind)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("hname,hid", HASHES)
def test_bulk_build_into_empty_table_fused_path(oracle, kname, cls, kind, hname, hid):
    """few duplicates + empty table = the fused bulk-build kernel (de-dup, count, one-deep carry look-back, placement in
    one launch); 3 % duplicates exercise first-value-wins inside it; compared with the oracle bit for bit"""
    n = 300_000
    keys = W.distinct_u64(n, seed=77)
    keys[::33] = keys[5::33][: len(keys[::33])]            # ~3 % duplicates, first occurrence later or earlier in the stream
    vals = np.arange(n, dtype=np.uint32)
    g = cls(128, 0.35, 0.8, hash=hname, seed=43)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8, hid, 43)
    g.profile_enable(True)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    prof = g.profile()
    assert "k_build_fused" in prof and "k_dedup" not in prof, prof      # the fused launch was taken and accepted
    check_state(g, o, kind)
    check_queries(g, o, W.queries_hits_and_misses(keys, 50_000, 0.5))
    # reducer form through the same kernel
    g2 = cls(128, 0.35, 0.8, hash=hname, seed=43)
    assert g2.insert_reduce_plus(dev(keys)) == o.size()
    uk, cnt = np.unique(keys, return_counts=True)
    sk, sv = g2.sorted_items()
    assert np.array_equal(sk, uk) and np.array_equal(sv, cnt.astype(np.uint32))
    g.close(); g2.close()


@pytest.mark.parametrize("hname,hid", HASHES)
def test_one_launch_relayouts_of_a_robin_hood_table(oracle, hname, hid):
    """the re-layouts of a non-empty RH table take the one-launch kernels and stay bit-exact: a second batch (k_insert_fused:
    table elements and batch records folded in LDS; 10 % of the batch repeats keys of the table, 3 % repeats itself), an erase
    (k_erase_fused at equal capacity: home bucket from the info byte, erase keys folded in LDS), a doubling rehash and a reserve"""
    n = 380_000          # 200 000 in the table + 200 000 records (180 000 new): predicted and actual capacity are both 2^19
    keys = W.distinct_u64(n, seed=31)
    if hname == "identity":
        keys = W.splitmix64(keys)
    vals = np.arange(n, dtype=np.uint32)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=hname, seed=43)
    o = oracle.OracleTable(0, 128, 0.35, 0.8, hid, 43)
    a = 200_000
    assert g.insert(dev(keys[:a]), dev(vals[:a])) == o.insert(keys[:a], vals[:a])
    k2 = np.concatenate([keys[a:], keys[:15_000], keys[a:a + 5_000]])
    k2 = k2[np.random.default_rng(4).permutation(len(k2))]
    v2 = np.arange(len(k2), dtype=np.uint32) + np.uint32(1_000_000)
    g.profile_enable(True)
    assert g.insert(dev(k2), dev(v2)) == o.insert(k2, v2) == n - a
    assert "k_insert_fused" in g.profile() and "k_dedup" not in g.profile(), g.profile()
    check_state(g, o, 0)
    g.profile_reset()
    assert g.erase(dev(keys[100_000:180_000])) == o.erase(keys[100_000:180_000]) == 80_000
    # (the erase keys are partitioned by chunk and dropped inside the one-launch re-layout: no random-access mark pass)
    assert "k_erase_fused" in g.profile() and "k_erase_mark" not in g.profile() and "k_chunk_place" not in g.profile(), g.profile()
    check_state(g, o, 0)
    g.profile_reset()
    g.rehash(2 * g.capacity()); o.rehash(2 * o.capacity())
    assert "k_rebuild_fused" in g.profile(), g.profile()
    check_state(g, o, 0)
    check_queries(g, o, np.concatenate([keys[:3000], keys[170_000:181_000]]))
    # counting insert into the non-empty table (the oracle has no reducer: checked against numpy): values add up (std::plus)
    g.profile_reset()
    uk, cnt = np.unique(k2, return_counts=True)
    before = dict(zip(*[x.tolist() for x in g.sorted_items()]))
    assert g.insert_reduce_plus(dev(k2)) == 0
    assert "k_insert_fused" in g.profile(), g.profile()
    after = dict(zip(*[x.tolist() for x in g.sorted_items()]))
    for k, c in zip(uk[:5000].tolist(), cnt[:5000].tolist()):
        if k in before:
            assert after[k] == (before[k] + c) & 0xFFFFFFFF
    assert np.array_equal(g.export_info(), o.export_info()) and g.size() == o.size()        # same key set, same layout
    g.close()


def test_one_launch_relayouts_of_a_linear_probe_table(oracle):
    """the LP variants: second batch through k_insert_fused (tombstones of an earlier erase are dropped, the home comes from
    the hash), doubling rehash through k_rebuild_fused; results (not layout) against the oracle"""
    n = 380_000
    keys = W.distinct_u64(n, seed=37)
    vals = np.arange(n, dtype=np.uint32)
    g = kh.hashmap_linearprobe_doubling(128, 0.35, 0.8); o = oracle.OracleTable(1, 128, 0.35, 0.8)
    a = 200_000
    assert g.insert(dev(keys[:a]), dev(vals[:a])) == o.insert(keys[:a], vals[:a])
    assert g.erase(dev(keys[:30_000])) == o.erase(keys[:30_000])              # tombstones
    k2 = np.concatenate([keys[a:], keys[20_000:40_000]])                         # new keys, erased keys coming back, keys of the table
    k2 = k2[np.random.default_rng(5).permutation(len(k2))]
    v2 = np.arange(len(k2), dtype=np.uint32) + np.uint32(5_000_000)
    g.profile_enable(True)
    assert g.insert(dev(k2), dev(v2)) == o.insert(k2, v2)
    assert "k_insert_fused" in g.profile() and "k_dedup" not in g.profile(), g.profile()
    check_state(g, o, 1)
    assert int((g.export_info() == 0x80).sum()) == 0                             # the re-layout dropped the tombstones
    g.profile_reset()
    g.rehash(2 * g.capacity()); o.rehash(2 * o.capacity())
    assert "k_rebuild_fused" in g.profile(), g.profile()
    check_state(g, o, 1)
    check_queries(g, o, np.concatenate([keys[:2000], keys[25_000:35_000], keys[-2000:]]))
    g.close()


def test_general_path_when_fused_build_is_disabled():
    """the same bulk builds through the general path (k_dedup / k_chunk_count / k_chunk_carry / k_chunk_place):
    a subset of this file re-run in a child process with KH_DISABLE_FUSED_BUILD=1"""
    import os
    import subprocess
    import sys
    if os.environ.get("KH_DISABLE_FUSED_BUILD"):
        pytest.skip("already inside the child run")
    env = dict(os.environ, KH_DISABLE_FUSED_BUILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_incremental_batches_host_arrays or test_capacity_rule_edges or test_pairs_layout or test_sentinel"],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # and a distinct-key bulk build that would otherwise take the fused launch
    code = ("import numpy as np, kmerhash_amd as kh\n"
            "from kmerhash_amd import workloads as W\n"
            "from oracle import oracle_py as O\n"
            "k = W.distinct_u64(300000, seed=77); v = np.arange(len(k), dtype=np.uint32)\n"
            "g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8); g.profile_enable(True)\n"
            "o = O.OracleTable(0, 128, 0.35, 0.8)\n"
            "assert g.insert(k, v) == o.insert(k, v)\n"
            "p = g.profile(); assert 'k_dedup' in p and 'k_build_fused' not in p, p\n"
            "assert np.array_equal(g.export_info(), o.export_info())\n"
            "a, b = g.sorted_items(), o.sorted_items(); assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])\n"
            "print('general path ok')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "general path ok" in r.stdout, r.stdout + r.stderr


def test_c_abi_argument_errors_and_dump_replay(tmp_path):
    """bad arguments come back as status codes (no exception crosses the ABI); the C++ driver replays a dump (-F)"""
    import ctypes as C
    import os
    import subprocess
    from kmerhash_amd import _capi as K, io_utils as IO
    L = K.lib()
    h = C.c_void_p()
    assert L.kh_create(C.byref(h), 0, 8, 4, 1, 43, 128, 0.35, 0.8, 0) == K.KH_OK
    n = C.c_uint64()
    assert L.kh_insert(h, None, None, 5, K.KH_MEM_HOST, C.byref(n)) == K.KH_ERR_INVALID
    assert b"null" in L.kh_last_error(h)
    assert L.kh_insert(h, None, None, 0, K.KH_MEM_HOST, C.byref(n)) == K.KH_OK and n.value == 0
    assert L.kh_count(h, None, 3, K.KH_MEM_HOST, None) == K.KH_ERR_INVALID
    assert L.kh_find_compact(h, None, 3, K.KH_MEM_HOST, None, None, C.byref(n)) == K.KH_ERR_INVALID
    assert L.kh_size(None, C.byref(n)) == K.KH_ERR_INVALID
    assert L.kh_create(C.byref(C.c_void_p()), 0, 8, 4, 1, 43, 128, 0.35, 0.8, 99) == K.KH_ERR_INVALID      # no such device
    assert L.kh_create(C.byref(C.c_void_p()), 5, 8, 4, 1, 43, 128, 0.35, 0.8, 0) == K.KH_ERR_INVALID       # no such kind
    assert L.kh_destroy(h) == K.KH_OK
    assert L.kh_release_cached_memory(0) == K.KH_OK
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binp = os.path.join(root, "benchmark", "_benchmark_hashtables")
    if not os.path.exists(binp):
        pytest.skip("driver not built (test_cpp_shim builds it)")
    keys, vals = W.w1_benchmark_hashtables(100_000, seed=5)
    p = str(tmp_path / "dump.bin")
    IO.serialize_pairs(keys, vals, p)
    r = subprocess.run([binp, "-m", "robinhood", "-F", p, "-Q", "10"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "N=100000 distinct=%d" % len(np.unique(keys)) in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_extreme_multiplicity_and_dense_chunks(oracle, kname, cls, kind):
    """k-mer counting meets keys that occur millions of times (poly-A ...): all copies of a key are one key of ONE
    partition, which is streamed through the LDS staging area tile by tile; a chunk denser than the staging area
    (identity hash, consecutive homes) forces the key-class rounds"""
    rng = np.random.default_rng(9)
    heavy = np.array([0x1111, 0xFFFFFFFFFFFFFFFF, 0xABCDEF0123456789], dtype=np.uint64)
    keys = np.concatenate([np.repeat(heavy, [1_500_000, 400_000, 300_000]), W.distinct_u64(200_000, seed=4)])
    keys = keys[rng.permutation(len(keys))]
    vals = np.arange(len(keys), dtype=np.uint32)
    g = cls(128, 0.35, 0.8)
    o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals) == 200_003
    check_state(g, o, kind)
    g2 = cls(128, 0.35, 0.8)
    g2.insert_reduce_plus(dev(keys))
    uk, cnt = np.unique(keys, return_counts=True)
    sk, sv = g2.sorted_items()
    assert np.array_equal(sk, uk) and np.array_equal(sv, cnt.astype(np.uint32))
    g.close(); g2.close()
    # one chunk holding 1800 consecutive homes (identity hash) + a sparse rest: > 1536 distinct keys in one partition
    dense = np.uint64(5 * 2048) + np.arange(1800, dtype=np.uint64)
    sparse = (np.arange(20_000, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 40)
    k2 = np.concatenate([dense, sparse, dense[:500]])
    k2 = k2[rng.permutation(len(k2))]
    v2 = np.arange(len(k2), dtype=np.uint32)
    g = cls(128, 0.35, 0.8, hash="identity")
    o = oracle.OracleTable(kind, 128, 0.35, 0.8, 0, 43)
    if o.insert(k2, v2) and not o.probe_overflow():
        assert g.insert(dev(k2), dev(v2)) == o.size()
        check_state(g, o, kind)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_high_load_runs_longer_than_a_chunk(oracle, kname, cls, kind):
    """found by tests/soak_fuzz.py: at load 0.94 a 2^22-bucket table holds runs of occupied slots far longer than one
    2048-slot chunk (probe distances stay small); re-laying out such a table (second insert, erase, rehash) must not depend
    on finding an empty slot within a chunk's length"""
    n = 3_940_000
    keys = W.distinct_u64(n, seed=19)
    vals = np.arange(n, dtype=np.uint32)
    g = cls(1 << 15, 0.35, 0.95, hash="murmur", seed=43)
    o = oracle.OracleTable(kind, 1 << 15, 0.35, 0.95, 2, 43)
    a = 3_000_000
    assert g.insert(dev(keys[:a]), dev(vals[:a])) == o.insert(keys[:a], vals[:a])
    assert g.insert(dev(keys[a:]), dev(vals[a:])) == o.insert(keys[a:], vals[a:])          # into a non-empty table at load 0.72 -> 0.94
    if kind == 0 and o.probe_overflow():
        pytest.skip("the reference itself overflows its 7-bit distance here")
    info = o.export_info()
    occ = (info >= 0x80) if kind == 0 else (info < 0x40)
    runs = np.diff(np.flatnonzero(np.concatenate([[True], ~occ, [True]])))
    assert runs.max() > 2048, runs.max()                                                   # the situation under test
    check_state(g, o, kind)
    extra = W.distinct_u64(5, seed=23)
    assert g.insert(dev(extra), dev(vals[:5])) == o.insert(extra, vals[:5])                # one more re-layout of the dense table
    check_state(g, o, kind)
    assert g.erase(dev(keys[:100_000])) == o.erase(keys[:100_000])
    check_state(g, o, kind)
    check_queries(g, o, np.concatenate([keys[99_000:101_000], extra]))
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_update_has_no_trailing_reserve(oracle, kname, cls, kind):
    """found by tests/soak_fuzz.py: insert(Iter,Iter) ends with reserve(size) (hashmap_robinhood.hpp:672), which grows a table
    whose max load factor was lowered below its load; update(k,v) (:1274) does not, so neither does a batch of updates --
    not even an empty one"""
    keys = W.distinct_u64(20_000, seed=5)
    vals = np.arange(len(keys), dtype=np.uint32)
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    g.set_max_load_factor(0.5); o.set_max_load_factor(0.5)          # size 20000 > max_load(32768 * 0.5)
    check_state(g, o, kind)
    g.update(keys[:0], vals[:0])
    check_state(g, o, kind)                                           # still capacity 32768
    for k, v in zip(keys[:3].tolist(), (vals[:3] + 7).tolist()):     # existing keys: each update(k,v) is an insert call that doubles once
        g.update(np.array([k], dtype=np.uint64), np.array([v], dtype=np.uint32)); o.update_one(k, v)
        check_state(g, o, kind)
    assert g.insert(dev(keys[:0]), dev(vals[:0])) == o.insert(keys[:0], vals[:0]) == 0     # the empty insert does end with reserve(size)
    check_state(g, o, kind)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_fused_build_long_probe_distances(oracle, kname, cls, kind):
    """bulk build with home buckets shared by 20..100 keys (identity hash, keys equal modulo the capacity): probe distances
    far beyond the 5-bit code the fused kernel keeps next to each staged record (>= 31 are re-derived from the key)"""
    cap = 1 << 16
    base = np.arange(30_000, dtype=np.uint64) * np.uint64(2)              # spread: every second bucket
    piles = []
    for b, cnt in ((100, 100), (2047, 60), (2048, 45), (40_000, 33), (65_535, 20), (12_345, 31), (12_400, 32)):
        piles.append(np.uint64(b) + (np.arange(1, cnt + 1, dtype=np.uint64) << np.uint64(32)))
    keys = np.concatenate([base] + piles)
    keys = keys[np.random.default_rng(3).permutation(len(keys))]
    vals = np.arange(len(keys), dtype=np.uint32)
    g = cls(cap, 0.35, 0.8, hash="identity")
    o = oracle.OracleTable(kind, cap, 0.35, 0.8, 0, 43)
    n_new = o.insert(keys, vals)
    assert not o.probe_overflow()
    assert g.insert(dev(keys), dev(vals)) == n_new == len(keys)
    assert g.capacity() == o.capacity() == cap
    check_state(g, o, kind)
    if kind == 0:
        assert int(np.flatnonzero(g.displacement_histogram())[-1]) >= 100
    check_queries(g, o, keys[:2000])
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("cap0", [0, 1, 2, 3, 5, 64, 1000, 5000])
def test_tiny_and_odd_initial_capacities(oracle, kname, cls, kind, cap0):
    """ctor capacity is rounded up to a power of two (next_power_of_2; 0 and 1 give 1); tables of 1, 2, 4 ... buckets
    double on the first calls (max_load = size_t(float(buckets) * 0.8f) is 0 for one bucket)"""
    keys = W.distinct_u64(40, seed=cap0 + 1)
    vals = np.arange(40, dtype=np.uint32)
    g = cls(cap0, 0.35, 0.8)
    o = oracle.OracleTable(kind, cap0, 0.35, 0.8)
    assert g.capacity() == o.capacity()
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 7), (7, 40)):
        assert g.insert(keys[a:b], vals[a:b]) == o.insert(keys[a:b], vals[a:b])
        check_state(g, o, kind)
    check_queries(g, o, np.concatenate([keys[:10], W.distinct_u64(10, seed=99)]))
    assert g.erase(keys[:35]) == o.erase(keys[:35])
    check_state(g, o, kind)
    for k in keys[35:]:
        assert g.erase_one(int(k)) == o.erase_one(int(k))
        assert g.capacity() == o.capacity()
    check_state(g, o, kind)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_counting_insert_with_key_only_records(kname, cls, kind, monkeypatch):
    """Reducer = std::plus with the implicit value 1 on a batch the sample finds heavy in duplicates (the k-mer counter's batches):
    the partition records are the keys alone (8 bytes).  Counts equal numpy's, and the table is bit-identical to the one the
    16-byte-record path builds (KH_DISABLE_REC8) and to the one behind exact partition offsets (KH_DISABLE_DUP_SLACK), for a batch into an
    empty table and a second one into the populated table."""
    rng = np.random.default_rng(11)
    uni = W.distinct_u64(1_200_000, seed=21)
    b1 = uni[rng.integers(0, 800_000, 5_000_000)]
    b2 = uni[rng.integers(400_000, 1_200_000, 5_000_000)]
    states = []
    for variant in ("", "KH_DISABLE_REC8", "KH_DISABLE_DUP_SLACK"):
        if variant:
            monkeypatch.setenv(variant, "1")
        g = cls(128, 0.35, 0.8)
        g.profile_enable(True)
        g.insert_reduce_plus(dev(b1))
        # sampled as duplicate-heavy: the general path -- behind a histogram-free partition whose slots are sized for the duplicates
        # (E[m^2] / E[m] from the sample), or behind exact offsets (histogram sweep) when that is switched off
        assert "k_dedup" in g.profile() and ("k_part_hist" in g.profile()) == (variant == "KH_DISABLE_DUP_SLACK"), g.profile()
        g.insert_reduce_plus(dev(b2))
        sk, sv = g.sorted_items()
        uk, cnt = np.unique(np.concatenate([b1, b2]), return_counts=True)
        assert np.array_equal(sk, uk) and np.array_equal(sv, cnt.astype(np.uint32))
        states.append((g.capacity(), g.export_info().copy()))
        g.close()
        if variant:
            monkeypatch.delenv(variant)
    for st in states[1:]:
        assert states[0][0] == st[0] and np.array_equal(states[0][1], st[1])


def test_shard_plan_equals_per_piece_permute():
    """kh_shard_plan: one count sweep for a batch cut into pieces; every piece's permutation and counts equal kh_shard_permute run on
    that piece alone (stable, same destination ranks)"""
    import ctypes as C
    from kmerhash_amd import _capi as K
    L = K.lib()
    for n, p, pieces in ((1_000_003, 8, 4), (70_000, 3, 5), (4096, 2, 3), (5, 4, 2)):
        keys = W.splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(n))
        keys[::7] = keys[0]                                   # duplicates
        vals = np.arange(n, dtype=np.uint32)
        dk, dv = dev(keys), dev(vals)
        plan = C.c_void_p()
        counts = (C.c_uint64 * (p * pieces))(); bounds = (C.c_uint64 * (pieces + 1))()
        assert L.kh_shard_plan_create(C.byref(plan), 1, 9876543, 0, 0, p, dk.data_ptr(), n, pieces, counts, bounds, 0, None) == K.KH_OK
        b = [int(x) for x in bounds]
        assert b[0] == 0 and b[-1] == n and all(x % 4096 == 0 for x in b[:-1]) and b == sorted(b)
        for i in range(pieces):
            m = b[i + 1] - b[i]
            ok = torch.empty(max(m, 1), dtype=torch.int64, device="cuda"); ov = torch.empty(max(m, 1), dtype=torch.int32, device="cuda")
            assert L.kh_shard_plan_permute(plan, i, dk.data_ptr(), dv.data_ptr(), ok.data_ptr(), ov.data_ptr(), None) == K.KH_OK
            rk = torch.empty(max(m, 1), dtype=torch.int64, device="cuda"); rv = torch.empty(max(m, 1), dtype=torch.int32, device="cuda")
            rc = (C.c_uint64 * p)()
            assert L.kh_shard_permute(1, 9876543, p, dk[b[i]:].data_ptr() if m else None, dv[b[i]:].data_ptr() if m else None, m,
                                      rk.data_ptr(), rv.data_ptr(), rc, 0, None) == K.KH_OK
            torch.cuda.synchronize()
            assert [int(counts[i * p + r]) for r in range(p)] == [int(x) for x in rc]
            assert torch.equal(ok[:m], rk[:m]) and torch.equal(ov[:m], rv[:m])
        L.kh_shard_plan_destroy(plan)


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_repeatable_streamed_insert_partitions_without_a_histogram(oracle, kname, cls, kind):
    """kh_insert_begin_ex(KH_INS_REPEATABLE): the caller keeps its pieces, so the pieces of a duplicate-free batch are partitioned
    without a histogram pass into slots they SHARE (one source for the build; 12-byte records).  Hidden duplicates / skew make
    kh_insert_end return KH_ERR_RETRY with the table unchanged; feeding the same pieces again the plain way gives the reference's
    result.  A duplicate-heavy first piece takes the exact layout from the start (no retry)."""
    n = 4_000_000
    keys = W.distinct_u64(n, seed=77); vals = np.arange(n, dtype=np.uint32)
    cuts = [0, 900_000, 2_100_000, 3_000_001, n]

    def feed_all(g, k, v, **kw):
        g.insert_begin(len(k), **kw)
        for a, b in zip(cuts[:-1], cuts[1:]):
            g.insert_feed(dev(k[a:b]), dev(v[a:b]) if v is not None else None)
        return g.insert_end()

    # 1. distinct keys: speculative layout holds
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    g.profile_enable(True)
    assert feed_all(g, keys, vals, repeatable=True) == o.insert(keys, vals)
    p = g.profile()
    assert "k_part_hist" not in p and p["k_part_scatter"][0] == 2 * (len(cuts) - 1) and "k_dedup" not in p, p
    check_state(g, o, kind)
    g.close()
    # 2. duplicates the first piece's sample cannot see (in the LAST piece): KhRetry, nothing inserted, then the plain way
    k2 = keys.copy()
    k2[3_500_000:3_500_300] = keys[10:310]
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    with pytest.raises(kh.KhRetry):
        feed_all(g, k2, vals, repeatable=True)
    assert g.size() == 0
    assert feed_all(g, k2, vals) == o.insert(k2, vals)
    check_state(g, o, kind)
    g.close()
    # 3. one key 40000 times in a later piece: a slot overflows -> KhRetry as well
    k3 = keys.copy()
    k3[2_200_000:2_240_000] = keys[5]
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    with pytest.raises(kh.KhRetry):
        feed_all(g, k3, vals, repeatable=True)
    assert feed_all(g, k3, vals) == o.insert(k3, vals)
    check_state(g, o, kind)
    g.close()
    # 4. duplicate-heavy batch: the first piece's sample sees it, exact layout, no retry
    kd, vd = W.w1_benchmark_hashtables(n, seed=5)
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    g.profile_enable(True)
    assert feed_all(g, kd, vd, repeatable=True) == o.insert(kd, vd)
    assert "k_part_hist" in g.profile()
    check_state(g, o, kind)
    g.close()
    # 5. counting form (Reducer = std::plus, values omitted), repeatable, into an empty table
    g = cls(128, 0.35, 0.8)
    feed_all(g, keys, None, reduce_plus=True, repeatable=True)
    sk, sv = g.sorted_items()
    assert np.array_equal(sk, np.sort(keys)) and (sv == 1).all()
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("pieces", [1, 2, 5, 16])
def test_streamed_insert_equals_one_insert(oracle, kname, cls, kind, pieces):
    """kh_insert_begin/feed/end: the pieces of one batch, each partitioned on arrival, count as ONE insert(Iter,Iter) in
    feed order (first value wins across pieces; the capacity rule sees one call sequence)"""
    rng = np.random.default_rng(pieces)
    for keys, vals in (W.w1_benchmark_hashtables(150_000, seed=31),                      # duplicates across pieces
                       (W.distinct_u64(200_000, seed=6), np.arange(200_000, dtype=np.uint32))):   # fused build at the end
        cuts = np.sort(np.concatenate([[0, len(keys)], rng.integers(0, len(keys), pieces - 1)]))
        g = cls(128, 0.35, 0.8)
        o = oracle.OracleTable(kind, 128, 0.35, 0.8)
        g.insert_begin(len(keys))
        for a, b in zip(cuts[:-1], cuts[1:]):
            g.insert_feed(dev(keys[a:b]), dev(vals[a:b])) if (a + b) % 2 else g.insert_feed(keys[a:b], vals[a:b])   # device and host pieces
        assert g.insert_end() == o.insert(keys, vals)
        check_state(g, o, kind)
        # second streamed batch into the now non-empty table, reducer form, values omitted
        k2 = keys[::3]
        g.insert_begin(len(k2), reduce_plus=True)
        h = len(k2) // 2
        g.insert_feed(dev(k2[:h])); g.insert_feed(dev(k2[h:]))
        g.insert_end()
        sk, sv = g.sorted_items()
        ok, ov = o.sorted_items()
        uk, cnt = np.unique(k2, return_counts=True)
        exp = ov.astype(np.uint64)
        exp[np.searchsorted(ok, uk)] += cnt.astype(np.uint64)
        assert np.array_equal(sk, ok) and np.array_equal(sv, (exp & 0xFFFFFFFF).astype(np.uint32))
        g.close()


def test_streamed_insert_misuse_is_refused():
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    k = W.distinct_u64(100, seed=1)
    v = np.arange(100, dtype=np.uint32)
    with pytest.raises(kh.KhError):
        g.insert_feed(k, v)                       # no begin
    g.insert_begin(100)
    with pytest.raises(kh.KhError):
        g.insert(k, v)                            # other mutation while streaming
    with pytest.raises(kh.KhError):
        g.insert_feed(np.concatenate([k, k]), np.concatenate([v, v]))     # more than announced
    g.insert_feed(k[:40], v[:40])
    with pytest.raises(kh.KhError):
        g.insert_end()                            # fewer than announced
    assert g.size() == 0
    g.insert_begin(100); g.insert_feed(k, v)
    assert g.insert_end() == 100 and g.count(k).all()
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("form", ["update_one", "insert_one", "batch"])
def test_more_than_one_doubling_pending(oracle, kname, cls, kind, form):
    """after set_max_load_factor() lowered the threshold so far that size > max_load(2 * capacity), ONE insert call
    re-doubles inside the Robin Hood rehash (copy() re-inserts through insert(), hashmap_robinhood.hpp:432-464,530): 20000
    keys at capacity 32768 with max load 0.25 end at 131072, not 65536.  The LP rehash copies without the check: one doubling
    per call.  The single-key insert(value_type) has no trailing reserve(size()), the batch forms do."""
    keys = W.distinct_u64(20_001, seed=5)
    vals = np.arange(len(keys), dtype=np.uint32)
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    assert g.insert(dev(keys[:20_000]), dev(vals[:20_000])) == o.insert(keys[:20_000], vals[:20_000])
    assert g.capacity() == 32768
    g.set_max_load_factor(0.25); o.set_max_load_factor(0.25)         # max_load(65536) = 16384 < 20000
    check_state(g, o, kind)
    new_k, new_v = int(keys[20_000]), int(vals[20_000])
    if form == "update_one":
        g.update(np.array([keys[5]], dtype=np.uint64), np.array([77], dtype=np.uint32)); o.update_one(int(keys[5]), 77)
    elif form == "insert_one":
        assert g.insert_one(new_k, new_v) == o.insert_one(new_k, new_v) == 1
    else:
        b = np.concatenate([keys[:3], keys[20_000:]]); bv = np.concatenate([vals[:3] + 9, vals[20_000:]])
        assert g.insert(dev(b), dev(bv)) == o.insert(b, bv) == 1
    if kind == 0:
        assert o.capacity() == 131072
    check_state(g, o, kind)
    # a second single-key insert of an existing key: an insert call like any other
    assert g.insert_one(int(keys[7]), 1) == o.insert_one(int(keys[7]), 1) == 0
    check_state(g, o, kind)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_single_key_insert_has_no_trailing_reserve(oracle, kname, cls, kind):
    """size == max_load(2 * capacity) exactly: insert(value_type) doubles once and leaves size = max_load + 1 in place; the batch
    form of the same key ends with reserve(size()) and doubles again (hashmap_robinhood.hpp:522-624 vs :633-673)"""
    for batch in (False, True):
        g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
        keys = W.distinct_u64(16_385, seed=11)
        vals = np.arange(len(keys), dtype=np.uint32)
        assert g.insert(keys[:16_384], vals[:16_384]) == o.insert(keys[:16_384], vals[:16_384])
        assert g.capacity() == 32768
        g.set_max_load_factor(0.25); o.set_max_load_factor(0.25)     # max_load(65536) = 16384 == size
        k, v = keys[16_384:], vals[16_384:]
        if batch:
            assert g.insert(k, v) == o.insert(k, v) == 1
        else:
            assert g.insert_one(int(k[0]), int(v[0])) == o.insert_one(int(k[0]), int(v[0])) == 1
        check_state(g, o, kind)
        g.close()


def test_calls_inside_a_streamed_insert_are_refused():
    """every entry point that resets the workspace arena refuses to run between kh_insert_begin and kh_insert_end
    (the streamed insert keeps its partition buffers there)"""
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    k = W.distinct_u64(5000, seed=1)
    v = np.arange(5000, dtype=np.uint32)
    g.insert(k[:1000], v[:1000])
    g.insert_begin(4000)
    g.insert_feed(dev(k[1000:3000]), dev(v[1000:3000]))
    for call in (lambda: g.find(k[:10]), lambda: g.find_values(k[:10]), lambda: g.count(k[:10]), g.to_vector, g.clear,
                 g.displacement_histogram, lambda: g.erase(k[:10]), lambda: g.rehash(1 << 16), lambda: g.update(k[:1], v[:1])):
        with pytest.raises(kh.KhError):
            call()
    g.insert_feed(k[3000:], v[3000:])
    assert g.insert_end() == 4000 and g.size() == 5000
    kk, vv = g.sorted_items()
    o = np.argsort(k)
    assert np.array_equal(kk, k[o]) and np.array_equal(vv, v[o])
    g.close()


@pytest.mark.parametrize("hname,hid", [("murmur3avx64", 1), ("farm", 3)])
def test_mid_size_batches_are_applied_in_place(oracle, hname, hid):
    """10^2..10^6 keys into a large Robin Hood table cost O(batch): the table is cut into regions owned by one
    lane each, keys whose displacement chain / backward shift would leave the region are deferred to a pass with shifted
    regions, the rest to a single lane (hashmap_robinhood.hpp:522-624,1294-1356).  Bit-exact info array after every step."""
    cap = 1 << 22                                              # 8192 regions of 512 slots; in place for 17..8192 keys
    base = W.distinct_u64(2_400_000, seed=77)
    bv = np.arange(len(base), dtype=np.uint32)
    g = kh.hashmap_robinhood_doubling(cap, 0.35, 0.8, hash=hname, seed=43)
    o = oracle.OracleTable(0, cap, 0.35, 0.8, hid, 43)
    assert g.insert(dev(base), dev(bv)) == o.insert(base, bv)
    fresh = W.distinct_u64(40_000, seed=78)
    pos = 0
    rng = np.random.default_rng(5)
    g.profile_enable(True)
    for n in (17, 100, 1000, 8000, 8192):
        new = fresh[pos:pos + n]; pos += n
        # new keys, duplicates of them, and keys the table already holds, shuffled
        k = np.concatenate([new, new[: n // 3], base[rng.integers(0, len(base), n // 4)]])[:n]
        k = k[rng.permutation(len(k))]
        v = rng.integers(0, 1 << 32, len(k), dtype=np.uint64).astype(np.uint32)
        assert g.insert(dev(k), dev(v)) == o.insert(k, v)
        check_state(g, o, 0)
    assert g.capacity() == cap
    prof = g.profile()
    assert prof["k_ip_apply"][0] == 5 and "k_rebuild_fused" not in prof and "k_insert_fused" not in prof, prof
    # update: existing keys take the last value, new ones are inserted
    k = np.concatenate([fresh[pos:pos + 3000], base[:2000], fresh[pos:pos + 500]]); pos += 3000
    v = np.arange(len(k), dtype=np.uint32) + np.uint32(9)
    g.update(dev(k), dev(v))
    for kk, vv in zip(k.tolist(), v.tolist()):
        o.update_one(kk, vv)
    check_state(g, o, 0)
    # erase in place: hits, misses and repeated keys
    for n in (50, 5000, 8192):
        e = np.concatenate([base[rng.integers(0, len(base), n - n // 5)], W.distinct_u64(n // 5, seed=1000 + n)])
        assert g.erase(dev(e)) == o.erase(e)
        check_state(g, o, 0)
    assert g.profile()["k_ip_apply"][0] == 9
    check_queries(g, o, np.concatenate([base[:3000], fresh[:3000], W.distinct_u64(1000, seed=4)]))
    # counting insert (std::plus) in place
    k = np.concatenate([fresh[pos:pos + 2000], fresh[pos:pos + 2000], base[:1000]])
    g.insert_reduce_plus(dev(k))
    sk, sv = g.sorted_items(); ok, ov = o.sorted_items()
    uk, cnt = np.unique(k, return_counts=True)
    exp = dict(zip(ok.tolist(), ov.tolist()))
    for a, c in zip(uk.tolist(), cnt.tolist()):
        exp[a] = (exp.get(a, 0) + c) & 0xFFFFFFFF
    assert np.array_equal(sk, np.array(sorted(exp), dtype=np.uint64))
    assert np.array_equal(sv, np.array([exp[a] for a in sorted(exp)], dtype=np.uint32))
    g.close()


def test_mid_size_in_place_region_boundaries(oracle):
    """identity hash, crafted homes: chains that cross a region boundary (pass 2), chains that cross regular AND shifted
    boundaries (a run of 1100 occupied slots: pass 3, the single lane), and more keys in one region than a bin holds"""
    cap = 1 << 16                                              # 128 regions of 512 slots; in place for 17..128 keys
    run = np.arange(1000, 2100, dtype=np.uint64)               # every slot of [1000, 2100) holds an element at its home
    sparse = np.arange(4096, 60_000, 7, dtype=np.uint64)
    keys = np.concatenate([run, sparse])
    vals = np.arange(len(keys), dtype=np.uint32)
    g = kh.hashmap_robinhood_doubling(cap, 0.35, 0.8, hash="identity")
    o = oracle.OracleTable(0, cap, 0.35, 0.8, 0, 43)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    hi = np.uint64(1) << np.uint64(32)
    new = np.concatenate([
        np.uint64(1010) + hi * np.arange(1, 4, dtype=np.uint64),            # home inside the long run: chain ends behind 2100
        np.uint64(2040) + hi * np.arange(1, 6, dtype=np.uint64),            # home just before a regular boundary
        np.uint64(1020) + hi * np.arange(5, 8, dtype=np.uint64),            # just before a shifted boundary
        np.uint64(30_000) + np.uint64(7) * np.arange(40, dtype=np.uint64) + hi,   # 40 keys in one region (bin capacity 16)
        np.uint64(65_530) + hi * np.arange(1, 9, dtype=np.uint64),          # runs over the end of the table into slot 0
    ])
    nv = np.arange(len(new), dtype=np.uint32) + np.uint32(1000)
    g.profile_enable(True)
    assert g.insert(dev(new), dev(nv)) == o.insert(new, nv) == len(new)
    assert "k_ip_serial" in g.profile() and g.capacity() == cap
    check_state(g, o, 0)
    check_queries(g, o, np.concatenate([new, run[:50], np.uint64(1010) + hi * np.arange(20, 25, dtype=np.uint64)]))
    e = np.concatenate([run[5:40], new[:8], new[-8:], np.uint64(30_000) + np.uint64(7) * np.arange(40, dtype=np.uint64) + hi])
    assert g.erase(dev(e)) == o.erase(e)
    check_state(g, o, 0)
    g.close()


def _revcomp(x, k):
    r = np.zeros_like(x)
    y = x.copy()
    for _ in range(k):
        r = (r << np.uint64(2)) | (np.uint64(3) - (y & np.uint64(3)))
        y >>= np.uint64(2)
    return r


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("k", [31, 21, 32])
def test_bimolecule_tables_key_transform(oracle, kname, cls, kind, k):
    """fsc::TransformedHash<Kmer, Hash, lex_less> + TransformedComparator (hash_new.hpp:387-1134) as the reference's
    tests instantiate the maps in bimolecule mode (test_hashmap_robinhood_doubling.cpp:560-626): a k-mer and its reverse complement
    are ONE key, hashed and compared as min(k-mer, reverse complement); what is stored -- and what find / to_vector return -- are the
    bits of the first occurrence.  Every path: bulk build, insert into a non-empty table, general path, small and mid-size batches in
    place, update, reducer, erase, queries."""
    rng = np.random.default_rng(k)
    top = np.uint64((1 << (2 * k)) - 1) if k < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    fw = W.distinct_u64(60_000, seed=k) & top
    rc = _revcomp(fw, k)
    g = cls(128, 0.35, 0.8, hash="farm", seed=43); o = oracle.OracleTable(kind, 128, 0.35, 0.8, 3, 43)
    g.set_key_transform(k); o.set_key_transform(k)
    # both strands of many k-mers in one batch, either strand first; duplicates
    keys = np.concatenate([fw[:30_000], rc[10_000:40_000], fw[20_000:50_000]])
    keys = keys[rng.permutation(len(keys))]
    vals = np.arange(len(keys), dtype=np.uint32)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    check_state(g, o, kind)                              # sorted (stored key, value) sets equal: the FIRST occurrence's bits are stored
    assert g.size() < len(np.unique(keys))               # strands were merged
    q = np.concatenate([fw[:2000], rc[:2000], rc[45_000:47_000], fw[55_000:56_000], rc[55_000:56_000]])
    check_queries(g, o, q)                               # find returns the stored pair, whichever strand is asked for
    # second batch into the non-empty table (fused insert), then the general path, then in place
    k2 = np.concatenate([rc[40_000:60_000], fw[:5000]]); v2 = np.arange(len(k2), dtype=np.uint32) + np.uint32(7)
    assert g.insert(dev(k2), dev(v2)) == o.insert(k2, v2)
    check_state(g, o, kind)
    extra = W.distinct_u64(3000, seed=1000 + k) & top
    for batch in (np.concatenate([extra[:5], _revcomp(extra[:5], k)]), np.concatenate([extra[5:300], _revcomp(extra[100:300], k)])):
        bv = np.arange(len(batch), dtype=np.uint32) + np.uint32(99)
        assert g.insert(dev(batch), dev(bv)) == o.insert(batch, bv)
        check_state(g, o, kind)
    u = np.concatenate([rc[:300], extra[300:600]]); uv = np.arange(len(u), dtype=np.uint32) + np.uint32(5000)
    g.update(dev(u), dev(uv))
    for kk, vv in zip(u.tolist(), uv.tolist()):
        o.update_one(kk, vv)
    check_state(g, o, kind)
    e = np.concatenate([rc[100:4000], fw[3000:9000], extra[:10]])
    assert g.erase(dev(e)) == o.erase(e)
    check_state(g, o, kind)
    check_queries(g, o, q)
    for x in (int(fw[20_000]), int(rc[20_001])):
        assert g.erase_one(x) == o.erase_one(x)
    check_state(g, o, kind)
    with pytest.raises(kh.KhError):
        g.set_key_transform(0)                           # not on a non-empty table
    # batched TransformedHash::operator()
    h = kh.hash_batch(q, "farm", 43, lex_less_k=k)
    assert np.array_equal(h, kh.hash_batch(np.minimum(q, _revcomp(q, k)), "farm", 43))
    g.close()


def test_fused_build_look_back_time_out_falls_back_to_the_general_path():
    """k_build_fused waits (bounded) for its predecessor workgroup's published run-over; a time-out raises a flag and the
    host redoes the batch on the general path.  KH_DEBUG_POLL_LIMIT=0 makes every look-back that does not find the word at its first
    read give up: the results must still be the oracle's, through k_dedup / k_chunk_place."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, kmerhash_amd as kh\n"
            "from kmerhash_amd import workloads as W\n"
            "from oracle import oracle_py as O\n"
            "k = W.distinct_u64(3000000, seed=78); v = np.arange(len(k), dtype=np.uint32)\n"
            "g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8); g.profile_enable(True)\n"
            "o = O.OracleTable(0, 128, 0.35, 0.8)\n"
            "assert g.insert(k, v) == o.insert(k, v)\n"
            "p = g.profile(); assert 'k_build_fused' in p and 'k_dedup' in p and 'k_chunk_place' in p, p\n"
            "assert np.array_equal(g.export_info(), o.export_info()) and g.capacity() == o.capacity()\n"
            "a, b = g.sorted_items(), o.sorted_items(); assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])\n"
            "k2 = W.distinct_u64(1000000, seed=79); v2 = np.arange(len(k2), dtype=np.uint32)\n"
            "assert g.insert(k2, v2) == o.insert(k2, v2)          # fused insert into the non-empty table: same time-out, same fall-back\n"
            "assert g.erase(k[:500000]) == o.erase(k[:500000])    # one-launch re-layout: likewise\n"
            "assert np.array_equal(g.export_info(), o.export_info())\n"
            "a, b = g.sorted_items(), o.sorted_items(); assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])\n"
            "print('time-out path ok')\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KH_DEBUG_POLL_LIMIT="0"), capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "time-out path ok" in r.stdout, r.stdout + r.stderr


def test_histogram_free_partition_and_its_fall_backs(oracle):
    """a large batch of (nearly) distinct hashed keys is partitioned without a histogram pass (fixed slots of mean + 7
    sigma per partition).  A sample decides how wide the slots are: a duplicate-heavy batch fills its partitions key by key, E[m^2] / E[m]
    records at a time, and gets slots of mean + 7 sigma of THAT distribution (factor estimated from the sample); duplicates the sample misses
    (one key repeated 30000 times among 4e6 distinct ones) overflow a slot and the batch is redone with exact offsets."""
    n = 4_000_000                                   # capacity 2^23: 4096 partitions (two partition passes) of ~977 records
    keys = W.distinct_u64(n, seed=123); vals = np.arange(n, dtype=np.uint32)
    for variant in ("distinct", "few_hidden_duplicates", "duplicate_heavy", "hidden_skew"):
        if variant == "few_hidden_duplicates":      # 300 keys given twice, none at a sampled position: the speculative no-fold build must notice
            k = keys.copy(); v = vals
            k[(n // 65536) * np.arange(1000, 1300) + 3] = keys[(n // 65536) * np.arange(5000, 5300) + 7]
        elif variant == "duplicate_heavy":
            k, v = W.w1_benchmark_hashtables(n, seed=9)
        elif variant == "hidden_skew":
            k = keys.copy(); k[(n // 65536) * np.arange(30_000) + 1] = keys[7]; v = vals      # between the sample's positions (stride n // 65536)
        else:
            k, v = keys, vals
        g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8); o = oracle.OracleTable(0, 128, 0.35, 0.8)
        g.profile_enable(True)
        assert g.insert(dev(k), dev(v)) == o.insert(k, v)
        p = g.profile()
        assert "k_sample_dups" in p
        if variant == "distinct":
            assert "k_part_hist" not in p and p["k_part_scatter"][0] == 2 and "k_dedup" not in p, p
        elif variant == "few_hidden_duplicates":
            # no-fold build on 12-byte records (no stream positions) notices, the batch is partitioned again with 16-byte records and exact
            # offsets, and the one-launch build WITH the fold takes it (300 duplicates do not change the capacity)
            assert "k_part_hist" in p and p["k_part_scatter"][0] == 4 and p["k_build_fused"][0] == 2 and "k_dedup" not in p, p
        elif variant == "duplicate_heavy":          # histogram-free as well, with slots sized for the duplicates (variance factor from the sample)
            assert "k_part_hist" not in p and p["k_part_scatter"][0] == 2 and "k_dedup" in p, p
        else:
            assert "k_part_hist" in p and p["k_part_scatter"][0] == 4, p       # histogram-free attempt, then exact
        check_state(g, o, 0)
        g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
def test_reducer_plus_speculation_never_touches_live_counts(kname, cls, kind):
    """a Reducer = std::plus insert into a NON-EMPTY table whose histogram-free partition overflows a slot (hidden
    skew) is repeated with exact offsets -- the counts of the keys the table already holds must be added ONCE.  k_dedup only lists
    those sums; k_apply_plus adds them after the overflow flag has been read.  One-shot form, then the repeatable streamed form:
    KhRetry must leave the table exactly as it was (kmerhash_amd.h: 'table unchanged'), and the plain re-feed gives numpy's sums."""
    n = 4_000_000
    base = W.distinct_u64(1000, seed=3)
    fresh = W.distinct_u64(n, seed=4)
    k = fresh.copy()
    # one key 30000 times, at positions neither sample looks at: the one-shot insert samples every (n // 65536) = 61st key, the
    # streamed form every (1_500_000 // 65536) = 22nd key of its first piece
    cand = np.arange(n)
    cand = cand[(cand % 61 != 0) & (cand % 22 != 0)]
    k[cand[::100][:30_000]] = fresh[7]
    k[(n // 65536) * np.arange(40_000, 40_600) + 2] = base[:600]      # keys the table already holds (counts must grow by exactly 1)
    k[(n // 65536) * np.arange(41_000, 41_100) + 2] = base[:100]      # ... some of them twice

    def expected(pre_k, pre_c, batch):
        uk, cnt = np.unique(np.concatenate([np.repeat(pre_k, pre_c.astype(np.int64)), batch]), return_counts=True)
        return uk, cnt.astype(np.uint32)

    # one-shot
    g = cls(128, 0.35, 0.8)
    g.insert_reduce_plus(dev(base))
    g.insert_reduce_plus(dev(base[:10]))                              # counts 2 for ten of them
    pk, pc = g.sorted_items()
    g.profile_enable(True)
    g.insert_reduce_plus(dev(k))
    p = g.profile()
    assert p["k_part_scatter"][0] == 4 and "k_apply_plus" in p, p      # histogram-free attempt overflowed, exact offsets after it
    sk, sv = g.sorted_items()
    uk, cnt = expected(pk, pc, k)
    assert np.array_equal(sk, uk) and np.array_equal(sv, cnt)
    g.close()
    # repeatable streamed form: KhRetry leaves the table unchanged
    g = cls(128, 0.35, 0.8)
    g.insert_reduce_plus(dev(base))
    g.insert_reduce_plus(dev(base[:10]))
    before = g.sorted_items(), g.capacity(), g.export_info().copy()
    cuts = [0, 1_500_000, 2_500_000, n]

    def feed_all(**kw):
        g.insert_begin(n, reduce_plus=True, **kw)
        for a, b in zip(cuts[:-1], cuts[1:]):
            g.insert_feed(dev(k[a:b]))
        return g.insert_end()

    with pytest.raises(kh.KhRetry):
        feed_all(repeatable=True)
    after = g.sorted_items(), g.capacity(), g.export_info()
    assert np.array_equal(before[0][0], after[0][0]) and np.array_equal(before[0][1], after[0][1]) and before[1] == after[1]
    assert np.array_equal(before[2], after[2])
    feed_all()
    sk, sv = g.sorted_items()
    assert np.array_equal(sk, uk) and np.array_equal(sv, cnt)
    # an aborted streamed insert leaves the table unchanged and usable
    g.insert_begin(1000, reduce_plus=True)
    g.insert_feed(dev(base[:500]))
    g.insert_abort()
    sk2, sv2 = g.sorted_items()
    assert np.array_equal(sk2, uk) and np.array_equal(sv2, cnt)
    g.insert_reduce_plus(dev(base[:5]))
    assert g.size() == len(uk)
    g.close()


def _fmix64(k):
    k = k.astype(np.uint64).copy()
    k ^= k >> np.uint64(33); k *= np.uint64(0xff51afd7ed558ccd); k ^= k >> np.uint64(33); k *= np.uint64(0xc4ceb9fe1a85ec53); k ^= k >> np.uint64(33)
    return k


def test_reducer_plus_class_restart_counts_once():
    """the in-kernel half of the same finding: a partition with more distinct keys than one LDS pass holds (1536) is swept in R classes
    (class = fmix64(key + c) >> 32 mod R), and a class that overflows restarts the sweep with 2R classes.  Sums of existing keys listed
    by the abandoned sweep must not be applied.  Crafted: a mid-size batch (in-place path, 4 partitions) whose 3400 keys all fall into
    partition 0, 900 of them in class 0 of R = 2 (fits: its sums were added by the old code) and 2500 in class 1 (overflows ->
    R = 4, everything again).  All keys exist already: every one is an update."""
    n0 = 2_000_000
    uni = W.distinct_u64(n0, seed=31)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    g.insert_reduce_plus(dev(uni))
    assert g.capacity() == 1 << 22
    h = kh.hash_batch(uni[:400_000], "murmur3avx64", 43)
    part0 = ((h >> np.uint64(11)) & np.uint64(3)) == 0                 # PB = 2: partition = bit-reversed low 2 bits of the chunk id; 0 stays 0
    cls1 = ((_fmix64(uni[:400_000] + np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)) % np.uint64(2)) == 1
    c0 = uni[:400_000][part0 & ~cls1][:900]
    c1 = uni[:400_000][part0 & cls1][:2500]
    assert len(c0) == 900 and len(c1) == 2500
    batch = np.concatenate([c0, c1, c0[:300]])                           # (some twice)
    batch = batch[W.shuffle_perm(len(batch), 1)]
    g.profile_enable(True)
    assert g.insert_reduce_plus(dev(batch)) == 0
    p = g.profile()
    assert "k_part_direct" in p and "k_apply_plus" in p, p               # the in-place path of mid-size batches
    sk, sv = g.sorted_items()
    uk, cnt = np.unique(np.concatenate([uni, batch]), return_counts=True)
    assert np.array_equal(sk, uk) and np.array_equal(sv, cnt.astype(np.uint32))
    g.close()


def test_batch_erase_streaming_form_and_its_fall_back(oracle, monkeypatch):
    """a large Robin Hood batch erase partitions its keys by chunk and drops them inside the one-launch re-layout
    (k_erase_fused).  Hits, misses, keys given twice, keys of a bimolecule table by their other strand; an erase batch too dense for
    the staging area (elements + erase keys of a chunk >= 2048) falls back to the mark + re-layout path; both equal the oracle."""
    n = 1_500_000
    keys = W.distinct_u64(n, seed=41); vals = np.arange(n, dtype=np.uint32)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8); o = oracle.OracleTable(0, 128, 0.35, 0.8)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    e1 = np.concatenate([keys[:200_000], W.distinct_u64(50_000, seed=99), keys[:30_000]])       # hits, misses, repeats
    e1 = e1[W.shuffle_perm(len(e1), 2)]
    g.profile_enable(True)
    assert g.erase(dev(e1)) == o.erase(e1) == 200_000
    p = g.profile()
    assert "k_erase_fused" in p and "k_erase_mark" not in p, p
    check_state(g, o, 0)
    check_queries(g, o, np.concatenate([keys[:5000], keys[300_000:305_000]]))
    # dense: erase (almost) everything -- ~1270 elements + ~1270 erase keys per chunk do not fit the staging area
    g.profile_reset()
    e2 = keys[200_000:1_450_000]
    assert g.erase(dev(e2)) == o.erase(e2) == len(e2)
    p = g.profile()
    assert "k_erase_fused" in p and "k_erase_mark" in p, p             # tried, rejected, marked + re-laid out
    check_state(g, o, 0)
    g.close()
    # the old path by itself (test hook) gives the same table
    monkeypatch.setenv("KH_DISABLE_STREAM_ERASE", "1")
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8); o = oracle.OracleTable(0, 128, 0.35, 0.8)
    g.insert(dev(keys), dev(vals)); o.insert(keys, vals)
    g.profile_enable(True)
    assert g.erase(dev(e1)) == o.erase(e1)
    assert "k_erase_mark" in g.profile() and "k_erase_fused" not in g.profile()
    check_state(g, o, 0)
    g.close()
    monkeypatch.delenv("KH_DISABLE_STREAM_ERASE")
    # key transform: erase by the other strand
    k = 21
    kk = W.distinct_u64(400_000, seed=5) & np.uint64((1 << (2 * k)) - 1)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash="farm"); g.set_key_transform(k)
    o = oracle.OracleTable(0, 128, 0.35, 0.8, 3, 43); o.set_key_transform(k)
    vv = np.arange(len(kk), dtype=np.uint32)
    assert g.insert(dev(kk), dev(vv)) == o.insert(kk, vv)
    er = np.concatenate([_revcomp(kk[:20_000], k), kk[150_000:155_000]])          # (~100 erase keys next to ~1560 elements per chunk)
    g.profile_enable(True)
    assert g.erase(dev(er)) == o.erase(er)
    assert "k_erase_fused" in g.profile() and "k_erase_mark" not in g.profile()
    check_state(g, o, 0)
    g.close()


@pytest.mark.parametrize("hname,hid", [("murmur3avx64", 1), ("farm", 3)])
def test_duplicate_heavy_batches_into_a_loaded_table(oracle, hname, hid):
    """duplicate-heavy batches into a loaded Robin Hood table on the general path (k_dedup: ~370 distinct keys per partition, half of them
    already in the 3*10^6-key table, every key three times): as a first-wins insert, as an update, as a reducer-plus insert with values
    and as a counting insert (8-byte records); key / value sets against the oracle and a numpy model, info bytes against the oracle.
    (Written for a variant of k_dedup that read the keys' chunk of the table as a stream instead of probing it key by key -- correct, but no
    faster: the probes of one partition already share their chunk's sectors in L2 -- and kept as a parity case of this shape.)"""
    n0 = 3_000_000
    base = W.distinct_u64(n0, seed=81); bv = np.arange(n0, dtype=np.uint32)
    fresh = W.distinct_u64(1_500_000, seed=82)
    dist = np.concatenate([base[:1_500_000], fresh])
    batch = np.concatenate([dist, dist[::-1], dist])[W.shuffle_perm(3 * len(dist), 5)]
    vals = (np.arange(len(batch), dtype=np.uint64) * np.uint64(2654435761) % np.uint64(2**32)).astype(np.uint32)
    # first-wins insert
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=hname); o = oracle.OracleTable(0, 128, 0.35, 0.8, hid, 43)
    assert g.insert(dev(base), dev(bv)) == o.insert(base, bv)
    g.profile_enable(True)
    assert g.insert(dev(batch), dev(vals)) == o.insert(batch, vals) == 1_500_000
    assert "k_dedup" in g.profile(), g.profile()          # (9*10^6 pairs predict four times the capacity: the general path)
    check_state(g, o, 0)
    # update of the same keys: every one of them is in the table now, the last value of the batch wins
    g.update(dev(batch), dev(vals))
    ok, ov = o.sorted_items()
    uk, first_rev = np.unique(batch[::-1], return_index=True)
    ov = ov.copy(); ov[np.searchsorted(ok, uk)] = vals[::-1][first_rev]
    sk, sv = g.sorted_items()
    assert np.array_equal(sk, ok) and np.array_equal(sv, ov)
    assert g.size() == o.size() and g.capacity() == o.capacity() and np.array_equal(g.export_info(), o.export_info())
    g.close()
    # reducer plus with values, then a counting insert (no values: 8-byte records)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=hname); o = oracle.OracleTable(0, 128, 0.35, 0.8, hid, 43)
    g.insert(dev(base), dev(bv)); o.insert(base, bv)
    assert g.insert_reduce_plus(dev(batch), dev(vals)) == 1_500_000
    o.insert(batch, np.zeros(len(batch), dtype=np.uint32))
    assert g.insert_reduce_plus(dev(batch)) == 0
    uk, inv = np.unique(batch, return_inverse=True)
    add = np.zeros(len(uk), dtype=np.uint64); np.add.at(add, inv, vals.astype(np.uint64) + np.uint64(1))
    exp = dict(zip(base.tolist(), bv.tolist()))
    for k, a in zip(uk.tolist(), add.tolist()):
        exp[k] = (exp.get(k, 0) + a) & 0xFFFFFFFF
    sk, sv = g.sorted_items()
    ek = np.array(sorted(exp), dtype=np.uint64)
    assert np.array_equal(sk, ek) and np.array_equal(sv, np.array([exp[int(k)] for k in ek], dtype=np.uint32))
    assert g.size() == o.size() and g.capacity() == o.capacity() and np.array_equal(g.export_info(), o.export_info())
    g.close()


@pytest.mark.parametrize("n0,nb,cap", [(3_000_000, 250_000, 1 << 22), (5_000_000, 1_000_000, 1 << 23)])
def test_insert_into_a_loaded_table_as_an_ordered_stream(oracle, monkeypatch, n0, nb, cap):
    """k_insert_stream: a batch into a loaded Robin Hood table whose capacity stays (the table's elements in slot order, the batch's records
    chained per home bucket).  New keys, keys the table holds, keys given several times inside the batch; exact partition offsets (2^22
    buckets) and fixed slots (2^23); first-wins insert against the oracle, reducer-plus (with values, then counting) against a numpy model,
    a bimolecule table fed the other strand; the staging form (test hook) leaves the same table."""
    base = W.distinct_u64(n0, seed=91); bv = np.arange(n0, dtype=np.uint32)
    fresh = W.distinct_u64(nb * 6 // 10, seed=92)
    batch = np.concatenate([fresh, base[: nb * 24 // 100], fresh[: nb * 16 // 100]])[W.shuffle_perm(nb, 7)]      # 60 % new, 24 % held, 16 % repeats
    vals = (np.arange(nb, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(2**32)).astype(np.uint32)
    infos = []
    for ordered in (True, False):
        if not ordered:
            monkeypatch.setenv("KH_DISABLE_ORDERED_INSERT", "1")
        g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8); o = oracle.OracleTable(0, 128, 0.35, 0.8)
        assert g.insert(dev(base), dev(bv)) == o.insert(base, bv) and g.capacity() == cap
        g.profile_enable(True)
        assert g.insert(dev(batch), dev(vals)) == o.insert(batch, vals) == len(fresh) and g.capacity() == cap
        p = g.profile()
        assert "k_insert_fused" in p and "k_dedup" not in p, p
        check_state(g, o, 0)
        check_queries(g, o, np.concatenate([fresh[:3000], base[:3000], W.distinct_u64(3000, seed=93)]))
        infos.append(g.export_info().copy())
        g.close()
    monkeypatch.delenv("KH_DISABLE_ORDERED_INSERT")
    assert np.array_equal(infos[0], infos[1])
    # Reducer = std::plus: with values, then the same batch again (every key held now)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    g.insert(dev(base), dev(bv))
    g.profile_enable(True)
    assert g.insert_reduce_plus(dev(batch), dev(vals)) == len(fresh)
    assert g.insert_reduce_plus(dev(batch), dev(vals)) == 0
    assert "k_insert_fused" in g.profile(), g.profile()
    uk, inv = np.unique(batch, return_inverse=True)
    add = np.zeros(len(uk), dtype=np.uint64); np.add.at(add, inv, vals.astype(np.uint64))
    exp = dict(zip(base.tolist(), bv.tolist()))
    for k, a in zip(uk.tolist(), add.tolist()):
        exp[k] = (exp.get(k, 0) + 2 * a) & 0xFFFFFFFF
    sk, sv = g.sorted_items()
    ek = np.array(sorted(exp), dtype=np.uint64)
    assert np.array_equal(sk, ek) and np.array_equal(sv, np.array([exp[int(k)] for k in ek], dtype=np.uint32))
    assert g.capacity() == cap and np.array_equal(g.export_info(), infos[0])
    g.close()
    if cap == 1 << 22:      # bimolecule table: the batch brings the other strand of keys the table holds, and both strands of new ones
        k = 31
        kk = W.distinct_u64(n0 + 200_000, seed=94) & np.uint64((1 << (2 * k)) - 1)
        kk = kk[np.unique(np.minimum(kk, _revcomp(kk, k)), return_index=True)[1]]
        kb, kn = kk[:n0 - 100_000], kk[n0 - 100_000:][:150_000]
        g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash="farm"); g.set_key_transform(k)
        o = oracle.OracleTable(0, 128, 0.35, 0.8, 3, 43); o.set_key_transform(k)
        vb = np.arange(len(kb), dtype=np.uint32)
        assert g.insert(dev(kb), dev(vb)) == o.insert(kb, vb)
        b2 = np.concatenate([kn, _revcomp(kn[:50_000], k), _revcomp(kb[:60_000], k)])[W.shuffle_perm(len(kn) + 110_000, 9)]
        v2 = np.arange(len(b2), dtype=np.uint32) + np.uint32(7_000_000)
        g.profile_enable(True)
        assert g.insert(dev(b2), dev(v2)) == o.insert(b2, v2) == len(kn)
        assert "k_insert_fused" in g.profile() and "k_dedup" not in g.profile(), g.profile()
        check_state(g, o, 0)
        g.close()


def test_bulk_build_of_a_table_filled_to_0_9(oracle):
    """max load factor 0.9, filled to the threshold exactly (7 549 747 keys -> 2^23 buckets, 1843 records per chunk on average): the lean bulk
    build's staging arrays (2016 records) would be too small for a chunk or two of 4096, so the build goes to k_build_fused -- ONE launch, no
    discarded attempt, no general path -- and equals the oracle; at 0.8 the same call takes the lean kernel (k_build_fused is the profile name
    of both; KH_DISABLE_LEAN_BUILD gives the same table)."""
    cap = 1 << 23
    n = int(np.float32(cap) * np.float32(0.9))
    keys = W.distinct_u64(n, seed=71); vals = np.arange(n, dtype=np.uint32)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.9); o = oracle.OracleTable(0, 128, 0.35, 0.9)
    g.profile_enable(True)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals) == n and g.capacity() == cap
    p = g.profile()
    assert p["k_build_fused"][0] == 1 and "k_dedup" not in p and p["k_part_scatter"][0] == 2, p
    check_state(g, o, 0)
    check_queries(g, o, np.concatenate([keys[:4000], W.distinct_u64(4000, seed=72)]))
    g.close()


@pytest.mark.parametrize("hname,hid", [("murmur3avx64", 1), ("farm", 3)])
def test_batch_erase_as_an_ordered_stream(oracle, monkeypatch, hname, hid):
    """the ordered-stream form of the Robin Hood batch erase (k_erase_stream: a chunk's slots scanned in slot order, survivors stored at
    max(home, slot of the one before + 1)): a table of 2^23 buckets (histogram-free partition of the erase keys), hits, misses and keys given
    twice; a second and a third batch on the table the first one left (the third empties most chunks); the staging form on the same input
    (test hook) leaves the same table; a bimolecule table erased by the other strand.  Info bytes, sizes, capacities, key / value sets and
    query results against the oracle."""
    n = 5_000_000
    keys = W.distinct_u64(n, seed=61); vals = np.arange(n, dtype=np.uint32)
    e1 = np.concatenate([keys[:600_000], W.distinct_u64(100_000, seed=98), keys[:50_000]])
    e1 = e1[W.shuffle_perm(len(e1), 3)]
    e2 = keys[400_000:1_700_000]                      # 2*10^5 of them are gone already
    e3 = keys[1_700_000:3_900_000][::-1].copy()      # ~540 keys per chunk: two per lane in k_erase_stream (slots of 715 <= 832)
    for ordered in (True, False):
        if not ordered:
            monkeypatch.setenv("KH_DISABLE_ORDERED_ERASE", "1")
        g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=hname); o = oracle.OracleTable(0, 128, 0.35, 0.8, hid, 43)
        assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals) and g.capacity() == 1 << 23
        g.profile_enable(True)
        for e in ((e1, e2, e3) if ordered else (e1,)):
            assert g.erase(dev(e)) == o.erase(e)
            check_state(g, o, 0)
        p = g.profile()
        assert "k_erase_fused" in p and "k_erase_mark" not in p, p
        check_queries(g, o, np.concatenate([keys[:3000], keys[2_000_000:2_003_000], keys[4_000_000:4_003_000]]))
        if ordered:                                     # the table goes on working: new keys into the gaps, another erase
            more = W.distinct_u64(300_000, seed=62); mv = np.arange(300_000, dtype=np.uint32)
            assert g.insert(dev(more), dev(mv)) == o.insert(more, mv)
            assert g.erase(dev(more[:100_000])) == o.erase(more[:100_000]) == 100_000
            check_state(g, o, 0)
        g.close()
    monkeypatch.delenv("KH_DISABLE_ORDERED_ERASE")
    k = 31
    kk = W.distinct_u64(4_500_000, seed=63) & np.uint64((1 << (2 * k)) - 1)
    kk = kk[np.unique(np.minimum(kk, _revcomp(kk, k)), return_index=True)[1]]      # one strand per k-mer
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash=hname); g.set_key_transform(k)
    o = oracle.OracleTable(0, 128, 0.35, 0.8, hid, 43); o.set_key_transform(k)
    vv = np.arange(len(kk), dtype=np.uint32)
    assert g.insert(dev(kk), dev(vv)) == o.insert(kk, vv) and g.capacity() == 1 << 23
    er = np.concatenate([_revcomp(kk[:300_000], k), kk[1_000_000:1_200_000], _revcomp(kk[:20_000], k)])
    g.profile_enable(True)
    assert g.erase(dev(er)) == o.erase(er) == 500_000
    assert "k_erase_fused" in g.profile() and "k_erase_mark" not in g.profile()
    check_state(g, o, 0)
    g.close()


@pytest.mark.parametrize("kname,cls,kind", KINDS)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sequences_with_reducer_plus_and_aborts(oracle, kname, cls, kind, seed):
    """random sequences mixing Reducer = std::plus inserts (one call / streamed, repeatable or not, with and without values), first-wins
    inserts, batch erases and ABORTED streamed inserts.  Key set, capacity and info bytes against the oracle (fed the same key batches);
    values against a dictionary model (wrapping 32-bit sums).  Covers the deferred (slot, sum) lists of k_dedup / k_apply_plus, the
    in-place and small-batch paths, the streaming batch erase and kh_insert_abort in whatever state the sequence reaches them."""
    rng = np.random.default_rng(1000 * seed + kind)
    uni = W.distinct_u64(300_000, seed=50 + seed)
    g = cls(128, 0.35, 0.8); o = oracle.OracleTable(kind, 128, 0.35, 0.8)
    model = {}
    for step in range(30):
        op = int(rng.integers(0, 6))
        m = int(rng.choice([1, 7, 300, 5_000, 60_000, 250_000]))
        ks = uni[rng.integers(0, len(uni), m)]
        vs = rng.integers(0, 2**32, m, dtype=np.uint32)
        if op <= 1:                                              # counting / summing insert, one call
            use_v = bool(rng.integers(0, 2))
            g.insert_reduce_plus(dev(ks), dev(vs) if use_v else None)
            o.insert(ks, vs)
            for k, v in zip(ks.tolist(), vs.tolist() if use_v else [1] * m):
                model[k] = (model.get(k, 0) + v) & 0xFFFFFFFF
        elif op == 2:                                            # the same, streamed; sometimes aborted half-way
            cuts = sorted(set([0, m] + [int(x) for x in rng.integers(0, m + 1, 3)]))
            rep = bool(rng.integers(0, 2))
            abort = rng.random() < 0.3
            try:
                g.insert_begin(m, reduce_plus=True, repeatable=rep)
                for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                    if abort and i == len(cuts) - 2:
                        break
                    g.insert_feed(dev(ks[a:b]))
                if abort:
                    g.insert_abort()
                else:
                    g.insert_end()
            except kh.KhRetry:
                assert rep and not abort
                g.insert_begin(m, reduce_plus=True)
                for a, b in zip(cuts[:-1], cuts[1:]):
                    g.insert_feed(dev(ks[a:b]))
                g.insert_end()
            if not abort:
                o.insert(ks, vs)
                for k in ks.tolist():
                    model[k] = (model.get(k, 0) + 1) & 0xFFFFFFFF
        elif op == 3:                                            # first value wins
            g.insert(dev(ks), dev(vs)); o.insert(ks, vs)
            for k, v in zip(ks.tolist(), vs.tolist()):
                model.setdefault(k, v)
        elif op == 4:
            assert g.erase(dev(ks)) == o.erase(ks)
            for k in ks.tolist():
                model.pop(k, None)
        else:
            fk, fv = g.find(dev(ks))
            fk = host(fk, np.uint64); fv = host(fv, np.uint32)
            exp = [(k, model[k]) for k in ks.tolist() if k in model]
            assert fk.tolist() == [k for k, _ in exp] and fv.tolist() == [v for _, v in exp]
        assert (g.size(), g.capacity()) == (o.size(), o.capacity()) == (len(model), o.capacity()), (step, op, m)
        if kind == 0:
            assert np.array_equal(g.export_info(), o.export_info()), (step, op, m)
        sk, sv = g.sorted_items()
        mk = np.array(sorted(model), dtype=np.uint64)
        assert np.array_equal(sk, mk) and np.array_equal(sv, np.array([model[k] for k in mk.tolist()], dtype=np.uint32)), (step, op, m)
    g.close()


@pytest.mark.parametrize("hidden", [0, 300])
def test_lean_bulk_build_with_key_transform(oracle, hidden):
    """k_build_lean (duplicate-free sample, 12-byte records, 4 workgroups per CU) on a bimolecule table: 4e6 31-mers hashed and compared as
    min(k-mer, reverse complement).  hidden = 300: that many k-mers occur a second time as their OTHER strand, none at a sampled
    position -- equal keys under the transform, which the group check after the placement must find (retry with 16-byte records, first
    occurrence's bits stored)."""
    k, n = 31, 4_000_000
    top = np.uint64((1 << (2 * k)) - 1)
    keys = W.distinct_u64(n, seed=91) & top
    if hidden:
        pos = (n // 65536) * np.arange(2000, 2000 + hidden) + 5
        keys[pos] = _revcomp(keys[(n // 65536) * np.arange(9000, 9000 + hidden) + 9], k)
    vals = np.arange(n, dtype=np.uint32)
    g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash="farm"); g.set_key_transform(k)
    o = oracle.OracleTable(0, 128, 0.35, 0.8, 3, 43); o.set_key_transform(k)
    g.profile_enable(True)
    assert g.insert(dev(keys), dev(vals)) == o.insert(keys, vals)
    p = g.profile()
    assert p["k_build_fused"][0] == (2 if hidden else 1) and "k_dedup" not in p, p      # (the label covers k_build_lean: one launch, or lean + its retry)
    check_state(g, o, 0)
    check_queries(g, o, np.concatenate([keys[:3000], _revcomp(keys[5000:8000], k), W.distinct_u64(2000, seed=92) & top]))
    g.close()
