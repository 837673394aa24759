"""CPU: randomized differential test of the oracle's LP restatement against the REAL reference LP table
(oracle/_ref/libref_lp.so, compiled from the reference tree where it lies).  Skipped where neither the
prebuilt library nor /root/reference exists."""
import numpy as np
import pytest

from kmerhash_amd import workloads as W


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.ref_available():
        pytest.skip("reference LP library not available")
    try:
        oracle.ref_lib()
    except Exception as e:  # pragma: no cover
        pytest.skip("reference LP library not loadable: %r" % (e,))
    return oracle


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("hid", [0, 1, 2, 3])
def test_lp_random_op_sequences(ref, seed, hid):
    rng = np.random.default_rng(seed * 100 + hid)
    o = ref.OracleTable(ref.KIND_LP, 128, 0.35, 0.8, hid, 43)
    r = ref.RefLPTable(128, 0.35, 0.8, hid, 43)
    universe = W.splitmix64(np.arange(40_000, dtype=np.uint64) + np.uint64(seed << 20))
    for step in range(40):
        op = rng.integers(0, 5)
        m = int(rng.integers(0, 6000))
        ks = universe[rng.integers(0, len(universe), m)]
        if op <= 1:
            vs = rng.integers(0, 2**32, m, dtype=np.uint32)
            assert o.insert(ks, vs) == r.insert(ks, vs)
        elif op == 2:
            assert np.array_equal(o.count(ks), r.count(ks))
            a, b = o.find_compact(ks), r.find_compact(ks)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        elif op == 3:
            assert o.erase(ks) == r.erase(ks)
        else:
            for k in ks[:20]:
                assert o.erase_one(int(k)) == r.erase_one(int(k))
                o.update_one(int(k), 7); r.update_one(int(k), 7)
        assert (o.size(), o.capacity(), o.max_load(), o.min_load()) == (r.size(), r.capacity(), r.max_load(), r.min_load())
        assert np.array_equal(o.export_info(), r.export_info())
        ok, ov = o.export_slots(); rk, rv = r.export_slots()
        occ = o.export_info() < 0x40
        assert np.array_equal(ok[occ], rk[occ]) and np.array_equal(ov[occ], rv[occ])


@pytest.mark.parametrize("n", [10, 5000, 200_000])
def test_rh_occupancy_equals_reference_lp(ref, n):
    """a Robin Hood table and a linear-probing table over the same keys, hash and capacity occupy the same
    slots; the canonical RH info array follows from occupancy and home buckets alone"""
    keys, vals = W.w1_benchmark_hashtables(n, seed=n)
    rh = ref.OracleTable(ref.KIND_RH, 128, 0.35, 0.8)
    lp = ref.RefLPTable(128, 0.35, 0.8)
    assert rh.insert(keys, vals) == lp.insert(keys, vals)
    assert rh.capacity() == lp.capacity()
    assert np.array_equal(rh.export_info() >= 0x80, lp.export_info() < 0x40)
    a, b = rh.sorted_items(), lp.sorted_items()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
