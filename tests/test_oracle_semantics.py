"""CPU: the reference's own test strategy (test/unit/test_hashmap_robinhood_doubling.cpp:97-334,
test_hashmap_linearprobe_doubling.cpp:84-195) restated for the oracle: differential against a first-wins
associative container (std::unordered_map::emplace == dict.setdefault)."""
import numpy as np
import pytest

from kmerhash_amd import workloads as W


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("hid", [0, 1, 2, 3])
def test_differential_vs_dict(oracle, kind, hid):
    rng = np.random.default_rng(hid * 10 + kind)
    # reference: 100000 random (key,val), uniform_int_distribution(2, max-2)
    keys = rng.integers(2, 2**63, 100_000, dtype=np.uint64)
    keys[::7] = keys[:: 7][::-1]   # force repeats
    keys[1::5] = keys[0::5][: len(keys[1::5])]
    vals = rng.integers(0, 2**32, len(keys), dtype=np.uint32)
    gold = {}
    for k, v in zip(keys.tolist(), vals.tolist()):
        gold.setdefault(k, v)
    t = oracle.OracleTable(kind, 128, None, None, hid, 43)
    assert t.insert(keys, vals) == len(gold)
    sk, sv = t.sorted_items()
    gk = np.array(sorted(gold), dtype=np.uint64)
    assert np.array_equal(sk, gk)
    assert np.array_equal(sv, np.array([gold[int(k)] for k in gk], dtype=np.uint32))
    q = np.concatenate([gk[:5000], rng.integers(2, 2**63, 5000, dtype=np.uint64)])
    assert np.array_equal(t.count(q), np.array([1 if int(k) in gold else 0 for k in q], dtype=np.uint8))
    # erase first half (reference *_map_erase tests): size equality, count==0 for erased, ==1 for kept
    half = gk[: len(gk) // 2]
    assert t.erase(half) == len(half)
    assert t.size() == len(gk) - len(half)
    assert not t.count(half).any() and t.count(gk[len(gk) // 2:]).all()
    assert len(np.unique(t.to_vector()[0])) == t.size()
