"""GPU: the HIP tables against the committed golden fixtures: results of the REAL reference LP table
(tests/golden/lp_ref_*.npz), smhasher KATs (murmur3_kat.npz) and the RH regression vectors.
Nothing here reads /root/reference."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import kmerhash_amd as kh  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HNAME = {0: "identity", 1: "murmur3avx64", 2: "murmur", 3: "farm"}


def test_murmur3_kat_on_gpu():
    g = np.load(os.path.join(GOLD, "murmur3_kat.npz"))
    one = np.array([1], dtype=np.uint64)
    assert int(kh.hash_batch(one, "murmur3avx64", 43)[0]) == 0xdbcde6617f85bf2a
    assert int(kh.hash_batch(one, "murmur", 43)[0]) == 0x252c590efc7e7503
    for si, s in enumerate(g["seeds"]):
        assert np.array_equal(kh.hash_batch(g["keys"], "murmur3avx64", int(s)), g["x86_128_lo64"][si])
        assert np.array_equal(kh.hash_batch(g["keys"], "murmur", int(s)), g["x64_128_h0"][si])


def replay(t, g, pfx, rh):
    assert t.insert(g["keys"], g["vals"]) == int(g[pfx + "n_inserted"])
    assert (t.size(), t.capacity()) == (int(g[pfx + "size1"]), int(g[pfx + "cap1"]))
    if rh is None:
        pass           # results only (a Robin Hood table replaying the reference LP table's outputs)
    elif rh:
        assert np.array_equal(t.export_info(), g[pfx + "info1"])
        assert np.array_equal(t.export_info() >= 0x80, g["ref_lp_occupied1"])
    else:
        # the GPU LP table stores clusters in home order; the occupied SLOT SET still equals the reference's
        assert np.array_equal(t.export_info() < 0x40, g[pfx + "info1"] < 0x40)
    assert np.array_equal(t.count(g["q"]), g[pfx + "count1"])
    fk, fv = t.find(g["q"])
    assert np.array_equal(fk, g[pfx + "findk1"]) and np.array_equal(fv, g[pfx + "findv1"])
    assert t.erase(g["er"]) == int(g[pfx + "n_erased"])
    assert (t.size(), t.capacity()) == (int(g[pfx + "size2"]), int(g[pfx + "cap2"]))
    if rh:
        assert np.array_equal(t.export_info(), g[pfx + "info2"])
    assert np.array_equal(t.count(g["q"]), g[pfx + "count2"])
    assert t.insert(g["keys2"], g["vals2"]) == int(g[pfx + "n_inserted2"])
    assert (t.size(), t.capacity()) == (int(g[pfx + "size3"]), int(g[pfx + "cap3"]))
    sk, sv = t.sorted_items()
    assert np.array_equal(sk, g[pfx + "items3k"]) and np.array_equal(sv, g[pfx + "items3v"])
    assert np.array_equal(t.count(g["q"]), g[pfx + "count3"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "lp_ref_*.npz"))), ids=os.path.basename)
def test_gpu_lp_matches_reference_fixture(path):
    g = np.load(path)
    cap0, hid, seed = (int(x) for x in g["params"])
    t = kh.hashmap_linearprobe_doubling(cap0, float(g["lfs"][0]), float(g["lfs"][1]), hash=HNAME[hid], seed=seed)
    replay(t, g, "lp_", rh=False)
    t.close()


KIND_FREE = ("n_inserted", "size1", "count1", "findk1", "findv1", "n_erased", "size2", "count2", "n_inserted2", "size3", "items3k", "items3v", "count3")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "rh_oracle_*.npz"))), ids=os.path.basename)
def test_gpu_rh_matches_fixture(path):
    g = np.load(path)
    cap0, hid, seed = (int(x) for x in g["params"])
    t = kh.hashmap_robinhood_doubling(cap0, float(g["lfs"][0]), float(g["lfs"][1]), hash=HNAME[hid], seed=seed)
    replay(t, g, "rh_", rh=True)
    t.close()
    # the RESULTS of the Robin Hood table pinned to the REAL reference: the sibling lp_ref_<name>.npz holds the reference LP table's
    # outputs for the same inputs, and everything but layout and post-erase capacity is kind-independent map semantics
    ref = np.load(path.replace("rh_oracle_", "lp_ref_"))
    for f in ("keys", "vals", "q", "er", "keys2", "vals2"):
        assert np.array_equal(g[f], ref[f])
    pinned = {k: ref[k] for k in ("keys", "vals", "q", "er", "keys2", "vals2")}
    pinned.update({"lp_" + f: ref["lp_" + f] for f in KIND_FREE})
    pinned.update({"lp_cap1": g["rh_cap1"], "lp_cap2": g["rh_cap2"], "lp_cap3": g["rh_cap3"], "lp_info1": None})
    t = kh.hashmap_robinhood_doubling(cap0, float(g["lfs"][0]), float(g["lfs"][1]), hash=HNAME[hid], seed=seed)
    replay(t, pinned, "lp_", rh=None)
    t.close()


@pytest.mark.parametrize("kname,cls", [("rh", kh.hashmap_robinhood_doubling), ("lp", kh.hashmap_linearprobe_doubling)])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ops_ref_*.npz"))), ids=os.path.basename)
def test_gpu_single_key_update_and_counting_match_reference(path, kname, cls):
    """insert(value_type), update(k,v), erase(key) and a counting insert against outputs of the REAL reference LP table's members
    (tests/golden/make_golden.py ops_scenario): kind-independent results for both GPU tables, capacities for the LP one"""
    g = np.load(path)
    cap0, hid, seed = (int(x) for x in g["params"])
    lp = kname == "lp"
    t = cls(cap0, float(g["lfs"][0]), float(g["lfs"][1]), hash=HNAME[hid], seed=seed)
    assert t.insert(g["keys"], g["vals"]) == int(g["n_inserted"])
    flags = np.array([t.insert_one(int(k), int(v)) for k, v in zip(g["one_k"], g["one_v"])], dtype=np.uint8)
    assert np.array_equal(flags, g["one_flags"])
    assert t.size() == int(g["size_one"]) and (not lp or t.capacity() == int(g["lp_cap_one"]))
    t.update(g["upd_k"], g["upd_v"])                 # kh_update = the update(k,v) calls in batch order: last value wins
    assert t.size() == int(g["size_upd"]) and (not lp or t.capacity() == int(g["lp_cap_upd"]))
    sk, sv = t.sorted_items()
    assert np.array_equal(sk, g["items_upd_k"]) and np.array_equal(sv, g["items_upd_v"])
    assert np.array_equal(np.array([t.erase_one(int(k)) for k in g["er_k"]], dtype=np.uint8), g["er_flags"])
    assert t.size() == int(g["size_er"]) and (not lp or t.capacity() == int(g["lp_cap_er"]))
    t.insert_reduce_plus(g["cnt_k"])                 # Reducer = std::plus, 1 per occurrence
    sk, sv = t.sorted_items()
    assert t.size() == int(g["size_cnt"]) and np.array_equal(sk, g["items_cnt_k"]) and np.array_equal(sv, g["items_cnt_v"])
    t.close()
