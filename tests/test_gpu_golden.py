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
    if rh:
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


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "rh_oracle_*.npz"))), ids=os.path.basename)
def test_gpu_rh_matches_fixture(path):
    g = np.load(path)
    cap0, hid, seed = (int(x) for x in g["params"])
    t = kh.hashmap_robinhood_doubling(cap0, float(g["lfs"][0]), float(g["lfs"][1]), hash=HNAME[hid], seed=seed)
    replay(t, g, "rh_", rh=True)
    t.close()
