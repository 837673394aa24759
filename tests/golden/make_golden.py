#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container, where /root/reference exists):

  murmur3_kat.npz   smhasher MurmurHash3 (scikit-learn 1.7.2's copy of the third-party code the reference's
                    scalar murmur functors call, hash_new.hpp:83,206-235) on fixed inputs; plus the two KATs
                    SURVEY.md §8c recorded from the reference's own AVX implementation.
  lp_ref_<name>.npz outputs of the REAL reference fsc::hashmap_linearprobe_doubling (oracle/_ref/libref_lp.so,
                    compiled from /root/reference/include/kmerhash/hashmap_linearprobe.hpp as it lies) on
                    seeded inputs: sizes, capacities, info bytes, slot contents, count / find / erase results.
  rh_oracle_<name>.npz  Robin Hood regression vectors produced by the ORACLE (the RH reference header is
                    unbuildable here -- see oracle/kh_oracle.hpp); they carry the occupancy bitmap of the real
                    reference LP table for the same keys, which the RH table must reproduce slot for slot.

  ops_ref_<name>.npz    single-key insert / update / single-key erase / a counting insert through the REAL reference LP table's members:
                    kind-independent map results, which the Robin Hood table (GPU and oracle) must reproduce as well.

Fixtures are data only (inputs + expected outputs); no reference source is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402
from kmerhash_amd import workloads as W  # noqa: E402


def murmur_kat():
    rng = np.random.default_rng(12345)
    keys = np.concatenate([
        np.array([0, 1, 2, 3, 0xFF, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFFFFFFFFFF, 1 << 63, 0x0123456789ABCDEF], dtype=np.uint64),
        rng.integers(0, 2**63, 600, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 600, dtype=np.uint64)])
    seeds = np.array([43, 0, 9876543, 0xFFFFFFFF], dtype=np.uint64)
    x86 = np.zeros((len(seeds), len(keys)), dtype=np.uint64)
    x64 = np.zeros_like(x86)
    for si, s in enumerate(seeds):
        for ki, k in enumerate(keys):
            b = int(k).to_bytes(8, "little")
            h = O.smhasher_x86_128(b, int(s))
            x86[si, ki] = np.uint64(h[0]) | (np.uint64(h[1]) << np.uint64(32))
            x64[si, ki] = O.smhasher_x64_128(b, int(s))[0]
    # variable-length inputs: every tail length of both variants
    blob = rng.integers(0, 256, 64, dtype=np.uint8)
    lens = np.arange(0, 49, dtype=np.int32)
    vx86 = np.zeros((len(lens), 4), dtype=np.uint32)
    vx64 = np.zeros((len(lens), 2), dtype=np.uint64)
    for i, L in enumerate(lens):
        vx86[i] = O.smhasher_x86_128(blob[:L].tobytes(), 43)
        vx64[i] = O.smhasher_x64_128(blob[:L].tobytes(), 43)
    np.savez_compressed(os.path.join(HERE, "murmur3_kat.npz"), keys=keys, seeds=seeds, x86_128_lo64=x86, x64_128_h0=x64,
                        blob=blob, lens=lens, var_x86_128=vx86, var_x64_128=vx64,
                        survey_kat=np.array([0xdbcde6617f85bf2a, 0x252c590efc7e7503], dtype=np.uint64))


def scenario(name, n, seed, hash_id, cap0=128, mn=0.35, mx=0.8):
    keys, vals = W.w1_benchmark_hashtables(n, seed=seed)
    if hash_id == O.HASH_IDENTITY:
        keys = W.splitmix64(keys)
    q = W.queries_hits_and_misses(keys, max(2 * n // 3, 4), 0.5, seed=seed + 1)
    er = q[: len(q) // 2]
    keys2, vals2 = W.w1_benchmark_hashtables(max(n // 2, 2), seed=seed + 100)
    out = dict(keys=keys, vals=vals, q=q, er=er, keys2=keys2, vals2=vals2,
               params=np.array([cap0, hash_id, 43], dtype=np.uint64), lfs=np.array([mn, mx], dtype=np.float32))

    def run(t, pfx):
        out[pfx + "n_inserted"] = np.uint64(t.insert(keys, vals))
        out[pfx + "size1"] = np.uint64(t.size()); out[pfx + "cap1"] = np.uint64(t.capacity())
        out[pfx + "info1"] = t.export_info()
        k, v = t.export_slots(); out[pfx + "slotk1"] = k; out[pfx + "slotv1"] = v
        out[pfx + "count1"] = t.count(q)
        fk, fv = t.find_compact(q); out[pfx + "findk1"] = fk; out[pfx + "findv1"] = fv
        out[pfx + "n_erased"] = np.uint64(t.erase(er))
        out[pfx + "size2"] = np.uint64(t.size()); out[pfx + "cap2"] = np.uint64(t.capacity())
        out[pfx + "info2"] = t.export_info()
        out[pfx + "count2"] = t.count(q)
        out[pfx + "n_inserted2"] = np.uint64(t.insert(keys2, vals2))
        out[pfx + "size3"] = np.uint64(t.size()); out[pfx + "cap3"] = np.uint64(t.capacity())
        sk, sv = t.sorted_items(); out[pfx + "items3k"] = sk; out[pfx + "items3v"] = sv
        out[pfx + "count3"] = t.count(q)

    ref = O.RefLPTable(cap0, mn, mx, hash_id, 43)
    run(ref, "lp_")
    np.savez_compressed(os.path.join(HERE, "lp_ref_%s.npz" % name), **out)

    # Robin Hood: oracle-generated + the reference LP occupancy for the same key set / capacity / hash
    rh = dict(keys=keys, vals=vals, q=q, er=er, keys2=keys2, vals2=vals2, params=out["params"], lfs=out["lfs"])
    out = rh
    o = O.OracleTable(O.KIND_RH, cap0, mn, mx, hash_id, 43)
    run(o, "rh_")
    rh["ref_lp_occupied1"] = (np.load(os.path.join(HERE, "lp_ref_%s.npz" % name))["lp_info1"] < 0x40)
    np.savez_compressed(os.path.join(HERE, "rh_oracle_%s.npz" % name), **rh)


def ops_scenario(name, n, seed, hash_id, cap0=128, mn=0.35, mx=0.8):
    """ops_ref_<name>.npz: the single-key members and update() of the REAL reference LP table, and a counting insert composed from
    its own find / update / insert (what Reducer = std::plus computes: old value + 1 per occurrence).  All results here are map
    semantics independent of the table kind (first value wins, last update wins, erase counts, sums): the Robin Hood GPU table and
    the Robin Hood oracle are held to them too; capacities are recorded for the LP kind only."""
    keys, vals = W.w1_benchmark_hashtables(n, seed=seed)
    if hash_id == O.HASH_IDENTITY:
        keys = W.splitmix64(keys)
    h = n // 2
    fresh = W.distinct_u64(400, seed=seed + 5)
    one_k = np.concatenate([keys[h:h + 200], fresh[:100], keys[:100]])                  # new, new, duplicates
    one_k = one_k[W.shuffle_perm(len(one_k), seed + 6)]
    one_v = (np.arange(len(one_k), dtype=np.uint32) + np.uint32(5_000_000))
    upd_k = np.concatenate([keys[:300], fresh[100:250], keys[100:200], fresh[100:150]])   # existing, new, and keys given twice
    upd_k = upd_k[W.shuffle_perm(len(upd_k), seed + 7)]
    upd_v = (np.arange(len(upd_k), dtype=np.uint32) + np.uint32(7_000_000))
    er_k = np.concatenate([keys[50:250], fresh[250:350], keys[50:100]])                   # hits, misses, already erased
    er_k = er_k[W.shuffle_perm(len(er_k), seed + 8)]
    rng = np.random.default_rng(seed)
    cnt_k = np.concatenate([keys[rng.integers(0, n, 3000)], fresh[300:400][rng.integers(0, 100, 800)]])
    out = dict(keys=keys[:h], vals=vals[:h], one_k=one_k, one_v=one_v, upd_k=upd_k, upd_v=upd_v, er_k=er_k, cnt_k=cnt_k,
               params=np.array([cap0, hash_id, 43], dtype=np.uint64), lfs=np.array([mn, mx], dtype=np.float32))
    t = O.RefLPTable(cap0, mn, mx, hash_id, 43)
    out["n_inserted"] = np.uint64(t.insert(keys[:h], vals[:h]))
    out["one_flags"] = np.array([1 if t.insert_one(int(k), int(v)) else 0 for k, v in zip(one_k, one_v)], dtype=np.uint8)
    out["size_one"] = np.uint64(t.size()); out["lp_cap_one"] = np.uint64(t.capacity())
    for k, v in zip(upd_k, upd_v):
        t.update_one(int(k), int(v))
    out["size_upd"] = np.uint64(t.size()); out["lp_cap_upd"] = np.uint64(t.capacity())
    sk, sv = t.sorted_items(); out["items_upd_k"] = sk; out["items_upd_v"] = sv
    out["er_flags"] = np.array([t.erase_one(int(k)) for k in er_k], dtype=np.uint8)
    out["size_er"] = np.uint64(t.size()); out["lp_cap_er"] = np.uint64(t.capacity())
    for k in cnt_k:                       # counting insert, one occurrence at a time, through the reference's own members
        fk, fv = t.find_compact(np.array([k], dtype=np.uint64))
        if len(fk):
            t.update_one(int(k), (int(fv[0]) + 1) & 0xFFFFFFFF)
        else:
            t.insert_one(int(k), 1)
    out["size_cnt"] = np.uint64(t.size())
    sk, sv = t.sorted_items(); out["items_cnt_k"] = sk; out["items_cnt_v"] = sv
    np.savez_compressed(os.path.join(HERE, "ops_ref_%s.npz" % name), **out)


def hll_and_io():
    """hll_ref.npz: registers/estimates of the REAL reference hyperloglog64<uint64_t,Hash,12>; io_ref_pairs.bin /
    io_ref_keys.bin: files written by the REAL reference serialize_vector (io_utils.hpp:57-81)."""
    out = {}
    keys, vals = W.w1_benchmark_hashtables(200_000, seed=71)
    for name, ign, hid, n in (("a", 0, O.HASH_MURMUR3_X86, 200_000), ("b", 3, O.HASH_MURMUR3_X86, 50_000),
                              ("c", 0, O.HASH_FARM, 1000), ("d", 0, O.HASH_MURMUR3_X64, 37)):
        r = O.RefHLL(ign, hid, 43)
        r.update(keys[:n])
        out["regs_" + name] = r.registers()
        out["est_" + name] = np.float64(r.estimate())
        out["cfg_" + name] = np.array([ign, hid, n], dtype=np.int64)
    r1 = O.RefHLL(0, O.HASH_MURMUR3_X86, 43); r1.update(keys[:1000])
    r2 = O.RefHLL(0, O.HASH_MURMUR3_X86, 43); r2.update(keys[1000:5000])
    r1.merge(r2)
    out["regs_merge"] = r1.registers(); out["est_merge"] = np.float64(r1.estimate())
    hv = O.hash_batch(O.HASH_MURMUR3_X86, 43, keys[:5000])
    r3 = O.RefHLL(0, O.HASH_MURMUR3_X86, 43); r3.update_via_hashval(hv)
    out["regs_hv"] = r3.registers()
    np.savez_compressed(os.path.join(HERE, "hll_ref.npz"), **out)
    O.ref_serialize_pairs(keys[:500], vals[:500], os.path.join(HERE, "io_ref_pairs.bin"))
    O.ref_serialize_u64(keys[:500], os.path.join(HERE, "io_ref_keys.bin"))


if __name__ == "__main__":
    O.build(("all", "ref", "smhasher"))
    murmur_kat()
    scenario("murmur_1k", 1000, 23, O.HASH_MURMUR3_X86)
    scenario("murmur_20k", 20000, 7, O.HASH_MURMUR3_X86)
    scenario("murmur64_5k", 5000, 11, O.HASH_MURMUR3_X64)
    scenario("farm_5k", 5000, 13, O.HASH_FARM)
    scenario("identity_5k", 5000, 17, O.HASH_IDENTITY)
    scenario("lpdefaults_3k", 3000, 19, O.HASH_MURMUR3_X86, 128, 0.2, 0.6)
    ops_scenario("murmur_4k", 4000, 29, O.HASH_MURMUR3_X86)
    ops_scenario("farm_4k", 4000, 31, O.HASH_FARM)
    hll_and_io()
    print("golden fixtures written to", HERE)
