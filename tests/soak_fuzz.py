"""soak: long differential fuzz of both tables against the CPU oracle (lives under tests/ because it uses oracle/; not collected by
pytest: run it by hand on the GPU box).
usage: python tests/soak_fuzz.py [seconds] [first_seed] [big] [one] [verbose]   -- prints the failing (seed, step, op) if anything differs"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
from oracle import oracle_py as O
from test_gpu_parity import check_state, check_queries, dev

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
only_one = "one" in sys.argv[3:]
verbose = "verbose" in sys.argv[3:]
big = "big" in sys.argv[3:]          # multi-million-key batches: two-pass partitions, many chunks
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KINDS = [(kh.hashmap_robinhood_doubling, 0), (kh.hashmap_linearprobe_doubling, 1)]
HASHES = [("murmur3avx64", 1), ("murmur", 2), ("farm", 3)]
t_end = time.time() + budget


def oracle_plus(o, ks, vs, mx):
    """Reducer = std::plus on the oracle (which has no reducer): membership, size and capacity through ONE insert of the batch (the
    capacity rule of kh_insert_reduce_plus is the batch insert's), then every key of the batch gets old value + its sum -- written with
    update_one under a max load factor of 4.0, so that the update's own insert() call cannot double the table (an LP table can be
    100 % full after its shrink quirk: 1.0 would not do)"""
    uk, inv = np.unique(ks, return_inverse=True)
    add = np.zeros(len(uk), dtype=np.uint64)
    np.add.at(add, inv, vs.astype(np.uint64) if vs is not None else np.uint64(1))
    old, found = o.find(uk)
    got = o.insert(ks, np.zeros(len(ks), dtype=np.uint32))
    o.set_max_load_factor(4.0)
    for k, a, ov, f in zip(uk.tolist(), add.tolist(), old.tolist(), found.tolist()):
        o.update_one(k, ((ov if f else 0) + a) & 0xFFFFFFFF)
    o.set_max_load_factor(mx)
    return got


runs = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    cls, kind = KINDS[int(rng.integers(0, 2))]
    hname, hid = HASHES[int(rng.integers(0, 3))]
    mn = float(rng.choice([0.1, 0.35, 0.4])); mx = float(rng.choice([0.5, 0.7, 0.8, 0.9, 0.95]))
    if big and mx > 0.9:
        mx = 0.9      # the reference LP insert itself goes quadratic with a million tombstones at load 0.95 (seed 700001): the CPU oracle, not the GPU, would stall
    cap0 = int(rng.choice([1, 128, 4096, 1 << 15]))
    usize = int(rng.choice([3_000, 60_000, 600_000])) if not big else int(rng.choice([2_000_000, 8_000_000]))
    g = cls(cap0, mn, mx, hash=hname, seed=43)
    o = O.OracleTable(kind, cap0, mn, mx, hid, 43)
    universe = W.splitmix64(np.arange(usize, dtype=np.uint64) + np.uint64(seed << 24))
    step = -1; op = -1
    try:
        for step in range(40 if not big else 14):
            op = int(rng.integers(0, 16))
            m = int(rng.choice([0, 1, 3, 50, 2000, 20_000, 150_000])) if not big else int(rng.choice([0, 5, 2000, 300_000, 1_500_000, 4_000_000]))
            if big and kind == 1 and op == 6:
                m = min(m, 300_000)      # a million tombstones make the REFERENCE's (hence the oracle's) later LP inserts quadratic (seed 950038: minutes per step)
            ks = universe[rng.integers(0, len(universe), m)]
            vs = rng.integers(0, 2**32, m, dtype=np.uint32)
            if verbose: print("step", step, "op", op, "m", m, "size", o.size(), "cap", o.capacity(), kind, hname, mn, mx, flush=True)
            if op <= 2:
                got = g.insert(dev(ks), dev(vs))
                if verbose: print("   gpu insert done:", got, flush=True)
                assert got == o.insert(ks, vs)
            elif op == 3:
                assert g.insert(ks, vs) == o.insert(ks, vs)                       # host buffers
            elif op == 4 and m <= 2000:
                g.update(ks, vs)
                for k, v in zip(ks.tolist(), vs.tolist()):
                    o.update_one(k, v)
            elif op == 5:
                if m: check_queries(g, o, np.concatenate([ks, universe[:100] ^ np.uint64(1 << 63)]))
            elif op == 6:
                assert g.erase(dev(ks)) == o.erase(ks)
            elif op == 7:
                for k in ks[:5]:
                    assert g.erase_one(int(k)) == o.erase_one(int(k))
            elif op == 8:
                r = int(rng.integers(0, usize)); g.reserve(r); o.reserve(r)
            elif op == 9:
                b = int(rng.choice([1024, 4096, 1 << 16, 1 << 18, 1 << 20]))
                if o.size() <= b * min(mx, 0.5):
                    g.rehash(b); o.rehash(b)
            elif op == 10 and m:                                                  # streamed insert in 1..5 feeds
                cuts = sorted(set([0, m] + [int(x) for x in rng.integers(0, m + 1, int(rng.integers(0, 5)))]))
                rep = bool(rng.integers(0, 2))                                    # repeatable: the speculative layout, fed again on KhRetry
                try:
                    g.insert_begin(m, repeatable=rep)
                    for a, b in zip(cuts[:-1], cuts[1:]):
                        g.insert_feed(dev(ks[a:b]), dev(vs[a:b]))
                    got = g.insert_end()
                except kh.KhRetry:
                    assert rep
                    g.insert_begin(m)
                    for a, b in zip(cuts[:-1], cuts[1:]):
                        g.insert_feed(dev(ks[a:b]), dev(vs[a:b]))
                    got = g.insert_end()
                assert got == o.insert(ks, vs)
            elif op == 11 and rng.random() < 0.3:
                g.clear(); o.clear()
            elif op == 13 and m <= 300_000:                                       # Reducer = std::plus, one call, with or without values
                use_v = bool(rng.integers(0, 2))
                assert g.insert_reduce_plus(dev(ks), dev(vs) if use_v else None) == oracle_plus(o, ks, vs if use_v else None, mx)
            elif op == 14 and 0 < m <= 300_000:                                   # the same streamed (repeatable or not), or aborted half-way
                cuts = sorted(set([0, m] + [int(x) for x in rng.integers(0, m + 1, int(rng.integers(0, 4)))]))
                rep = bool(rng.integers(0, 2)); abort = rng.random() < 0.25
                try:
                    g.insert_begin(m, reduce_plus=True, repeatable=rep)
                    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                        if abort and i == len(cuts) - 2:
                            break
                        g.insert_feed(dev(ks[a:b]))
                    got = None
                    if abort:
                        g.insert_abort()
                    else:
                        got = g.insert_end()
                except kh.KhRetry:
                    assert rep and not abort
                    g.insert_begin(m, reduce_plus=True)
                    for a, b in zip(cuts[:-1], cuts[1:]):
                        g.insert_feed(dev(ks[a:b]))
                    got = g.insert_end()
                if not abort:
                    assert got == oracle_plus(o, ks, None, mx)
            elif op == 15 and m:                                                  # erase of a batch with misses and repeats (streaming form / in place / small)
                e = np.concatenate([ks, universe[:50] ^ np.uint64(1 << 62), ks[: m // 3]])
                assert g.erase(dev(e)) == o.erase(e)
            elif op == 12:
                f = float(rng.choice([0.5, 0.7, 0.8, 0.9]))
                if f > mn: g.set_max_load_factor(f); o.set_max_load_factor(f); mx = f
            check_state(g, o, kind)
    except (AssertionError, Exception) as e:
        import traceback; traceback.print_exc()
        print("oracle probe_overflow:", o.probe_overflow())
        print("state: gpu size/cap", g.size(), g.capacity(), g.load_thresholds(), " oracle", o.size(), o.capacity(), o.min_load(), o.max_load(), "m", m)
        print("FAIL seed", seed, "step", step, "op", op, "kind", kind, hname, mn, mx, cap0, usize, repr(e)[:300], flush=True)
        sys.exit(1)
    g.close()
    runs += 1; seed += 1
    if big or runs % 100 == 0: print("seq", runs, "seed", seed - 1, "%.0f s left" % (t_end - time.time()), flush=True)
    if only_one: break
print("soak ok: %d sequences of 40 steps, seeds up to %d" % (runs, seed - 1))
