"""Deterministic synthetic k-mer key streams.

The reference's benchmark generator (BenchmarkHashTables.cpp:182-227) uses glibc rand()/random_shuffle,
which is not portable; as SURVEY §8d prescribes, every stream here comes from splitmix64 so that the
CPU oracle and the GPU table see byte-identical inputs on any host.  All functions are numpy
(vectorised) and need no GPU.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """splitmix64 output function applied element-wise to a uint64 array of states (bijective)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(seed, start, n):
    """n successive splitmix64 outputs of the generator seeded with `seed`, starting at draw `start`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        st = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15) - np.uint64(0x9E3779B97F4A7C15)
    return splitmix64(st)


def distinct_u64(n, seed=1, start=0):
    """W2: n DISTINCT uniform 64-bit keys (splitmix64 of a counter is a bijection)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(start, start + n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
    return splitmix64(ctr)


def shuffle_perm(n, seed):
    """A deterministic permutation of range(n) (argsort of splitmix64 keys; ties impossible for n < 2^64)."""
    return np.argsort(_stream(seed ^ 0x5bd1e995, 0, n), kind="stable")


def w1_benchmark_hashtables(n_pairs, seed=23, repeats=10, bits=62):
    """W1: the benchmark_hashtables shape (BenchmarkHashTables.cpp:192-223): draw a key (masked to `bits`
    bits = 31-mer sanitize), emit (key, i), then `draw % repeats` more copies (key, ++i); shuffle.
    Returns (keys u64[n_pairs], vals u32[n_pairs])."""
    # generate enough base keys: mean multiplicity (repeats+1)/2
    est = int(n_pairs / ((repeats + 1) / 2.0) * 1.1) + 16
    while True:
        base = _stream(seed, 0, est) & np.uint64((1 << bits) - 1)
        freq = (_stream(seed + 1, 0, est) % np.uint64(repeats)).astype(np.int64) + 1
        csum = np.cumsum(freq)
        if csum[-1] >= n_pairs:
            break
        est *= 2
    m = int(np.searchsorted(csum, n_pairs, side="left")) + 1
    keys = np.repeat(base[:m], freq[:m])[:n_pairs]
    vals = np.arange(n_pairs, dtype=np.uint32)
    p = shuffle_perm(n_pairs, seed + 2)
    return keys[p].copy(), vals[p].copy()


def w3_kmers_5x(n_distinct, mult=5, seed=3, bits=62):
    """W3: n_distinct distinct `bits`-bit keys (2-bit packed 31-mers), each exactly `mult` times, shuffled."""
    # distinctness must survive the `bits`-bit mask, so mix a counter with maps that are bijections on
    # `bits` bits: multiplication by an odd constant mod 2^bits and xorshift-right.
    with np.errstate(over="ignore"):
        ctr = np.arange(n_distinct, dtype=np.uint64) + np.uint64(seed * 1000003)
        base = (ctr * np.uint64(0x9E3779B97F4A7C15)) & np.uint64((1 << bits) - 1)
        base ^= base >> np.uint64(31)          # xorshift is a bijection on `bits` bits when shift < bits
        base = (base * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64((1 << bits) - 1)
        base ^= base >> np.uint64(29)
    keys = np.repeat(base, mult)
    n = len(keys)
    vals = np.arange(n, dtype=np.uint32)
    p = shuffle_perm(n, seed + 7)
    return keys[p].copy(), vals[p].copy()


def queries_hits_and_misses(inserted_keys, n_q, miss_fraction=0.5, seed=11):
    """n_q query keys: hits drawn from the inserted stream's first keys, misses from a fresh splitmix range."""
    n_miss = int(n_q * miss_fraction)
    n_hit = n_q - n_miss
    hits = inserted_keys[:n_hit]
    misses = distinct_u64(n_miss, seed=seed + 977)
    q = np.concatenate([hits, misses])
    return q[shuffle_perm(len(q), seed)].copy()
