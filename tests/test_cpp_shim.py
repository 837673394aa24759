"""The C++11 template shim (include/kmerhash_amd/hashmap.hpp + the drop-in include/kmerhash/*.hpp paths):
CPU: it compiles with plain g++ -std=c++11 (the reference's language level) and links against the C-ABI
library only.  GPU: the reference-style differential test binary passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_shim.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "_test_shim")
BENCH_SRC = os.path.join(ROOT, "benchmark", "benchmark_hashtables.cpp")
BENCH_BIN = os.path.join(ROOT, "benchmark", "_benchmark_hashtables")


def _compile(src, out):
    from kmerhash_amd.build import build_library
    build_library()
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-Wno-comment", "-I" + os.path.join(ROOT, "include"), src,
           "-L" + os.path.join(ROOT, "kmerhash_amd"), "-lkmerhash_amd",
           "-Wl,-rpath," + os.path.join(ROOT, "kmerhash_amd"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_shim_compiles_as_cxx11_against_the_c_abi_only():
    _compile(SRC, BIN)
    _compile(BENCH_SRC, BENCH_BIN)
    # the binaries depend on the C-ABI library, not on a HIP toolchain at build time
    out = subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    assert "libkmerhash_amd.so" in out


@pytest.mark.gpu
def test_shim_differential_on_gpu():
    _compile(SRC, BIN)
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all shim tests passed" in r.stdout


@pytest.mark.gpu
def test_cpp_benchmark_driver_small():
    _compile(BENCH_SRC, BENCH_BIN)
    for m in ("robinhood", "linearprobe"):
        r = subprocess.run([BENCH_BIN, "-m", m, "-N", "200000", "-Q", "10", "-R", "10"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "count2" in r.stdout


def test_cpp_benchmark_driver_flag_errors():
    """flag surface of the reference driver (BenchmarkHashTables.cpp:1439-1486): what this path does not have is refused
    with a message, before any GPU work (runs without a GPU)"""
    _compile(BENCH_SRC, BENCH_BIN)
    for argv, msg in ((["-A", "dna5"], "only dna"), (["-I", "bogus"], "unknown insert mode"),
                      (["-m", "linearprobe", "-I", "sort"], "robinhood map only"), (["--no-such-flag"], "usage")):
        r = subprocess.run([BENCH_BIN] + argv + ["-N", "10"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and msg in r.stderr, (argv, r.stderr)


@pytest.mark.gpu
def test_cpp_benchmark_driver_reference_flags():
    _compile(BENCH_SRC, BENCH_BIN)
    for extra in (["-I", "iter"], ["-I", "integrated", "-c"], ["-I", "shuffle", "-f", "--measured_op", "find", "--insert_prefetch", "8"]):
        r = subprocess.run([BENCH_BIN, "-m", "robinhood", "-N", "100000", "-A", "dna"] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "SELF-CHECK FAILED" not in r.stdout and "count2" in r.stdout
    assert "*find" in r.stdout
