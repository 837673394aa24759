"""The C++ sharded map (include/kmerhash_amd_dist.h -> libkmerhash_amd_dist.so: the reference's MPI layer replaced by RCCL,
host side in C++ like the reference's).  CPU: the library builds, exports every declared symbol and links against the C-ABI
library and librccl only.  GPU: tests/cpp/test_dist.cpp -- one RCCL rank, then 4 / 3 / 2 ranks as threads on the one GPU over
the in-process transport, compared with a single table that receives the pairs in (piece, source rank, position) order."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_dist.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "_test_dist")


def _build():
    from kmerhash_amd import build as B
    lib = B.build_dist_library()
    cmd = [B.hipcc(), "-O2", "-std=c++17", "-Wall", "-Wno-unused-value", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), SRC,
           "-L" + os.path.join(ROOT, "kmerhash_amd"), "-lkmerhash_amd_dist", "-lkmerhash_amd", "-Wl,-rpath," + os.path.join(ROOT, "kmerhash_amd"), "-o", BIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return lib


def test_dist_library_builds_and_exports_the_header():
    lib = _build()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kmerhash_amd_dist.h")).read(), flags=re.S)
    decl = sorted(set(re.findall(r"\b(khd_[a-z0-9_]+)\s*\(", txt)))
    assert len(decl) >= 14
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (khd_[a-z0-9_]+)", out))
    assert exported == set(decl), (sorted(exported ^ set(decl)))
    needed = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
    assert "libkmerhash_amd.so" in needed and "librccl" in needed


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_dist_map_on_gpu():
    _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "all dist tests passed" in r.stdout


DRV_SRC = os.path.join(ROOT, "benchmark", "benchmark_dist_hashtables.cpp")
DRV_BIN = os.path.join(ROOT, "benchmark", "_benchmark_dist_hashtables")


def _build_driver():
    from kmerhash_amd import build as B
    B.build_dist_library()
    cmd = [B.hipcc(), "-O2", "-std=c++17", "-Wall", "-Wno-unused-value", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), DRV_SRC,
           "-L" + os.path.join(ROOT, "kmerhash_amd"), "-lkmerhash_amd_dist", "-lkmerhash_amd", "-Wl,-rpath," + os.path.join(ROOT, "kmerhash_amd"), "-o", DRV_BIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_dist_driver_flag_surface():
    """benchmarkHT's flags (BenchmarkDistHashTables.cpp:787-802): what is not part of this path is refused with a message before any
    GPU work (runs without a GPU)"""
    _build_driver()
    for argv, msg in ((["--hybrid"], "not part of this path"), (["--no-such-flag"], "usage"), (["-m", "sparsehash"], "unknown map type"),
                      (["--missing-frac", "2"], "out of range"), (["--nranks", "2", "--rank", "0"], "needs --id-file")):
        r = subprocess.run([DRV_BIN] + argv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and msg in r.stderr, (argv, r.stderr)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_dist_driver_on_gpu(tmp_path):
    _build_driver()
    # four ranks as threads on the one GPU, the reference's flag set; then one RCCL rank replaying a dumped key file (-F)
    r = subprocess.run([DRV_BIN, "--local-ranks", "4", "-C", "4000000", "-R", "4", "--missing-frac", "0.25", "--max_load", "0.7", "--min_load", "0.3",
                        "--insert_prefetch", "8", "--query_prefetch", "16"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SELF-CHECK FAILED" not in r.stdout + r.stderr, r.stdout + r.stderr
    assert "ranks 4" in r.stdout and "erase" in r.stdout
    import numpy as np
    from kmerhash_amd import io_utils
    keys = (np.arange(200_000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(2)
    fn = str(tmp_path / "keys.bin")
    io_utils.serialize_keys(keys, fn)
    r = subprocess.run([DRV_BIN, "--nranks", "1", "--rank", "0", "-F", fn, "-C", "1000000", "-R", "5", "-m", "linearprobe"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SELF-CHECK FAILED" not in r.stdout + r.stderr, r.stdout + r.stderr
    m = re.search(r"global size after insert (\d+)", r.stdout)
    assert m and 190_000 < int(m.group(1)) <= 200_000, r.stdout        # 10^6 draws from the file's 200000 keys
