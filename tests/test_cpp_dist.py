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


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_dist_library_in_process_one_rcc_rank_forced_collectives():
    """libkmerhash_amd_dist.so loaded INTO this process (ctypes): one RCCL rank with KHD_OPT_FORCE_COLLECTIVES runs the count exchange, the
    grouped all-to-all-v of every piece, the votes and the pipelined find as self-exchanges; results equal the plain table's"""
    import ctypes as C
    import numpy as np
    import torch
    import kmerhash_amd as kh
    from kmerhash_amd import workloads as W
    lib = _build()
    L = C.CDLL(lib)
    vp, u64 = C.c_void_p, C.c_uint64
    L.khd_unique_id.argtypes = [vp]
    L.khd_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64, u64, C.c_float, C.c_float, C.c_int, u64]
    L.khd_set_option.argtypes = [vp, C.c_int, C.c_longlong]
    L.khd_insert.argtypes = [vp, vp, vp, u64, C.c_int, C.c_int, C.POINTER(u64)]
    L.khd_find.argtypes = [vp, vp, u64, vp, vp, vp]
    L.khd_erase.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.khd_size.argtypes = [vp, C.POINTER(u64)]
    L.khd_synchronize.argtypes = [vp]
    L.khd_destroy.argtypes = [vp]
    L.khd_last_error.argtypes = [vp]
    L.khd_last_error.restype = C.c_char_p
    ident = C.create_string_buffer(128)
    assert L.khd_unique_id(ident) == 0
    m = vp()
    assert L.khd_create(C.byref(m), ident, 1, 0, 0, 0, 1, 43, 128, 0.35, 0.8, 1, 9876543) == 0
    assert L.khd_set_option(m, 1, 1) == 0 and L.khd_set_option(m, 2, 2) == 0          # force collectives; 2 query pieces
    n = 600_000
    keys = W.distinct_u64(n, seed=61)
    keys[-2000:] = keys[:2000]
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    dv = torch.arange(n, dtype=torch.int32, device="cuda")
    ni = u64()
    assert L.khd_insert(m, dk.data_ptr(), dv.data_ptr(), n, 3, 0, C.byref(ni)) == 0, L.khd_last_error(m)
    plain = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    assert plain.insert(dk, dv) == ni.value == n - 2000
    gs = u64()
    assert L.khd_size(m, C.byref(gs)) == 0 and gs.value == n - 2000
    q = torch.cat([dk[:50_000], torch.from_numpy(W.distinct_u64(20_000, seed=62).view(np.int64)).cuda()])
    ok = torch.empty_like(q)
    ov = torch.empty(q.numel(), dtype=torch.int32, device="cuda")
    of = torch.empty(q.numel(), dtype=torch.uint8, device="cuda")
    assert L.khd_find(m, q.data_ptr(), q.numel(), ok.data_ptr(), ov.data_ptr(), of.data_ptr()) == 0
    assert L.khd_synchronize(m) == 0
    pv, pf = plain.find_values(q)
    assert torch.equal(ok, q) and torch.equal(of, pf) and torch.equal(ov[of == 1], pv[pf == 1])
    ne = u64()
    assert L.khd_erase(m, q.data_ptr(), q.numel(), C.byref(ne)) == 0 and ne.value == plain.erase(q)
    assert L.khd_destroy(m) == 0
    plain.close()
