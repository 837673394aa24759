"""CPU: bench.py's launcher plumbing: `--gpus N` starts N rank processes itself, refuses loudly when fewer
devices are visible, and a rank refuses a WORLD_SIZE that differs from --gpus.  No GPU is touched here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_follow_the_metric():
    a = bench.parse([])
    assert (a.gpus, a.keys, a.queries) == (1, 107_374_184, 10_000_000)       # load exactly 0.800 at capacity 2^27
    assert bench.KEYS_LOAD_080 == int(__import__("numpy").float32(2 ** 27) * __import__("numpy").float32(0.8))
    b = bench.parse(["--gpus", "8"])
    assert (b.gpus, b.keys) == (8, 100_000_000)
    assert bench.parse(["--gpus", "2", "--keys", "1000"]).keys == 1000


def test_child_env_and_argv():
    e = bench.child_env(3, 8, 29511, base={"PATH": "/bin", "WORLD_SIZE": "1"})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "29511")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin"
    argv = bench.child_argv(["--gpus", "8", "--steps", "2"])
    assert argv[0] == sys.executable and argv[1].endswith("bench.py")
    assert argv[2:] == ["--gpus", "8", "--steps", "2", "--no-cpu-baseline"]
    assert bench.child_argv(["--no-cpu-baseline"]).count("--no-cpu-baseline") == 1


def _run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          universal_newlines=True, timeout=300)


def test_more_gpus_than_visible_fails_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n), "--no-cpu-baseline", "--keys", "1000", "--queries", "10"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing to run fewer ranks" in r.stderr and r.stdout.strip() == ""


def test_rank_refuses_world_size_mismatch():
    r = _run(["--gpus", "4", "--no-cpu-baseline"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2 but --gpus 4" in r.stderr and r.stdout.strip() == ""


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_rank_control_flow_rehearsal_on_one_gpu():
    """the N > 1 path of bench.py end to end -- launcher, rank processes, barriers, pipelined sharded insert, sharded find, the result
    reductions and assertions -- with both ranks on GPU 0 over gloo (KH_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device)"""
    import json
    r = _run(["--gpus", "2", "--keys", "3000000", "--queries", "300000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
             {"KH_BENCH_REHEARSAL": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["config"]["exchange_pieces"] == 4
    assert set(d["phases_ms_per_step_rank0"]) >= {"count_pass", "permute", "exchange", "feed", "build"}


def test_bench_generators_and_host_cores_on_cpu():
    """the device-side workload generators of bench.py's informational legs (run here on CPU tensors) keep their shapes: W1 = 62-bit keys
    with mean multiplicity 5.5 and a permutation as values, W3 = distinct 62-bit keys x mult; the torch splitmix64 equals the numpy one"""
    import numpy as np
    import torch
    import bench
    from kmerhash_amd import workloads as W
    x = np.arange(2000, dtype=np.uint64) * np.uint64(0x123456789ABCDEF1)
    assert np.array_equal(W.splitmix64(x), bench._splitmix64_t(torch.from_numpy(x.view(np.int64))).numpy().view(np.uint64))
    k, v = bench.gpu_w1(50_000, torch.device("cpu"))
    assert k.numel() == 50_000 and int(k.min()) >= 0 and int(k.max()) < (1 << 62)
    assert 4.5 < 50_000 / torch.unique(k).numel() < 6.5 and sorted(v.tolist()) == list(range(50_000))
    base, k3, v3 = bench.gpu_w3(5_000, torch.device("cpu"))
    assert torch.unique(base).numel() == 5_000 and k3.numel() == 25_000 and torch.unique(k3).numel() == 5_000 and int(base.max()) < (1 << 62)
    hc = bench.host_cores()
    assert hc["logical"] >= 1 and hc["affinity"] >= 1 and (hc["physical"] is None or 1 <= hc["physical"] <= hc["logical"])
    s = bench._stat([1.0, 3.0, 2.0], 1_000_000, 50)
    assert s["ms_median"] == 2.0 and s["ms_min"] == 1.0 and abs(s["ops_per_s"] - 5e8) < 1 and abs(s["frac"] - 5e8 * 50 / 8e12) < 1e-12
