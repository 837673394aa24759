"""Build recipe for libkmerhash_amd.so (hipcc, gfx950 only, in-tree output)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "kmerhash_amd.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "kh_kernels.h"), os.path.join(HERE, "..", "include", "kmerhash_amd", "kh_hash.h"),
        os.path.join(HERE, "..", "include", "kmerhash_amd.h")]
# KH_LIB_SUFFIX: experiment builds next to the product (libkmerhash_amd<suffix>.so, built with KH_EXTRA_FLAGS) for A/B runs on one box
LIB = os.path.join(HERE, "libkmerhash_amd%s.so" % os.environ.get("KH_LIB_SUFFIX", ""))
RES = os.path.join(HERE, "kernel_resources.json")      # per-kernel registers / LDS / occupancy reported by the compiler


DIST_SRC = os.path.join(HERE, "csrc", "kmerhash_amd_dist.cpp")
DIST_LIB = os.path.join(HERE, "libkmerhash_amd_dist.so")
DIST_DEPS = [DIST_SRC, os.path.join(HERE, "..", "include", "kmerhash_amd_dist.h"), os.path.join(HERE, "..", "include", "kmerhash_amd.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build_library(force=False, verbose=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    if not force and not stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-comment", "-Rpass-analysis=kernel-resource-usage", "-o", LIB, SRC] + os.environ.get("KH_EXTRA_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True)
    other = [l for l in p.stderr.splitlines() if "kernel-resource-usage" not in l]
    if p.returncode != 0 or verbose:
        print("\n".join(other))
    if p.returncode != 0:
        raise subprocess.CalledProcessError(p.returncode, cmd)
    _write_resources(p.stderr)
    return LIB


def build_dist_library(force=False, verbose=False):
    """libkmerhash_amd_dist.so: the sharded table over RCCL (host-only C++ on top of the C-ABI library and librccl)."""
    build_library()
    if not force and os.path.exists(DIST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(DIST_LIB) for d in DIST_DEPS + [LIB] if os.path.exists(d)):
        return DIST_LIB
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-value", "-Wno-unused-result", "-o", DIST_LIB, DIST_SRC, "-L" + HERE, "-lkmerhash_amd", "-lrccl",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True)
    if p.returncode != 0 or verbose:
        print(p.stderr)
    if p.returncode != 0:
        raise subprocess.CalledProcessError(p.returncode, cmd)
    return DIST_LIB


def _write_resources(remarks):
    """kernel_resources.json: {mangled kernel name: {"VGPRs":..,"LDS":..,"Occupancy":..,"Scratch":..}} from the compiler's
    -Rpass-analysis=kernel-resource-usage remarks (tests/test_kernel_resources.py guards the occupancy-critical kernels)"""
    import json
    import re
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            k = {"VGPRs": "VGPRs", "TotalSGPRs": "SGPRs", "ScratchSize [bytes/lane]": "Scratch", "Occupancy [waves/SIMD]": "Occupancy",
                 "LDS Size [bytes/block]": "LDS", "VGPRs Spill": "VGPRSpill", "SGPRs Spill": "SGPRSpill"}.get(k.strip())
            if k:
                cur[k] = int(v.strip())
    with open(RES, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_dist_library(force=True, verbose=True))
