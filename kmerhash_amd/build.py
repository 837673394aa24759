"""Build recipe for libkmerhash_amd.so (hipcc, gfx950 only, in-tree output)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "kmerhash_amd.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "kh_kernels.h"), os.path.join(HERE, "csrc", "kh_hash.h"),
        os.path.join(HERE, "..", "include", "kmerhash_amd.h")]
LIB = os.path.join(HERE, "libkmerhash_amd.so")


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
           "-Wno-unused-value", "-Wno-comment", "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
