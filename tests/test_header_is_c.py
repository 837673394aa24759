"""CPU: include/kmerhash_amd.h is a plain C header (C99) -- any FFI that speaks C can bind it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_compiles_as_c99_and_links(tmp_path):
    from kmerhash_amd.build import build_library
    build_library()
    src = tmp_path / "use.c"
    src.write_text('#include "kmerhash_amd.h"\n#include <stdio.h>\n'
                   'int main(void) {\n  kh_table* t = 0; uint64_t n = 0;\n'
                   '  kh_status s = kh_create(&t, KH_KIND_ROBINHOOD, 8, 4, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, 0);\n'
                   '  if (s == KH_OK) { kh_size(t, &n); kh_destroy(t); }\n'
                   '  printf("%s status=%d\\n", kh_version(), (int)s);\n  return 0;\n}\n')
    exe = tmp_path / "use"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), str(src),
                        "-L" + os.path.join(ROOT, "kmerhash_amd"), "-lkmerhash_amd",
                        "-Wl,-rpath," + os.path.join(ROOT, "kmerhash_amd"), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "gfx950" in r.stdout     # status is KH_ERR_HIP (5) without a GPU, KH_OK with one


def test_every_environment_switch_is_documented():
    """every KH_* / KHD_* variable the libraries read (getenv in kmerhash_amd/csrc) is listed in INTEGRATION.md's table of switches"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in ("kmerhash_amd/csrc/kmerhash_amd.hip", "kmerhash_amd/csrc/kmerhash_amd_dist.cpp", "kmerhash_amd/csrc/kh_kernels.h"):
        names |= set(re.findall(r'getenv\("(KHD?_[A-Z0-9_]+)"\)', open(os.path.join(root, f)).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert len(names) >= 15 and not missing, missing
