"""CPU: the C-ABI shared library loads and exports every symbol include/kmerhash_amd.h declares; without a
GPU the product fails loudly instead of falling back to anything."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from kmerhash_amd.build import build_library
    build_library()
    from kmerhash_amd import _capi
    return _capi


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "kmerhash_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol(capi):
    L = capi.lib()
    decl = declared_symbols()
    assert len(decl) >= 30
    for s in decl:
        assert hasattr(L, s), "library does not export %s" % s
    assert sorted(capi.SYMBOLS) == decl
    assert b"gfx950" in L.kh_version()


def test_no_cpu_fallback_without_gpu(capi):
    import ctypes as C
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    h = C.c_void_p()
    st = capi.lib().kh_create(C.byref(h), 0, 8, 4, 1, 43, 128, 0.35, 0.8, 0)
    assert st == capi.KH_ERR_HIP and not h.value
    import kmerhash_amd as kh
    with pytest.raises(kh.KhError):
        kh.hashmap_robinhood_doubling()


def test_unsupported_widths_are_refused(capi):
    import ctypes as C
    h = C.c_void_p()
    assert capi.lib().kh_create(C.byref(h), 0, 16, 4, 1, 43, 128, 0.35, 0.8, 0) == capi.KH_ERR_UNSUPPORTED
    assert capi.lib().kh_create(C.byref(h), 0, 8, 8, 1, 43, 128, 0.35, 0.8, 0) == capi.KH_ERR_UNSUPPORTED


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under kmerhash_amd/, include/, benchmark/ or scripts/ may import, link or call it
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do)"""
    bad = []
    for base in ("kmerhash_amd", "include", "benchmark", "scripts"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".sh")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"oracle_py|kh_oracle|libkh_oracle|libref_lp|from oracle|import oracle", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
