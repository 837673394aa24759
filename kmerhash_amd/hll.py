"""fsc::hyperloglog64<T, Hash, precision> on the GPU (reference hyperloglog64.hpp:142-475) over the C-ABI."""
import ctypes as C

import numpy as np

from . import _capi as K
from .table import _Buf, _hash_id, torch


class hyperloglog64:
    def __init__(self, precision=12, ignore_msb=0, hash="murmur3avx64", seed=43, device=0):
        self._L = K.lib()
        self._h = C.c_void_p()
        self.precision = precision
        self.device = device
        st = self._L.kh_hll_create(C.byref(self._h), precision, ignore_msb, _hash_id(hash), seed, device)
        if st != K.KH_OK:
            self._h = C.c_void_p()
            raise K.KhError(st, "kh_hll_create failed")

    # est_error_rate (hyperloglog64.hpp:262): 1.04 / 2^(precision/2)
    @property
    def est_error_rate(self):
        return 1.04 / float(1 << (self.precision >> 1))

    def _chk(self, st, what):
        if st != K.KH_OK:
            raise K.KhError(st, what)

    def _stream(self, b):
        if b.where == K.KH_MEM_DEVICE and torch is not None:
            self._L.kh_hll_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def update(self, keys):
        b = _Buf(keys, np.uint64, 8)
        self._stream(b)
        self._chk(self._L.kh_hll_update(self._h, b.ptr, b.n, b.where), "kh_hll_update")

    def update_via_hashval(self, hashes):
        b = _Buf(hashes, np.uint64, 8)
        self._stream(b)
        self._chk(self._L.kh_hll_update_via_hashval(self._h, b.ptr, b.n, b.where), "kh_hll_update_via_hashval")

    def merge(self, other):
        self._chk(self._L.kh_hll_merge(self._h, other._h), "kh_hll_merge")

    def clear(self):
        self._chk(self._L.kh_hll_clear(self._h), "kh_hll_clear")

    def registers(self):
        out = np.zeros(1 << self.precision, dtype=np.uint8)
        self._chk(self._L.kh_hll_registers(self._h, out.ctypes.data), "kh_hll_registers")
        return out

    def estimate(self):
        d = C.c_double()
        self._chk(self._L.kh_hll_estimate(self._h, C.byref(d)), "kh_hll_estimate")
        return d.value

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.kh_hll_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
