"""fsc::hyperloglog64<T, Hash, precision> on the GPU (reference hyperloglog64.hpp:142-475) over the C-ABI."""
import ctypes as C

import numpy as np

from . import _capi as K
from .table import _Buf, _hash_id, torch


def estimate_from_registers(registers, precision=12):
    """internal_estimate (hyperloglog64.hpp:201-236) on host registers (uint8[2^precision]); needs no GPU"""
    regs = np.ascontiguousarray(registers, dtype=np.uint8)
    if regs.size != (1 << precision):
        raise ValueError("expected %d registers" % (1 << precision))
    d = C.c_double()
    st = K.lib().kh_hll_estimate_registers(regs.ctypes.data, precision, C.byref(d))
    if st != K.KH_OK:
        raise K.KhError(st, "kh_hll_estimate_registers")
    return d.value


def estimate_global(hll, group=None):
    """estimate_global (hyperloglog64.hpp:482-484): the estimate over the registers of ALL ranks, merged with an all-reduce(max)
    (merge_distributed :477-479; RCCL when the process group's backend is nccl).  `hll`: anything with registers() -> uint8 array and
    .precision.  Collective."""
    import torch.distributed as dist
    regs = np.ascontiguousarray(hll.registers(), dtype=np.uint8)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.from_numpy(regs.copy())
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        regs = t.cpu().numpy()
    return estimate_from_registers(regs, hll.precision)


def estimate_average_per_rank(hll, group=None):
    """estimate_average_per_rank (hyperloglog64.hpp:487-489): the global estimate divided by the number of ranks (hashed keys spread
    evenly): what a rank's local table must hold"""
    import torch.distributed as dist
    p = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    return estimate_global(hll, group) / float(p)


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
