"""
ORACLE (test infrastructure -- NOT product code): ctypes binding of
oracle/_build/libfus_oracle*.so (C restatement of numba-cpu/operators.py etc.,
see fus_oracle_impl.h for the per-function citations).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")

_c_i64, _c_int, _vp = C.c_int64, C.c_int, C.c_void_p


def build(native: bool = False):
    """Compile the oracle (gcc). ``native=True`` rebuilds with -march=native on
    the machine that will time it."""
    target = "native" if native else "_build/libfus_oracle.so"
    subprocess.run(["make", "-C", HERE, target], check=True, capture_output=True)


def _ptr(a):
    return a.ctypes.data_as(_vp)


class OracleLib:
    def __init__(self, native: bool = False):
        name = "libfus_oracle_native.so" if native else "libfus_oracle.so"
        path = os.path.join(BUILD, name)
        if not os.path.exists(path):
            build(native)
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.oracle_max_threads.restype = _c_int
        for suf, ct in (("f64", C.c_double), ("f32", C.c_float)):
            f = getattr(self.lib, f"oracle_stiffness_apply_{suf}")
            f.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_i64]
            f.restype = _c_int
            f = getattr(self.lib, f"oracle_stiffness_apply_omp_{suf}")
            f.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_i64, _c_int]
            f.restype = _c_int
            f = getattr(self.lib, f"oracle_mass_apply_{suf}")
            f.argtypes = [_vp, _vp, _vp, _vp, _vp, _c_int, _c_i64]
            f.restype = _c_int
            getattr(self.lib, f"oracle_axpy_{suf}").argtypes = [ct, _vp, _vp, _c_i64]
            getattr(self.lib, f"oracle_copy_{suf}").argtypes = [_vp, _vp, _c_i64]
            getattr(self.lib, f"oracle_fill_{suf}").argtypes = [ct, _vp, _c_i64]
            getattr(self.lib, f"oracle_pointwise_divide_{suf}").argtypes = [_vp, _vp, _vp, _c_i64]
            getattr(self.lib, f"oracle_square_{suf}").argtypes = [_vp, _vp, _c_i64]
            for nm in ("pack", "unpack_rev", "unpack_fwd"):
                getattr(self.lib, f"oracle_{nm}_{suf}").argtypes = [_vp, _vp, _vp, _c_i64]
        self.lib.oracle_contract_f64.argtypes = [_c_int] * 5 + [_vp] * 3
        self.lib.oracle_transpose_f64.argtypes = [_c_int] * 6 + [_vp] * 2

    @staticmethod
    def _suf(a):
        if a.dtype == np.float64:
            return "f64"
        if a.dtype == np.float32:
            return "f32"
        raise TypeError(f"unsupported dtype {a.dtype}")

    @staticmethod
    def _chk(*arrs):
        for a in arrs:
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError("oracle expects C-contiguous arrays")

    def max_threads(self) -> int:
        return int(self.lib.oracle_max_threads())

    def stiffness_apply(self, P, dphi, x, cell_constants, y, G, dofmap, threads: int = 1):
        dphi = np.ascontiguousarray(dphi, dtype=x.dtype).reshape(-1)
        self._chk(x, cell_constants, y, G, dofmap)
        assert dofmap.dtype == np.int32 and dofmap.shape[1] == (P + 1) ** 3
        suf = self._suf(x)
        nc = dofmap.shape[0]
        if threads == 1:
            rc = getattr(self.lib, f"oracle_stiffness_apply_{suf}")(
                _ptr(x), _ptr(cell_constants), _ptr(y), _ptr(G), _ptr(dofmap), _ptr(dphi), P, nc)
        else:
            rc = getattr(self.lib, f"oracle_stiffness_apply_omp_{suf}")(
                _ptr(x), _ptr(cell_constants), _ptr(y), _ptr(G), _ptr(dofmap), _ptr(dphi), P, nc, threads)
        if rc != 0:
            raise RuntimeError(f"oracle stiffness failed rc={rc}")

    def mass_apply(self, x, entity_constants, y, entity_detJ, entity_dofmap):
        self._chk(x, entity_constants, y, entity_detJ, entity_dofmap)
        assert entity_dofmap.dtype == np.int32
        ne, N = entity_dofmap.shape
        rc = getattr(self.lib, f"oracle_mass_apply_{self._suf(x)}")(
            _ptr(x), _ptr(entity_constants), _ptr(y), _ptr(entity_detJ), _ptr(entity_dofmap), N, ne)
        if rc != 0:
            raise RuntimeError(f"oracle mass failed rc={rc}")

    def axpy(self, alpha, x, y, n=None):
        getattr(self.lib, f"oracle_axpy_{self._suf(x)}")(float(alpha), _ptr(x), _ptr(y), y.size if n is None else n)

    def copy(self, a, b):
        getattr(self.lib, f"oracle_copy_{self._suf(a)}")(_ptr(a), _ptr(b), a.size)

    def fill(self, alpha, x):
        getattr(self.lib, f"oracle_fill_{self._suf(x)}")(float(alpha), _ptr(x), x.size)

    def pointwise_divide(self, a, b, c):
        getattr(self.lib, f"oracle_pointwise_divide_{self._suf(a)}")(_ptr(a), _ptr(b), _ptr(c), c.size)

    def square(self, a, b):
        getattr(self.lib, f"oracle_square_{self._suf(a)}")(_ptr(a), _ptr(b), a.size)

    def pack(self, in_, out_, index):
        index = np.ascontiguousarray(index, dtype=np.int64)
        getattr(self.lib, f"oracle_pack_{self._suf(in_)}")(_ptr(in_), _ptr(out_), _ptr(index), index.size)

    def unpack_rev(self, in_, out_, index):
        index = np.ascontiguousarray(index, dtype=np.int64)
        getattr(self.lib, f"oracle_unpack_rev_{self._suf(in_)}")(_ptr(in_), _ptr(out_), _ptr(index), index.size)

    def unpack_fwd(self, in_, out_, index):
        index = np.ascontiguousarray(index, dtype=np.int64)
        getattr(self.lib, f"oracle_unpack_fwd_{self._suf(in_)}")(_ptr(in_), _ptr(out_), _ptr(index), index.size)
