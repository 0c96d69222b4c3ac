"""
ctypes binding of ``csrc/libfusgpu.so`` (C ABI declared in include/fus_gpu.h).

There is no CPU fallback: if the HIP library has not been built, or no GPU is
visible when a kernel is requested, this module raises.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FUS_LIB_PATH: load another build of the same ABI (A/B runs of two builds on one device)
LIB_PATH = os.environ.get("FUS_LIB_PATH") or os.path.join(_HERE, "csrc", "libfusgpu.so")

_i64, _int, _vp = C.c_int64, C.c_int, C.c_void_p

# every symbol include/fus_gpu.h declares -> (argtypes); restype is int unless noted
_SUFFIXES = (("f64", C.c_double), ("f32", C.c_float))


def _signatures():
    sig = {
        "fus_abi_version": [],
        "fus_device_info": [_int, C.c_char_p, C.POINTER(_int), C.POINTER(_i64), C.POINTER(_int)],
        "fus_set_tuning": [_int, _int],
        "fus_get_tuning": [_int],
        "fus_stiffness_plan_bytes": [_int, _i64],
        "fus_plan_entities_per_batch": [_int],
        "fus_plan_bytes": [_int, _int, _i64],
        "fus_plan_build": [_vp, _int, _int, _i64, _vp, _i64, _vp],
        "fus_plan_build_ordered": [_vp, _vp, _int, _int, _i64, _vp, _i64, _vp],
        "fus_plan_release": [_vp],
        "fus_plan_encoding": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int), C.POINTER(_int)],
        "fus_plan_mark_exclusive": [_vp, _int, _int, _i64, _vp, _i64, _vp],
        "fus_mass_gather_plan_bytes": [_int, _i64, _i64],
        "fus_mass_gather_plan_build": [_vp, _int, _i64, _i64, _vp, _i64, _vp],
        "fus_mass_gather_plan_build_rows": [_vp, _int, _i64, _i64, _vp, _int, _vp, _i64, _vp],
        "fus_mass_gather_static_bytes": [_int, _i64, _int],
        "fus_mass_gather_static_build_f64": [_vp, _vp, _vp, _i64, _vp],
        "fus_mass_gather_static_build_f32": [_vp, _vp, _vp, _i64, _vp],
        "fus_mass_apply_gather_static_f64": [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp],
        "fus_mass_apply_gather_static_f32": [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp],
        "fus_mass_gather_plan_info": [_vp, _vp],
        "fus_stiffness_plan_build": [_vp, _int, _i64, _vp, _i64, _vp],
        # communicator + halo exchange (csrc/halo_comm.hpp)
        "fus_comm_unique_id": [_vp],
        "fus_comm_create": [_vp, _int, _int, C.POINTER(_vp)],
        "fus_comm_create_local": [_int, _int, _int, C.POINTER(_vp)],
        "fus_comm_create_peer": [_int, _int, C.POINTER(_vp)],
        "fus_halo_ipc_blob_bytes": [_vp],
        "fus_halo_ipc_export": [_vp, _vp],
        "fus_halo_ipc_connect": [_vp, _int, _vp],
        "fus_halo_ipc_status": [_vp, _vp],
        "fus_comm_rank": [_vp],
        "fus_comm_size": [_vp],
        "fus_comm_stream": [_vp],
        "fus_comm_last_error": [_vp],
        "fus_comm_fork": [_vp, _vp],
        "fus_comm_fork_lazy": [_vp, _vp],
        "fus_comm_fork_ex": [_vp, _vp, _int],
        "fus_comm_fork_flush": [_vp],
        "fus_comm_join": [_vp, _vp],
        "fus_comm_arm_join": [_vp],
        "fus_comm_health": [_vp, _vp],
        "fus_comm_health_detail": [_vp, _vp],
        "fus_comm_sync_timeouts": [_vp, _vp],
        "fus_comm_destroy": [_vp],
        "fus_halo_create": [_vp, _int, _i64, _i64, _int, _vp, _vp, _vp, _int, _vp, _vp, _vp, C.POINTER(_vp)],
        "fus_halo_is_direct": [_vp],
        "fus_halo_destroy": [_vp],
        "fus_halo_forward_begin": [_vp, _vp, _vp],
        "fus_halo_forward_end": [_vp, _vp, _vp],
        "fus_halo_reverse_begin": [_vp, _vp, _vp],
        "fus_halo_reverse_end": [_vp, _vp, _vp],
        "fus_halo_forward_begin_group": [_vp, _vp, _int, _vp],
        "fus_halo_reverse_begin_group": [_vp, _vp, _int, _vp],
        "fus_halo_forward": [_vp, _vp, _vp],
        "fus_halo_reverse": [_vp, _vp, _vp],
    }
    for suf, ct in _SUFFIXES:
        sig[f"fus_stiffness_apply_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]
        sig[f"fus_stiffness_apply_planned_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]
        sig[f"fus_stiffness_apply_planned_affine_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]
        sig[f"fus_stiffness_apply_planned_geom_{suf}"] = [_vp] * 9 + [_int, _i64, _vp]
        sig[f"fus_mass_apply_planned_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _int, _int, _i64, _vp]
        sig[f"fus_mass_apply_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]
        sig[f"fus_mass_apply_gather_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]
        sig[f"fus_facet_terms_{suf}"] = [_vp, _vp, ct, _vp, ct, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _int, _vp]
        sig[f"fus_facet_terms_dev_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _int, _vp]
        sig[f"fus_axpy_{suf}"] = [ct, _vp, _vp, _i64, _vp]
        sig[f"fus_scale_{suf}"] = [ct, _vp, _vp, _i64, _vp]
        sig[f"fus_copy_{suf}"] = [_vp, _vp, _i64, _vp]
        sig[f"fus_fill_{suf}"] = [ct, _vp, _i64, _vp]
        sig[f"fus_pointwise_divide_{suf}"] = [_vp, _vp, _vp, _i64, _vp]
        sig[f"fus_square_{suf}"] = [_vp, _vp, _i64, _vp]
        sig[f"fus_muladd_{suf}"] = [_vp, _vp, _vp, _i64, _vp]
        sig[f"fus_geometry_factors_{suf}"] = [_vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _vp]
        sig[f"fus_facet_jacobian_{suf}"] = [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp, _vp]
        sig[f"fus_westervelt_cell_apply_planned_{suf}"] = [_vp] * 12 + [_int, _i64, _vp]
        sig[f"fus_westervelt_cell_apply_planned_geom_{suf}"] = [_vp] * 14 + [_int, _i64, _vp]
        sig[f"fus_rk4_stage_nl_{suf}"] = [ct, ct, _int] + [_vp] * 9 + [_i64, _i64, _vp]
        sig[f"fus_rk4_stage_nl2_{suf}"] = [ct, ct, _int] + [_vp] * 10 + [ct, _vp, _i64, _i64, _vp]
        sig[f"fus_rk4_stage_{suf}"] = [ct, ct, _int] + [_vp] * 8 + [_i64, _i64, _vp]
        sig[f"fus_pack_fwd_{suf}"] = [_vp, _vp, _vp, _i64, _vp]
        sig[f"fus_unpack_fwd_{suf}"] = [_vp, _vp, _vp, _i64, _i64, _vp]
        sig[f"fus_pack_rev_{suf}"] = [_vp, _vp, _vp, _i64, _i64, _vp]
        sig[f"fus_unpack_rev_{suf}"] = [_vp, _vp, _vp, _i64, _vp]
    return sig


SIGNATURES = _signatures()

TUNE_STIFFNESS_VARIANT = 1
TUNE_XCD_REMAP = 2
TUNE_MASS_VARIANT = 3
TUNE_PLAN_VARIANT = 4
TUNE_PLAN_RUNS = 5
TUNE_VECTOR_STREAM = 6

ABI_VERSION = 3  # include/fus_gpu.h FUS_ABI_VERSION

_lib = None


class FusGpuError(RuntimeError):
    pass


def load():
    """Load libfusgpu.so (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FusGpuError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C fenicsx-fus-gpu_amd/csrc`). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = {"fus_stiffness_plan_bytes": _i64, "fus_plan_bytes": _i64, "fus_comm_stream": _vp,
                      "fus_mass_gather_plan_bytes": _i64, "fus_mass_gather_static_bytes": _i64,
                      "fus_halo_ipc_blob_bytes": _i64, "fus_comm_last_error": C.c_char_p}.get(name, _int)
    lib.fus_error_string.argtypes = [_int]
    lib.fus_error_string.restype = C.c_char_p
    lib.fus_source_hash.argtypes = []
    lib.fus_source_hash.restype = C.c_char_p
    if lib.fus_abi_version() != ABI_VERSION:
        raise FusGpuError(f"{LIB_PATH}: ABI version {lib.fus_abi_version()}, this package needs {ABI_VERSION} "
                          "(include/fus_gpu.h FUS_ABI_VERSION): rebuild the library")
    _lib = lib
    return lib


ERR_COMM = -5
ERR_UNSUPPORTED_ENTITY = -3


def tree_source_hash():
    """The hash csrc/Makefile embeds (fus_source_hash()), recomputed from the sources in this tree."""
    import hashlib
    import re

    csrc = os.path.join(_HERE, "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    names = re.search(r"^SRC = (.*)$", mk, re.M).group(1).split() + re.search(r"^HDR = (.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for n in sorted(names):
        with open(os.path.join(csrc, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_from_tree():
    """True if the loaded library was built from exactly the sources in this tree."""
    return load().fus_source_hash().decode() == tree_source_hash()


def check(rc: int, what: str = "", comm=None):
    if rc != 0:
        msg = load().fus_error_string(rc).decode()
        if rc == ERR_COMM:
            detail = load().fus_comm_last_error(comm)
            msg += ": " + (detail.decode() if detail else "?")
        elif comm is not None and rc == -1:  # a misuse the communicator explains (fork / join stream contract)
            detail = load().fus_comm_last_error(comm)
            if detail:
                msg += ": " + detail.decode()
        raise FusGpuError(f"{what or 'libfusgpu call'} failed: {msg} (code {rc})")


def torch_dtype(float_type):
    dt = np.dtype(float_type)
    if dt == np.float64:
        return torch.float64
    if dt == np.float32:
        return torch.float32
    raise TypeError(f"float_type must be np.float32 or np.float64, got {float_type!r}")


def suffix(dtype: torch.dtype) -> str:
    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    raise TypeError(f"unsupported dtype {dtype}")


def require_device_tensor(t, dtype, name: str):
    """Type checks mirroring what numba would reject at dispatch time."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a device array (torch.Tensor on the GPU), got {type(t).__name__}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise FusGpuError(f"{name}: tensor is on {t.device}; the operators only run on the GPU (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"{name}: array must be C-contiguous")
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The caller's current HIP stream (torch's), as the C ABI's ``void* stream``.  The raw getters cost
    ~0.3 us; ``torch.cuda.current_stream()`` builds a Stream object and re-validates the device on every
    call (~3 us of a ~10 us launch from Python: tools/prof_host.py)."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def device_info(device: int = 0):
    lib = load()
    name = C.create_string_buffer(256)
    cu, hbm, lds = _int(0), _i64(0), _int(0)
    check(lib.fus_device_info(device, name, C.byref(cu), C.byref(hbm), C.byref(lds)), "fus_device_info")
    return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": hbm.value, "lds_bytes_per_cu": lds.value}


def set_tuning(key: int, value: int):
    check(load().fus_set_tuning(key, value), "fus_set_tuning")


def get_tuning(key: int) -> int:
    return load().fus_get_tuning(key)
