"""
``numba.cuda``-shaped device shim over torch (ROCm) tensors, so drivers written
against the reference's cuda flavour (cuda/demo_linear_box.py:41-51,364-385,
536,574-575) keep their shape:

    cuda.select_device(rank); d = cuda.to_device(a); e = cuda.device_array(n, dtype)
    cuda.synchronize(); a = d.copy_to_host()
"""

from __future__ import annotations

import numpy as np
import torch


class DeviceArray(torch.Tensor):
    """torch.Tensor with numba's ``copy_to_host`` / ``size`` conveniences."""

    def copy_to_host(self, ary=None):
        host = self.detach().cpu().numpy()
        if ary is not None:
            ary[...] = host
            return ary
        return host


def _wrap(t: torch.Tensor) -> DeviceArray:
    return t.as_subclass(DeviceArray)


def is_available() -> bool:
    return torch.cuda.is_available()


def detect():
    if not torch.cuda.is_available():
        print("No GPU visible")
        return False
    for i in range(torch.cuda.device_count()):
        p = torch.cuda.get_device_properties(i)
        print(f"id {i}  {p.name}  {p.total_memory / 2**30:.0f} GiB  {p.multi_processor_count} CUs")
    return True


def select_device(index: int):
    torch.cuda.set_device(int(index) % max(torch.cuda.device_count(), 1))


def get_current_device():
    return torch.device("cuda", torch.cuda.current_device())


def to_device(ary) -> DeviceArray:
    """Host array (numpy) -> device array (contiguous copy)."""
    if isinstance(ary, torch.Tensor):
        return _wrap(ary.contiguous().to(get_current_device()))
    a = np.ascontiguousarray(ary)
    return _wrap(torch.from_numpy(a).to(get_current_device()))


def device_array(shape, dtype=np.float64) -> DeviceArray:
    from ._lib import torch_dtype

    dt = np.dtype(dtype)
    tdt = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}.get(dt)
    if tdt is None:
        tdt = torch_dtype(dt)
    return _wrap(torch.empty(shape, dtype=tdt, device=get_current_device()))


def copy_to_host(d_ary) -> np.ndarray:
    return d_ary.detach().cpu().numpy()


def synchronize():
    torch.cuda.synchronize()
