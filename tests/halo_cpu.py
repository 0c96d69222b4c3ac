"""Test-only stand-ins injected into the product's scatterer / HaloApply so the
N > 1 host logic (partition, halo plan, exchange ordering) runs under gloo on
CPU tensors.  Backed by the ORACLE; never imported by the product."""

import numpy as np
import torch

from oracle import oracle_np


class OracleHaloKernels:
    def __init__(self, dtype=torch.float64):
        self.dtype = dtype

    def index_tensor(self, idx_np):
        return torch.from_numpy(np.ascontiguousarray(idx_np, dtype=np.int64))

    def buffer(self, n):
        return torch.zeros(int(n), dtype=self.dtype)

    def pack_fwd(self, in_, out, index):
        oracle_np.pack(in_.numpy(), out.numpy(), index.numpy())

    def unpack_fwd(self, in_, out, index, N):
        oracle_np.unpack_fwd(in_.numpy(), out.numpy()[N:], index.numpy())

    def pack_rev(self, in_, out, index, N):
        oracle_np.pack(in_.numpy()[N:], out.numpy(), index.numpy())

    def unpack_rev(self, in_, out, index):
        oracle_np.unpack_rev(in_.numpy(), out.numpy(), index.numpy())


def global_cell_constants(mesh, dtype=np.float64):
    """Rank-independent per-cell constants (function of the global cell index)."""
    ijk = mesh._cell_ijk + np.array([mesh.cell_range[a][0] for a in range(3)])[None, :]
    return (1.0 + 0.1 * ((ijk[:, 0] * 7 + ijk[:, 1] * 3 + ijk[:, 2]) % 5)).astype(dtype)
