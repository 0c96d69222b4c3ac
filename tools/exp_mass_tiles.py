#!/usr/bin/env python3
"""Experiment (VERDICT r3 item 4 ii): cell mass apply at config 3 (P = 4, 54^3 perturbed cells) with batches of more sharing than
10 cells in a row -- 2x2xK tiles of up to 32 cells per batch (the plan's 4 096-entry limit), through the C ABI directly
(fus_plan_build_ordered with an explicit cell order and entities_per_batch).  Prints distinct dofs per cell, ms, TB/s."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fusgpu_loader  # noqa: E402

_lib, boxmesh, gll, pre = (fusgpu_loader.submodule(m) for m in ("_lib", "boxmesh", "gll", "precompute"))
lib = _lib.load()
P, N = 4, 54
n = P + 1
dev = torch.device("cuda", 0)
mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
pts, wts, _ = gll.tabulate_1d(P)
detJ = torch.empty((mesh.ncells, n**3), dtype=torch.float64, device=dev)
pre.compute_scaled_jacobian_determinant_device(detJ, (torch.from_numpy(mesh.x_dofs).to(dev), torch.from_numpy(mesh.x_g).to(dev)), mesh.ncells,
                                               torch.from_numpy(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))).to(dev),
                                               torch.from_numpy(gll.tensor_weights_3d(wts)).to(dev))
x = torch.randn(mesh.ndofs, dtype=torch.float64, device=dev)
cc = torch.randn(mesh.ncells, dtype=torch.float64, device=dev)
dm = torch.from_numpy(mesh.dofmap).to(dev)
ijk = mesh._cell_ijk
cx, cy, cz = ijk[:, 0], ijk[:, 1], ijk[:, 2]
cases = [
    ("1x1x10 rows, 10 cells/batch (shipped)", None, 10),
    ("1x1x5 rows, 5 cells/batch", None, 5),
    ("1x1x6 rows, 6 cells/batch", None, 6),
    ("1x1x8 rows, 8 cells/batch", None, 8),
    ("1x1x12 rows, 12 cells/batch", None, 12),
    ("1x1x16 rows, 16 cells/batch", None, 16),
    ("1x1x32 rows, 32 cells/batch", None, 32),
    ("2x2x8 tiles, 32 cells/batch", np.lexsort((cx % 2, cy % 2, cz, cy // 2, cx // 2)), 32),
    ("2x2x5 tiles, 20 cells/batch", np.lexsort((cx % 2, cy % 2, cz, cy // 2, cx // 2)), 20),
    ("1x2x8 tiles, 16 cells/batch", np.lexsort((cy % 2, cz, cy // 2, cx)), 16),
]
bpc = n**3 * 8 + 4 * n**3 + 3 * 8 * P**3 + 8
ref = None
stream = _lib.stream_ptr()
for name, order, epb in cases:
    nbytes = int(lib.fus_plan_bytes(n**3, epb, mesh.ncells))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    od = torch.from_numpy(order.astype(np.int32)).to(dev) if order is not None else None
    _lib.check(lib.fus_plan_build_ordered(dm.data_ptr(), od.data_ptr() if od is not None else None, n**3, epb, mesh.ncells, ws.data_ptr(), nbytes, stream), "plan")
    nb = (mesh.ncells + epb - 1) // epb
    nu = float((ws[256:256 + 4 * nb].view(torch.int32) & 0xFFFF).double().sum().item()) / mesh.ncells

    def run(y):
        _lib.check(lib.fus_mass_apply_planned_f64(x.data_ptr(), cc.data_ptr(), y.data_ptr(), detJ.data_ptr(), ws.data_ptr(), n**3, epb, mesh.ncells, stream), "mass")

    y = torch.zeros_like(x)
    run(y)
    if ref is None:
        ref = y.clone()
    else:
        assert float((y - ref).norm() / ref.norm()) < 1e-13
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(y)
        e0.record()
        for _ in range(50):
            run(y)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50)
    ms = float(np.median(ts))
    print(f"{name:42s} distinct dofs/cell {nu:6.1f}   {ms:.4f} ms = {mesh.ncells * bpc / ms / 1e9:.2f} TB/s ({100 * mesh.ncells * bpc / ms / 1e9 / 8:.1f} %)", flush=True)
    lib.fus_plan_release(ws.data_ptr())
