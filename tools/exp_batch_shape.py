#!/usr/bin/env python3
"""Experiment: what would batches with more dof sharing than 10 cells in a row buy?  The cells of a 50^3
box (P = 4) are physically re-ordered so that 10 consecutive cells form a 1x1x10 row (the lexicographic
order), a 1x2x5 tile or a 2x1x5 tile; locality re-ordering of the plan is off.  Prints distinct dofs per
batch and the time of the planned general and in-kernel-geometry kernels."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import fusgpu_loader  # noqa: E402
from conftest import build_problem  # noqa: E402

ops, gll = fusgpu_loader.submodule("operators"), fusgpu_loader.submodule("gll")
P, N = 4, 50
pb = build_problem(P, N, perturb=0.16)
mesh = pb["mesh"]
dev = torch.device("cuda", 0)
ijk = mesh._cell_ijk  # [ncell, 3] (x, y, z) of each cell in dofmap order
cx, cy, cz = ijk[:, 0], ijk[:, 1], ijk[:, 2]
orders = {
    "1x1x10 rows (lexicographic)": np.lexsort((cz, cy, cx)),
    "1x2x5 tiles": np.lexsort((cz % 5, cy % 2, cz // 5, cy // 2, cx)),
    "2x1x5 tiles": np.lexsort((cz % 5, cx % 2, cz // 5, cy, cx // 2)),
}
pts, wts, _ = gll.tabulate_1d(P)
ops.use_locality_order(False)
x = torch.from_numpy(pb["x"]).to(dev)
y = torch.zeros_like(x)
for name, perm in orders.items():
    dm = torch.from_numpy(np.ascontiguousarray(mesh.dofmap[perm])).to(dev)
    G = torch.from_numpy(np.ascontiguousarray(pb["G"][perm])).to(dev)
    cc = torch.from_numpy(np.ascontiguousarray(pb["cc"][perm])).to(dev)
    xd = torch.from_numpy(np.ascontiguousarray(mesh.x_dofs[perm])).to(dev)
    op = ops.stiffness_operator(P, pb["D"].flatten(), np.float64)
    opg = ops.stiffness_operator(P, pb["D"].flatten(), np.float64, geometry=(xd, mesh.x_g, pts, wts))
    res = []
    for fn in (lambda: op(x, cc, y, G, dm), lambda: opg(x, cc, y, None, dm)):
        ts = []
        for _ in range(5):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        res.append(float(np.median(ts)))
    ws, epb = ops._PLANS.get(dm)
    nb = (mesh.ncells + epb - 1) // epb
    nu = (ws[256:256 + 4 * nb].view(torch.int32) & 0xFFFF).float().mean().item()
    print(f"{name:30s} distinct dofs/batch {nu:7.1f}   general {res[0]:.4f} ms   in-kernel geometry {res[1]:.4f} ms", flush=True)
    ops._PLANS.clear()
