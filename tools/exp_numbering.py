#!/usr/bin/env python3
"""Numbering-sensitivity experiment -- the one kernel experiment the reference ships
(cuda/exp_kernel_speed.py:176-218: stiffness kernel timed with the basix global dof numbering vs a
tensor-product numbering).  Same operator, same mesh (BASELINE config 3 by default), different
orderings of the cells and of the global dofs:

  cells:  lex      consecutive cells adjacent (x slowest) -- what BoxMesh gives
          random   seeded random permutation of the dofmap rows (and of G, constants)
          sorted   the random order put back by sorting cells on their smallest dof
                   (what ``operators.locality_cell_order`` does at set-up)
  dofs:   lex      lexicographic over the global GLL grid
          morton   Z-curve over the global GLL grid
          basix    entity-wise: all vertex dofs, then edge, then face, then cell-interior dofs
                   (the shape of a dolfinx / basix numbering before any bandwidth reordering)
          random   seeded random permutation (worst case)

For each combination: distinct dofs per batch (``nu``: what the planned kernel gathers / flushes per
10-cell workgroup), time and DOF/s of the planned and the plan-free kernel.  Atomic requests per cell
come from a separate rocprofv3 --pmc TCC_EA0_ATOMIC_sum run of this script with --only <combo>."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def morton_key(i, j, k, bits=10):
    key = np.zeros(i.shape, dtype=np.int64)
    for b in range(bits):
        key |= ((i >> b) & 1) << (3 * b + 2)
        key |= ((j >> b) & 1) << (3 * b + 1)
        key |= ((k >> b) & 1) << (3 * b)
    return key


def dof_renumbering(kind, P, dims, seed=5):
    """new_id[old lexicographic id] for the global GLL grid ``dims``."""
    nd = int(np.prod(dims))
    if kind == "lex":
        return np.arange(nd, dtype=np.int64)
    i, j, k = np.unravel_index(np.arange(nd), dims)
    if kind == "random":
        return np.random.default_rng(seed).permutation(nd).astype(np.int64)
    if kind == "morton":
        order = np.argsort(morton_key(i, j, k), kind="stable")
    elif kind == "basix":
        on = (i % P == 0).astype(int) + (j % P == 0).astype(int) + (k % P == 0).astype(int)  # 3 vertex, 2 edge, 1 face, 0 interior
        order = np.argsort(-on, kind="stable")
    else:
        raise ValueError(kind)
    new = np.empty(nd, dtype=np.int64)
    new[order] = np.arange(nd)
    return new


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--cells", type=int, default=54)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default=None, help="cells:dofs, e.g. random:lex")
    a = ap.parse_args()
    import torch

    import bench
    import fusgpu_loader
    from conftest import build_problem

    ops = fusgpu_loader.submodule("operators")
    P = a.degree
    pb = build_problem(P, a.cells, perturb=0.16)
    mesh = pb["mesh"]
    dev = torch.device("cuda", 0)
    dims = mesh.global_dof_dims
    rng = np.random.default_rng(7)
    cell_perm = {"lex": np.arange(mesh.ncells), "random": rng.permutation(mesh.ncells)}
    combos = [("lex", "lex"), ("random", "lex"), ("sorted", "lex"), ("lex", "morton"), ("lex", "basix"), ("random", "basix"),
              ("sorted", "basix"), ("lex", "random"), ("random", "random")]
    if a.only:
        combos = [tuple(a.only.split(":"))]
    op = ops.stiffness_operator(P, pb["D"].flatten(), np.float64)
    bpc = bench.stiffness_bytes_per_cell(P, 8)
    G_h, cc_h = torch.from_numpy(pb["G"]), torch.from_numpy(pb["cc"])
    print(f"P={P} {a.cells}^3 cells {mesh.ndofs} dofs; batch = {256 // (P + 1) ** 2} cells = {256 // (P + 1) ** 2 * (P + 1) ** 3} (cell, dof) entries")
    print(f"{'cells':8s} {'dofs':8s} {'nu/batch':>9s} {'planned ms':>11s} {'GDOF/s':>8s} {'% roof':>7s} {'plan-free ms':>13s} {'GDOF/s':>8s}")
    for ck, dk in combos:
        new = dof_renumbering(dk, P, dims)
        dm_np = new[mesh.dofmap].astype(np.int32)
        x_np = np.empty_like(pb["x"])
        x_np[new] = pb["x"]
        cperm = cell_perm["random"] if ck in ("random", "sorted") else cell_perm["lex"]
        dm_np, G_t, cc_t = dm_np[cperm], G_h[cperm], cc_h[cperm]
        if ck == "sorted":
            order = ops.locality_cell_order(torch.from_numpy(dm_np).to(dev)).cpu().numpy()
            dm_np, G_t, cc_t = dm_np[order], G_t[order], cc_t[order]
        dm = torch.from_numpy(np.ascontiguousarray(dm_np)).to(dev)
        G, cc, x = G_t.contiguous().to(dev), cc_t.contiguous().to(dev), torch.from_numpy(x_np).to(dev)
        y = torch.zeros_like(x)
        res = {}
        for planned in (True, False):
            ops.use_plan(planned)
            ts = []
            for _ in range(a.rounds):
                op(x, cc, y, G, dm)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    op(x, cc, y, G, dm)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.reps)
            res[planned] = float(np.median(ts))
        ops.use_plan(True)
        ws, epb = ops._PLANS.get(dm)
        nbatch = (mesh.ncells + epb - 1) // epb
        nu = ws[256:256 + 4 * nbatch].view(torch.int32).cpu().numpy() & 0xFFFF
        tp, tf = res[True], res[False]
        print(f"{ck:8s} {dk:8s} {nu.mean():9.1f} {tp:11.4f} {mesh.ndofs / tp / 1e6:8.2f} "
              f"{100 * mesh.ncells * bpc / (tp * 1e-3) / 8e12:7.1f} {tf:13.4f} {mesh.ndofs / tf / 1e6:8.2f}", flush=True)
        del G, cc, x, y, dm
        ops._PLANS.clear()


if __name__ == "__main__":
    main()
