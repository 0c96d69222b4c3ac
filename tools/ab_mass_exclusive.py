#!/usr/bin/env python3
"""Cell mass apply with and without exclusive-dof marks in the batch plan (csrc/plan.hpp, fus_plan_mark_exclusive), ~10 M
dofs per degree, K back-to-back launches between one HIP-event pair, alternating rounds.  Prints ms, TB/s of the
algorithmic bytes and the fraction of the plan's (batch, dof) sums that are finished without an atomic."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degrees", default="4,2,3,5,6")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    ops, boxmesh, gll, pre = (fusgpu_loader.submodule(m) for m in ("operators", "boxmesh", "gll", "precompute"))
    lib = fusgpu_loader.submodule("_lib").load()
    dev = torch.device("cuda", 0)
    for P in (int(v) for v in a.degrees.split(",")):
        n = P + 1
        N = max(2, round((10.2e6) ** (1 / 3) / P))
        mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
        pts, wts, _ = gll.tabulate_1d(P)
        detJ = torch.empty((mesh.ncells, n**3), dtype=torch.float64, device=dev)
        pre.compute_scaled_jacobian_determinant_device(detJ, (torch.from_numpy(mesh.x_dofs).to(dev), torch.from_numpy(mesh.x_g).to(dev)), mesh.ncells,
                                                       torch.from_numpy(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))).to(dev),
                                                       torch.from_numpy(gll.tensor_weights_3d(wts)).to(dev))
        x = torch.randn(mesh.ndofs, dtype=torch.float64, device=dev)
        y = torch.zeros_like(x)
        cc = torch.randn(mesh.ncells, dtype=torch.float64, device=dev)
        dm = torch.from_numpy(mesh.dofmap).to(dev)
        opsd = {"atomics only": ops.mass_operator(n**3, np.float64, atomic=True), "exclusive marks": ops.mass_operator(n**3, np.float64, exclusive=True, atomic=True)}
        ref = None
        res = {k: [] for k in opsd}
        for k, op in opsd.items():
            y.zero_()
            op(x, cc, y, detJ, dm)
            if ref is None:
                ref = y.clone()
            else:
                err = float((y - ref).norm() / ref.norm())
                assert err < 1e-13, err
        for _ in range(a.rounds):
            for k, op in opsd.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                op(x, cc, y, detJ, dm)
                e0.record()
                for _ in range(a.reps):
                    op(x, cc, y, detJ, dm)
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / a.reps)
        ws, epb = ops._PLANS.get(dm, exclusive_ndofs=mesh.ndofs)
        nbatch = (mesh.ncells + epb - 1) // epb
        nbytes = int(lib.fus_plan_bytes(n**3, epb, mesh.ncells))
        words = (epb * n**3 + 31) // 32
        ex = ws[nbytes - ((nbatch * words * 4 + 255) // 256 * 256):][: nbatch * words * 4].view(torch.int32).cpu().numpy().view(np.uint32)
        marked = int(np.unpackbits(ex.view(np.uint8)).sum())
        nu = int((ws[256:256 + 4 * nbatch].view(torch.int32).cpu().numpy() & 0xFFFF).sum())
        bpc = n**3 * 8 + 4 * n**3 + 3 * 8 * P**3 + 8
        line = f"P={P} {mesh.ncells} cells {mesh.ndofs} dofs, {epb} cells/batch, {nu / nbatch:.0f} distinct dofs/batch, {100.0 * marked / nu:.1f} % exclusive:"
        for k, v in res.items():
            ms = float(np.median(v))
            line += f"  {k} {ms:.4f} ms = {mesh.ncells * bpc / ms / 1e9:.2f} TB/s ({100 * mesh.ncells * bpc / ms / 1e9 / 8:.1f} %)"
        print(line, flush=True)
        del detJ, x, y, cc, dm


if __name__ == "__main__":
    main()
