#!/usr/bin/env python3
"""
Operator timing harness -- the counterpart of the reference's numba-cpu/time_operators.py and
cuda/time_operators.py (warm-up, then ``nreps`` timed applies of the cell mass, stiffness and
boundary-facet mass operators, mean +/- std; ``b`` zeroed outside the timed region, :181-187,
227-233, 254-260).  The reference hard-codes P = 4, N = 32; BASELINE config 1 asks for P = 2,
~50 k dofs, so both are parameters.

    python fenicsx-fus-gpu_amd/time_operators.py --degree 2 --cells 18            # BASELINE config 1
    python fenicsx-fus-gpu_amd/time_operators.py --degree 4 --cells 32            # the reference's own setting
"""

import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--cells", type=int, default=18)
    ap.add_argument("--nreps", type=int, default=10)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    boxmesh, gll, pre, ops = (fusgpu_loader.submodule(m) for m in ("boxmesh", "gll", "precompute", "operators"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    P, n = a.degree, a.degree + 1
    ft = np.float64 if a.dtype == "f64" else np.float32
    mesh = boxmesh.BoxMesh(P, a.cells, dtype=ft)
    print(f"Number of degrees-of-freedom: {mesh.ndofs_global}")
    pts, wts, D = gll.tabulate_1d(P, ft)
    td = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    gm = (td(mesh.x_dofs), td(mesh.x_g))
    tdt = torch.float64 if ft == np.float64 else torch.float32
    G = torch.empty((mesh.ncells, n**3, 6), dtype=tdt, device=dev)
    detJ = torch.empty((mesh.ncells, n**3), dtype=tdt, device=dev)
    pre.compute_scaled_geometrical_factor_device(
        G, gm, mesh.ncells, td(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts), ft)),
        td(gll.tensor_weights_3d(wts).astype(ft)), detJ=detJ)
    bd = mesh.boundary_facets()
    detJ_f = torch.empty((bd.shape[0], n * n), dtype=tdt, device=dev)
    pre.compute_boundary_facets_scaled_jacobian_determinant_device(
        detJ_f, gm, td(bd.astype(np.int32)), td(pre.tabulate_facet_gradients(pts, ft)), td(gll.tensor_weights_2d(wts).astype(ft)))
    dofmap, fdm = td(mesh.dofmap), td(mesh.facet_dofmap(bd))
    xyz = mesh.dof_coordinates()
    u = td((100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])).astype(ft))
    b = torch.zeros(mesh.ndofs, dtype=tdt, device=dev)
    cc, fc = torch.ones(mesh.ncells, dtype=tdt, device=dev), torch.ones(bd.shape[0], dtype=tdt, device=dev)
    mass_cell, mass_facet = ops.mass_operator(n**3, ft), ops.mass_operator(n * n, ft)
    stiff = ops.stiffness_operator(P, D.flatten(), ft)

    def timeit(name, fn):
        fn()  # warm-up (the reference's JIT call)
        ts = []
        for _ in range(a.nreps):
            ops.fill(0.0, b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        ts = np.array(ts)
        print(f"Elapsed time ({name}): {ts.mean():.6e} +/- {ts.std():.2e} s   ({mesh.ndofs / ts.mean() / 1e9:.2f} GDOF/s)")

    timeit("mass operator", lambda: mass_cell(u, cc, b, detJ, dofmap))
    timeit("stiffness operator", lambda: stiff(u, cc, b, G, dofmap))
    timeit("boundary facet operator", lambda: mass_facet(u, fc, b, detJ_f, fdm))


if __name__ == "__main__":
    main()
