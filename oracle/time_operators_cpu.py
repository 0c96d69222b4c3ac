#!/usr/bin/env python3
"""BASELINE config 1 (the reference's own CPU-runnable case): numba-cpu/time_operators.py's
protocol -- warm-up, 10 timed reps of the cell mass, stiffness and boundary-facet mass applies,
mean +/- std, b zeroed outside the timed region (:181-187, 227-233, 254-260) -- run with the
ORACLE's C restatement of the reference's numba-cpu operators (numba is not installed anywhere in
this pipeline).  Part of the oracle (test / baseline infrastructure, not product code).  Single thread, as the reference (njit without parallel=True).

    python oracle/time_operators_cpu.py [--degree 2 --cells 18] [--threads 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--cells", type=int, default=18)
    ap.add_argument("--nreps", type=int, default=10)
    ap.add_argument("--threads", type=int, default=1)
    a = ap.parse_args()
    from conftest import build_problem, pkg
    from oracle import oracle_c

    try:
        oracle_c.build(native=True)
        O = oracle_c.OracleLib(native=True)
    except Exception:
        O = oracle_c.OracleLib()
    P, n = a.degree, a.degree + 1
    pb = build_problem(P, a.cells, random_constants=False)
    mesh = pb["mesh"]
    print(f"Number of degrees-of-freedom: {mesh.ndofs_global}")
    gll, pre = pkg("gll"), pkg("precompute")
    bd = mesh.boundary_facets()
    fdm = mesh.facet_dofmap(bd)
    dF = np.zeros((bd.shape[0], n * n))
    pre.compute_boundary_facets_scaled_jacobian_determinant(
        dF, (mesh.x_dofs, mesh.x_g), bd, pre.tabulate_facet_gradients(pb["pts"]), gll.tensor_weights_2d(pb["wts"]))
    fc = np.ones(bd.shape[0])
    b = np.zeros(mesh.ndofs)
    u1 = np.ones(mesh.ndofs)

    def timeit(name, fn):
        fn()
        ts = []
        for _ in range(a.nreps):
            b[:] = 0.0
            t0 = time.perf_counter_ns()
            fn()
            ts.append((time.perf_counter_ns() - t0) * 1e-9)
        ts = np.array(ts)
        print(f"Elapsed time ({name}): {ts.mean():.6e} +/- {ts.std():.2e} s   ({mesh.ndofs / ts.mean() / 1e6:.1f} MDOF/s)")

    timeit("mass operator", lambda: O.mass_apply(u1, pb["cc"], b, pb["detJ"], mesh.dofmap))
    timeit("stiffness operator", lambda: O.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], b, pb["G"], mesh.dofmap, threads=a.threads))
    timeit("boundary facet operator", lambda: O.mass_apply(u1, fc, b, dF, fdm))


if __name__ == "__main__":
    main()
