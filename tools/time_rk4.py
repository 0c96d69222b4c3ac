#!/usr/bin/env python3
"""Time one RK4 step of the linear wave solver (BASELINE config 3 shape) on one GPU:
the reference's launch sequence vs the fused stage."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=54)
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--westervelt", action="store_true")
    ap.add_argument("--no-affine", action="store_true", help="general per-quadrature-point G even on the affine box")
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    torch.cuda.set_device(0)
    if os.environ.get("FUS_VECTOR_STREAM") is not None:  # A/B: non-temporal accesses in the vector kernels off (0) / auto (1) / always (2)
        lib_mod = fusgpu_loader.submodule("_lib")
        lib_mod.set_tuning(lib_mod.TUNE_VECTOR_STREAM, int(os.environ["FUS_VECTOR_STREAM"]))
    boxmesh, ls = fusgpu_loader.submodule("boxmesh"), fusgpu_loader.submodule("linear_solver")
    L = 0.12
    mesh = boxmesh.BoxMesh(a.degree, a.cells, length=L)
    h = ls.time_step_parameters(mesh, a.degree, 1500.0, 0.5e6, L)
    dt, tf, nstep = ls.snap_time_step(h, a.degree, 1500.0, 0.5e6, L)
    print(f"P={a.degree} cells={mesh.ncells} dofs={mesh.ndofs} dt={dt:.3e} steps to final time={nstep}")
    nls = fusgpu_loader.submodule("nonlinear_solver")
    for fused in (False, True):
        s = nls.WesterveltSpectral3D(mesh, np.float64, fused=fused) if a.westervelt else ls.LinearSpectral3D(mesh, np.float64, fused=fused, affine=False if a.no_affine else "auto")
        s.init()
        s.rk4(0.0, tf, dt, max_steps=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.rk4(3 * dt, tf, dt, max_steps=a.steps)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / a.steps
        print(f"{'fused' if fused else 'reference sequence'}: {el * 1e3:.3f} ms/step  "
              f"({mesh.ndofs / el / 1e9:.2f} GDOF-steps/s; full run of {nstep} steps = {el * nstep:.1f} s)", flush=True)
        del s
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
