#!/usr/bin/env python3
"""Sweep the stiffness (and mass) apply over degrees / dtypes at ~10 M dofs and print
time, DOF/s and fraction of the 8 TB/s HBM roofline (algorithmic bytes)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def mass_bytes_per_cell(P, T):
    n = P + 1
    return n**3 * T + 4 * n**3 + 3 * T * P**3 + T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dofs", type=float, default=10e6)
    ap.add_argument("--degrees", default="2,3,4,5,6,8")
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--affine", action="store_true")
    a = ap.parse_args()
    import torch

    import bench
    import fusgpu_loader
    from conftest import build_problem

    ops = fusgpu_loader.submodule("operators")
    dev = torch.device("cuda", 0)

    def timeit(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    for P in [int(v) for v in a.degrees.split(",")]:
        N = max(2, int(round((a.dofs ** (1 / 3) - 1) / P)))
        for dname in a.dtypes.split(","):
            dt = np.float64 if dname == "f64" else np.float32
            T = np.dtype(dt).itemsize
            pb = build_problem(P, N, dtype=dt, perturb=0.0 if a.affine else 0.16)
            mesh = pb["mesh"]
            x, cc = torch.from_numpy(pb["x"]).to(dev), torch.from_numpy(pb["cc"]).to(dev)
            G, detJ = torch.from_numpy(pb["G"]).to(dev), torch.from_numpy(pb["detJ"]).to(dev)
            dm = torch.from_numpy(mesh.dofmap).to(dev)
            y = torch.zeros(mesh.ndofs, dtype=x.dtype, device=dev)
            op = ops.stiffness_operator(P, pb["D"].flatten(), dt)
            mop = ops.mass_operator((P + 1) ** 3, dt)
            res = []
            for plan in (True, False):
                ops.use_plan(plan)
                t = timeit(lambda: op(x, cc, y, G, dm))
                gbs = mesh.ncells * bench.stiffness_bytes_per_cell(P, T) / (t * 1e-3) / 1e9
                res.append(f"K[{'plan' if plan else 'atomic'}] {t:.4f} ms {mesh.ndofs / t / 1e6:6.2f} GDOF/s {100 * gbs / 8000:5.1f}%")
            ops.use_plan(True)
            if a.affine:
                gll = fusgpu_loader.submodule("gll")
                opa = ops.stiffness_operator(P, pb["D"].flatten(), dt, affine_weights=gll.tensor_weights_3d(pb["wts"]))
                lib = fusgpu_loader.submodule("_lib")
                for av in (-1, 40, 42, -1, 40, 42):  # builds of the affine kernel, interleaved twice
                    lib.set_tuning(lib.TUNE_PLAN_VARIANT, av)
                    t = timeit(lambda: opa(x, cc, y, G, dm))
                    res.append(f"K[affine build {av}] {t:.4f} ms {mesh.ndofs / t / 1e6:6.2f} GDOF/s")
                lib.set_tuning(lib.TUNE_PLAN_VARIANT, -1)
            gll = fusgpu_loader.submodule("gll")
            pts1, wts1 = gll.gll_points_weights(P)
            opg = ops.stiffness_operator(P, pb["D"].flatten(), dt, geometry=(mesh.x_dofs, mesh.x_g, pts1, wts1))
            for _ in range(100):  # the in-kernel-geometry kernel needs the steady state (profiles/r05d_geom_variance_probe.log)
                opg(x, cc, y, None, dm)
            t = timeit(lambda: opg(x, cc, y, None, dm))
            gbs = mesh.ncells * bench.geom_bytes_per_cell(P, T) / (t * 1e-3) / 1e9
            res.append(f"K[in-kernel geometry] {t:.4f} ms {mesh.ndofs / t / 1e6:6.2f} GDOF/s {100 * gbs / 8000:5.1f}% of its own contract")
            mops = ops.mass_operator((P + 1) ** 3, dt, static_detJ=True)
            # default (atomic-free from P = 3 up), detJ streamed in row order (opt-in), the float-atomic twin
            for name, fn in (("M", mop), ("M[static detJ]", mops), ("M[atomic]", mop.atomic)):
                t = timeit(lambda: fn(x, cc, y, detJ, dm))
                gbs = mesh.ncells * mass_bytes_per_cell(P, T) / (t * 1e-3) / 1e9
                res.append(f"{name} {t:.4f} ms {mesh.ndofs / t / 1e6:6.2f} GDOF/s {100 * gbs / 8000:5.1f}%")
            print(f"P={P} N={N} {dname} cells={mesh.ncells} dofs={mesh.ndofs}: " + " | ".join(res), flush=True)
            del x, cc, G, detJ, dm, y, pb
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
