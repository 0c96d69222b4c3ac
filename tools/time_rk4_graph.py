#!/usr/bin/env python3
"""RK4 step time of the linear solver with and without hipGraph replay (rk4 vs rk4_graph) over mesh sizes:
where the launches, not the kernels, bound the step (small meshes) the graph removes the host cost."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="2:18,4:12,4:25,4:54", help="P:N,...")
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    boxmesh, ls = fusgpu_loader.submodule("boxmesh"), fusgpu_loader.submodule("linear_solver")
    L = 0.12
    for case in a.cases.split(","):
        P, N = (int(v) for v in case.split(":"))
        mesh = boxmesh.BoxMesh(P, N, length=L)
        h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
        dt, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
        steps = min(a.steps, nstep // 3)
        res = {}
        for name in ("rk4", "rk4_graph"):
            s = ls.LinearSpectral3D(mesh, np.float64)
            s.init()
            fn = getattr(s, name)
            t, _ = fn(0.0, tf, dt, max_steps=steps)  # warm-up (and capture)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t, done = fn(t, tf, dt, max_steps=steps)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / done * 1e3
        print(f"P={P} N={N} dofs={mesh.ndofs}: rk4 {res['rk4']:.4f} ms/step   rk4_graph {res['rk4_graph']:.4f} ms/step   "
              f"({res['rk4'] / res['rk4_graph']:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
