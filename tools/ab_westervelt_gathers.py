#!/usr/bin/env python3
"""Fused Westervelt step (config 5's shape: P = 6, 36^3 bowl-warped cells): the single-gather form (the vector pass writes w = u_n + kappa v_n, the
cell pass is a plain stiffness apply on w: available when c4 / c3 is uniform) against the two-gather form (cell pass gathers u_n and v_n), with the
general G array and with G formed in the kernel -- alternating rounds in one process, ms per step."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=36)
    ap.add_argument("--degree", type=int, default=6)  # (config 3's shape: --degree 4 --cells 54)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    torch.cuda.set_device(0)
    boxmesh, ls, nls = (fusgpu_loader.submodule(m) for m in ("boxmesh", "linear_solver", "nonlinear_solver"))
    L, P = 0.12, a.degree

    def bowl(xg):
        out = xg.copy()
        yy, zz = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
        out[:, 0] = xg[:, 0] + 0.15 * (L / a.cells) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
        return out

    mesh = boxmesh.BoxMesh(P, a.cells, length=L, warp=bowl)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    variants = {}
    for geo in (False, True):
        for two in (False, True):
            s = nls.WesterveltSpectral3D(mesh, np.float64, speed_of_sound=1500.0, source_frequency=0.5e6, fused=True, in_kernel_geometry=geo,
                                         uniform_ratio=False if two else True, keep_G=False)
            s.init()
            s.rk4(0.0, tf, dt, max_steps=3)
            variants[f"{'in-kernel geometry' if geo else 'general G'}, {'two gathers' if two else 'single gather (w = u + kappa v)'}"] = s
    res = {k: [] for k in variants}
    t = 3 * dt
    for _ in range(a.rounds):
        for k, s in variants.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.rk4(t, tf, dt, max_steps=a.steps)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / a.steps * 1e3)
        t += a.steps * dt
    for k, v in res.items():
        print(f"P={P} {a.cells}^3 cells  {k:58s} {np.median(v):.4f} ms/step   rounds {' '.join(f'{x:.4f}' for x in v)}", flush=True)


if __name__ == "__main__":
    main()
