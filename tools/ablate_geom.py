#!/usr/bin/env python3
"""Steady-state time of the in-kernel-geometry stiffness kernel at config 3 with the library named by FUS_LIB_PATH: run once per ABLATION
build (a copy of csrc/ compiled with -DFUS_ABLATE=<bits>: 1 = no global atomics, 2 = no x gather, 4 = no geometry arithmetic) to price the
kernel's parts.  Results are NOT checked (ablated kernels compute garbage)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    import torch

    import fusgpu_loader

    ops, gll, boxmesh, lib_mod = (fusgpu_loader.submodule(m) for m in ("operators", "gll", "boxmesh", "_lib"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    P, N = int(os.environ.get("ABLATE_P", "4")), int(os.environ.get("ABLATE_N", "54"))
    mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
    pts, wts, D = gll.tabulate_1d(P)
    xyz = mesh.dof_coordinates()
    x = torch.from_numpy(100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])).to(dev)
    y = torch.zeros_like(x)
    cc = torch.from_numpy(np.random.default_rng(1234).standard_normal(mesh.ncells)).to(dev)
    dm = torch.from_numpy(mesh.dofmap).to(dev)
    op = ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(mesh.x_dofs, mesh.x_g, pts, wts))
    op(x, cc, y, None, dm)
    torch.cuda.synchronize()
    print(f"   one apply on y = 0: sum |y| = {float(y.abs().sum()):.15e}  y[12345] = {float(y[12345]):.15e}", flush=True)
    short = os.environ.get("ABLATE_SHORT") == "1"  # under rocprofv3 --pmc: a few launches are enough
    for _ in range(10 if short else 300):
        op(x, cc, y, None, dm)
    ts = []
    for _ in range(1 if short else 5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nrep = 20 if short else 200
        for _ in range(nrep):
            op(x, cc, y, None, dm)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / nrep * 1e3)
    print(f"P={P} {N}^3  {os.path.basename(lib_mod.LIB_PATH):28s} {np.median(ts):7.1f} us   rounds {' '.join(f'{t:.1f}' for t in ts)}", flush=True)
    phase_report(lib_mod, lambda: op(x, cc, y, None, dm), np.median(ts))


def phase_report(lib_mod, launch, t_us):
    """Instrumented builds only (-DFUS_ABLATE bit 32): thread 0 of every workgroup stores the shader-clock counter (s_memtime) at 6 points of
    the kernel and the constant 100 MHz counter (s_memrealtime) at its first and last into a per-workgroup log (plain stores); the
    differences are the time a workgroup spends between the points, cycles / real time is the shader clock the kernel actually ran at."""
    import ctypes

    import torch

    lib = ctypes.CDLL(lib_mod.LIB_PATH)
    if not hasattr(lib, "fus_ablate_phase_clk"):
        return
    n = 16384
    buf = (ctypes.c_ulonglong * (n * 8))()
    launch()
    torch.cuda.synchronize()
    lib.fus_ablate_phase_clk(buf, 0)
    v = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    v = v[v[:, 0] != 0]
    nwg = v.shape[0]
    d = np.diff(v[:, :6], axis=1)
    life_cyc = (v[:, 5] - v[:, 0])
    life_us = (v[:, 7] - v[:, 6]) / 100.0
    span_us = (v[:, 7].max() - v[:, 6].min()) / 100.0
    names = ["start -> plan lists, vertices, x gathered (barrier 1)", "column geometry (5 x column_g_at)", "u to LDS, barrier 2, forward + flux, barrier 3",
             "zero sums, barrier, backward + LDS pre-reduction", "flush: one global atomic per distinct dof"]
    print(f"   {nwg} workgroups logged; first start -> last end {span_us:6.1f} us; workgroup lifetime mean {life_cyc.mean():7.0f} (median {np.median(life_cyc):7.0f}) shader cycles ="
          f" {life_us.mean():5.2f} us -> shader clock {life_cyc.sum() / life_us.sum() / 1e3:5.2f} GHz; resident workgroups ~ {life_us.sum() / span_us:5.0f}", flush=True)
    for k, nm in enumerate(names):
        c = d[:, k]
        print(f"      mean {c.mean():7.0f}  median {np.median(c):7.0f}  p10 {np.percentile(c, 10):7.0f}  p90 {np.percentile(c, 90):7.0f} cycles  {100 * c.sum() / life_cyc.sum():5.1f} %   {nm}", flush=True)


if __name__ == "__main__":
    main()
