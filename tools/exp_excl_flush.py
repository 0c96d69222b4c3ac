#!/usr/bin/env python3
"""EXPERIMENT (not a product path): the general-G stiffness kernel finishing the dofs that only ONE batch touches with a plain load +
store instead of a memory-side float atomic.  Needs a library built from a copy of csrc/ with the experimental flush
(FUS_LIB_PATH=tools/_bin/libfusgpu_excl.so); with the shipped library the marks are built and ignored.  Config 3; prints the steady-state
time, the share of marked dofs, and a checksum of one apply (compare across libraries)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import fusgpu_loader
    from conftest import build_problem

    ops, lib = fusgpu_loader.submodule("operators"), fusgpu_loader.submodule("_lib")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    P, N = 4, 54
    pb = build_problem(P, N, dtype=np.float64, perturb=0.16)
    mesh = pb["mesh"]
    x, cc, G = (torch.from_numpy(pb[k]).to(dev) for k in ("x", "cc", "G"))
    dm = torch.from_numpy(mesh.dofmap).to(dev)
    y = torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
    op = ops.stiffness_operator(P, pb["D"].flatten(), np.float64)
    ws, epb = ops._PLANS.get(dm)
    use = torch.zeros(mesh.ndofs, dtype=torch.int32, device=dev)
    clib = lib.load()
    lib.check(clib.fus_plan_mark_exclusive(ws.data_ptr(), (P + 1) ** 3, epb, mesh.ncells, use.data_ptr(), mesh.ndofs, lib.stream_ptr()), "mark")
    torch.cuda.synchronize()
    print(f"dofs touched by exactly one batch: {int((use == 1).sum())} of {mesh.ndofs} ({100.0 * float((use == 1).sum()) / mesh.ndofs:.1f} %); "
          f"(batch, dof) incidences {int(use.sum())}")
    op(x, cc, y, G, dm)
    torch.cuda.synchronize()
    print(f"one apply on y = 0: sum |y| = {float(y.abs().sum()):.15e}  y[12345] = {float(y[12345]):.15e}")
    for _ in range(200):
        op(x, cc, y, G, dm)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            op(x, cc, y, G, dm)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 200 * 1e3)
    print(f"{os.path.basename(lib.LIB_PATH):28s} {np.median(ts):7.1f} us   rounds {' '.join(f'{t:.1f}' for t in ts)}")


if __name__ == "__main__":
    main()
