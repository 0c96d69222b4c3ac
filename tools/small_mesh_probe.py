#!/usr/bin/env python3
"""Small meshes (BASELINE config 2: P = 4, 25^3 cells = 1 563 workgroups on 1 024 resident slots): time of one stiffness
apply against the number of workgroups, back to back and isolated, HBM-cold (working set cycled through buffers larger
than the 256 MB Infinity Cache) and cache-warm (same arrays every launch).  Prints one line per mesh size."""
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

    sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,20,22,25,27,29,32,40").split(",")]
    reps = 200
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ops = fusgpu_loader.submodule("operators")
    P = 4
    for N in sizes:
        pb = build_problem(P, N, perturb=0.16)
        mesh = pb["mesh"]
        x, cc, G = (torch.from_numpy(pb[k]).to(dev) for k in ("x", "cc", "G"))
        dm = torch.from_numpy(mesh.dofmap).to(dev)
        y = torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
        op = ops.stiffness_operator(P, pb["D"].flatten(), np.float64)
        op.prepare(dm)
        # cold: enough copies of the big per-cell array that a launch never finds its G in the Infinity Cache
        ncopy = max(2, int(600e6 // (G.numel() * 8)) + 1)
        Gs = [G.clone() for _ in range(ncopy)]

        def timed(fn, n=reps):
            fn(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3

        warm = timed(lambda i: op(x, cc, y, G, dm))
        cold = timed(lambda i: op(x, cc, y, Gs[i % ncopy], dm))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for a, b in ev:
            torch.cuda.synchronize()
            a.record()
            op(x, cc, y, G, dm)
            b.record()
        torch.cuda.synchronize()
        iso = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
        nwg = (mesh.ncells + 9) // 10
        byts = mesh.ncells * 8044
        print(f"N={N:3d} cells {mesh.ncells:7d} workgroups {nwg:6d} = {nwg / 1024:5.2f} x 1024 slots | back-to-back warm {warm:7.1f} us "
              f"({byts / warm / 1e6:5.2f} TB/s) cold {cold:7.1f} us ({byts / cold / 1e6:5.2f} TB/s) | isolated launch {iso:7.1f} us | "
              f"us per 1024-workgroup round (warm) {warm / (nwg / 1024):6.1f}", flush=True)
        del Gs


if __name__ == "__main__":
    main()
