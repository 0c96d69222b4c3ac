#!/usr/bin/env python3
"""Chained plans for the in-kernel-geometry stiffness kernel (csrc/plan.hpp fus_plan_chain, plan_tiles.chain_order): one workgroup walks
over up to L consecutive batches that are sideways neighbours and keeps the partial sums of their shared face in LDS instead of
flushing it twice with global float atomics.  Config 3 (P = 4, 54^3 perturbed cells) and config 5's shape (P = 6, 36^3): kernel time
for L = 1 (off: the row-ordered plan, one workgroup per batch) and L = 2 ... in alternating rounds under sustained load; every
variant's result against the row-ordered plan's."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--warm", type=int, default=1500)
    ap.add_argument("--chains", default="1,2,3,4,6,8")
    ap.add_argument("--cases", default="4:54,6:36")
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    ops, gll, boxmesh = (fusgpu_loader.submodule(m) for m in ("operators", "gll", "boxmesh"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    Ls = [int(v) for v in a.chains.split(",")]

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3

    for case in a.cases.split(","):
        P, N = (int(v) for v in case.split(":"))
        mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
        pts, wts, D = gll.tabulate_1d(P)
        xyz = mesh.dof_coordinates()
        x = torch.from_numpy(100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])).to(dev)
        y = torch.zeros_like(x)
        cc = torch.from_numpy(np.random.default_rng(1234).standard_normal(mesh.ncells)).to(dev)
        dm = torch.from_numpy(mesh.dofmap).to(dev)
        xd = torch.from_numpy(mesh.x_dofs).to(dev)
        op = ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(xd, mesh.x_g, pts, wts))
        ops.use_strip_order(False)
        info, y_ref = {}, None

        def run(L):
            ops.use_plan_chain(L)
            op(x, cc, y, None, dm)

        for L in Ls:  # build every plan once (each stays cached under its own key), check the result
            ops.use_plan_chain(L)
            ops._PLANS.last_chains = None
            ws, epb = ops._PLANS.get(dm, strips=True)
            nb = (mesh.ncells + epb - 1) // epb
            info[L] = (nb, getattr(ops._PLANS, "last_chains", None))
            y.zero_()
            op(x, cc, y, None, dm)
            torch.cuda.synchronize()
            if y_ref is None:
                y_ref = y.clone()
            else:
                err = float((y - y_ref).norm() / y_ref.norm())
                assert err < 1e-13, (L, err)
        for _ in range(a.warm):
            run(Ls[0])
        torch.cuda.synchronize()
        res = {L: [] for L in Ls}
        for _ in range(a.rounds):
            for L in Ls:
                res[L].append(timed(lambda: run(L)))
        base = float(np.median(res[Ls[0]]))
        for L in Ls:
            t = float(np.median(res[L]))
            nb, nch = info[L]
            print(f"P={P} {N}^3 cells  chain length <= {L}: {nb} batches in {nch if nch else nb} workgroups   {t:7.1f} us   {100 * (t / base - 1):+5.1f} %", flush=True)
        ops._PLANS.clear()
        ops.use_plan_chain(0)
        del op, x, y, cc, dm, xd


if __name__ == "__main__":
    main()
