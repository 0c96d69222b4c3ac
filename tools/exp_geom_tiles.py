#!/usr/bin/env python3
"""Batch shapes for the in-kernel-geometry stiffness kernel, config 3 (P = 4, 54^3 perturbed cells) and
config 5's shape (P = 6, 36^3): distinct dofs per cell and kernel time for

  rows          CPB consecutive cells of a row (the general-G kernels' plan)
  strips        two adjacent rows interleaved (plan_tiles.two_row_strip_order: 2 x 5 pieces at P = 4) -- through the operator
  2x2x5         P = 4 only, 20 cells per batch, 512-thread workgroups, 65 kB of LDS (2 workgroups per CU): needs a library built
                with -DFUS_EXPERIMENT_GEOM_CPB20 (a copy of csrc/: ``make -B libfusgpu.so EXTRA_CXXFLAGS=-DFUS_EXPERIMENT_GEOM_CPB20``,
                then FUS_LIB_PATH=tools/_bin/libfusgpu_cpb20.so); skipped otherwise.  Round 6 took that experimental branch OUT of the shipped
                dispatch (measured +12.7 %, profiles/r05f_*): it lives in the history at d4461de:fenicsx-fus-gpu_amd/csrc/dispatch_geometry.hip

Alternating rounds, medians."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--warm", type=int, default=2000, help="applies before the first timed round (the device's clocks settle under load)")
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    ops, gll, boxmesh, lib_mod = (fusgpu_loader.submodule(m) for m in ("operators", "gll", "boxmesh", "_lib"))
    lib = lib_mod.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)

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

    for P, N in ((4, 54), (6, 36)):
        n = P + 1
        mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
        pts, wts, D = gll.tabulate_1d(P)
        xyz = mesh.dof_coordinates()
        x = torch.from_numpy(100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])).to(dev)
        y = torch.zeros_like(x)
        cc = torch.from_numpy(np.random.default_rng(1234).standard_normal(mesh.ncells)).to(dev)
        dm = torch.from_numpy(mesh.dofmap).to(dev)
        xd = torch.from_numpy(mesh.x_dofs).to(dev)
        op = ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(xd, mesh.x_g, pts, wts))
        variants = {}

        def distinct(ws, epb):
            nb = (mesh.ncells + epb - 1) // epb
            return float((ws[256:256 + 4 * nb].view(torch.int32) & 0xFFFF).sum().item()) / mesh.ncells

        ops.use_strip_order(False)
        ws_rows, epb = ops._PLANS.get(dm, strips=True)
        d_rows = distinct(ws_rows, epb)
        # reference result (rows) for the parity of the other orders
        y.zero_()
        op(x, cc, y, None, dm)
        y_rows = y.clone()
        ops._PLANS.clear()
        ops.use_strip_order(True)
        ws_strips, _ = ops._PLANS.get(dm, strips=True)
        perm = ops._PLANS.last_order  # the strip order, if the plan cache kept it
        d_strips = distinct(ws_strips, epb)
        y.zero_()
        op(x, cc, y, None, dm)
        err = float((y - y_rows).norm() / y_rows.norm())
        assert err < 1e-13, err

        def run_rows():
            ops.use_strip_order(False)
            op(x, cc, y, None, dm)

        def run_strips():
            ops.use_strip_order(True)
            op(x, cc, y, None, dm)

        # both plans stay cached under their own keys: the switch costs a dictionary lookup
        ops.use_strip_order(False)
        ops._PLANS.get(dm, strips=True)
        variants[f"rows ({epb} cells / batch)"] = (run_rows, d_rows)
        variants["two-row strips (order inside the plan)"] = (run_strips, d_strips)
        # the same strips with the per-cell arrays physically permuted (no order indirection in the kernel's prologue)
        if perm is not None:
            pl = perm.long()
            dm_p, xd_p, cc_p = dm[pl].contiguous(), xd[pl].contiguous(), cc[pl].contiguous()
            op_p = ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(xd_p, mesh.x_g, pts, wts))
            ops.use_strip_order(False)
            ws_p, _ = ops._PLANS.get(dm_p, strips=True)
            y.zero_()
            op_p(x, cc_p, y, None, dm_p)
            assert float((y - y_rows).norm() / y_rows.norm()) < 1e-13

            def run_phys():
                ops.use_strip_order(False)
                op_p(x, cc_p, y, None, dm_p)

            variants["two-row strips (arrays permuted)"] = (run_phys, distinct(ws_p, epb))
        if P == 4:
            # 2 x 2 x 5 tiles of 20 cells: the order from the cell lattice, the plan through the generic entry points
            ijk = mesh._cell_ijk
            cx, cy, cz = ijk[:, 0], ijk[:, 1], ijk[:, 2]
            order = np.lexsort((cz % 5, cy % 2, cx % 2, cz // 5, cy // 2, cx // 2)).astype(np.int32)
            od = torch.from_numpy(order).to(dev)
            nbytes = lib.fus_plan_bytes(n**3, 20, mesh.ncells)
            ws20 = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            lib_mod.check(lib.fus_plan_build_ordered(dm.data_ptr(), od.data_ptr(), n**3, 20, mesh.ncells, ws20.data_ptr(), int(nbytes), lib_mod.stream_ptr()),
                          "fus_plan_build_ordered")
            xg = torch.from_numpy(mesh.x_g).to(dev)
            pt, wt = torch.from_numpy(pts).to(dev), torch.from_numpy(wts).to(dev)
            Dd = torch.from_numpy(D.flatten().copy()).to(dev)

            def run_20():
                return lib.fus_stiffness_apply_planned_geom_f64(x.data_ptr(), cc.data_ptr(), y.data_ptr(), xg.data_ptr(), xd.data_ptr(), pt.data_ptr(),
                                                                wt.data_ptr(), ws20.data_ptr(), Dd.data_ptr(), P, mesh.ncells, lib_mod.stream_ptr())

            y.zero_()
            rc = run_20()
            if rc == 0:
                torch.cuda.synchronize()
                err = float((y - y_rows).norm() / y_rows.norm())
                assert err < 1e-13, err
                variants["2x2x5 tiles (20 cells / batch, 512 threads)"] = (run_20, distinct(ws20, 20))
            else:
                print(f"P={P}: 2x2x5 skipped (this library has no 20-cell build: rc {rc})", flush=True)
        res = {k: [] for k in variants}
        for _ in range(a.warm):
            run_rows()
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for k, (fn, _) in variants.items():
                res[k].append(timed(fn))
        base = float(np.median(res[next(iter(res))]))
        for k, (_, d) in variants.items():
            t = float(np.median(res[k]))
            print(f"P={P} {N}^3 cells  {k:46s} distinct dofs / cell {d:6.1f}   {t:7.1f} us   {100 * (t / base - 1):+5.1f} %", flush=True)
        ops._PLANS.clear()
        ops.use_strip_order(False)  # the default
        del op, x, y, cc, dm, xd


if __name__ == "__main__":
    main()
