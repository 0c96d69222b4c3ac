#!/usr/bin/env python3
"""A/B: the cell mass apply through the batch plan (LDS pre-reduction + float atomics, mass_plan_kernel) against the
atomic-free transposed-dofmap kernel (csrc/mass_gather.hpp), interleaved in one process; checks both against each other,
the gather kernel for run-to-run bitwise reproducibility, and prints the plan-build time.

    python tools/ab_mass_gather.py [--degree 4] [--cells 54] [--dtype f64] [--rounds 7] [--reps 20] [--order lex|random]
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
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--cells", type=int, default=54)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--order", default="lex", choices=["lex", "random"])
    ap.add_argument("--variants", default="0", help="builds of the gather kernel to compare (FUS_TUNE_MASS_VARIANT: rows per thread 1, 2, 4; 0 = chosen by size and type)")
    a = ap.parse_args()
    import torch

    import fusgpu_loader
    from conftest import build_problem

    lib = fusgpu_loader.submodule("_lib")
    ops = fusgpu_loader.submodule("operators")
    L = lib.load()
    dt = np.float64 if a.dtype == "f64" else np.float32
    T = np.dtype(dt).itemsize
    pb = build_problem(a.degree, a.cells, dtype=dt, perturb=0.16)
    mesh = pb["mesh"]
    dev = torch.device("cuda", 0)
    perm = np.arange(mesh.ncells) if a.order == "lex" else np.random.default_rng(7).permutation(mesh.ncells)
    x = torch.from_numpy(pb["x"]).to(dev)
    cc = torch.from_numpy(pb["cc"][perm]).to(dev)
    detJ = torch.from_numpy(pb["detJ"][perm]).to(dev)
    dm = torch.from_numpy(mesh.dofmap[perm]).to(dev)
    nent, N = dm.shape
    nd = mesh.ndofs
    P = a.degree
    alg = nent * (N * T + 4 * N + 3 * T * P**3 + T)

    t0 = time.perf_counter()
    nbytes = L.fus_mass_gather_plan_bytes(N, nent, nd)
    assert nbytes > 0, nbytes
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    lib.check(L.fus_mass_gather_plan_build(dm.data_ptr(), N, nent, nd, ws.data_ptr(), int(nbytes), lib.stream_ptr()), "gather build")
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    import ctypes as C

    info = (C.c_int64 * 4)()
    lib.check(L.fus_mass_gather_plan_info(ws.data_ptr(), info), "info")
    print(f"P={P} cells={a.cells}^3 {a.dtype} order={a.order} dofs={nd}: gather plan {nbytes / 1e6:.1f} MB, rows {info[0]}, dense {info[1]}, "
          f"max entries per dof {info[2]}, build {t_build * 1e3:.1f} ms (first call: includes the sort's code load)")
    fn_g = getattr(L, f"fus_mass_apply_gather_{a.dtype}")
    mop = ops.mass_operator(N, dt)

    def plan(y):  # the float-atomic batch-plan kernel
        mop.atomic(x, cc, y, detJ, dm)

    def gather(y):
        lib.check(fn_g(x.data_ptr(), cc.data_ptr(), y.data_ptr(), detJ.data_ptr(), ws.data_ptr(), N, nent, lib.stream_ptr()), "gather")

    # the static-detJ form (round 5): detJ in row order next to the transposed dofmap, streamed instead of gathered
    sbytes = L.fus_mass_gather_static_bytes(N, nent, T)
    sws, fn_s = None, getattr(L, f"fus_mass_apply_gather_static_{a.dtype}")
    if sbytes > 0:
        sws = torch.empty(int(sbytes), dtype=torch.uint8, device=dev)
        rc = getattr(L, f"fus_mass_gather_static_build_{a.dtype}")(ws.data_ptr(), detJ.data_ptr(), sws.data_ptr(), int(sbytes), lib.stream_ptr())
        if rc != 0:
            print(f"static companion declined (rc {rc})")
            sws = None

    def static(y):
        lib.check(fn_s(x.data_ptr(), cc.data_ptr(), y.data_ptr(), ws.data_ptr(), sws.data_ptr(), N, nent, lib.stream_ptr()), "gather static")

    ya, yb, yc = (torch.zeros(nd, dtype=x.dtype, device=dev) for _ in range(3))
    if sws is not None:
        static(yc)
        gather(yb)
        torch.cuda.synchronize()
        print(f"static-detJ gather vs gather: bitwise equal: {bool(torch.equal(yb, yc))}")
        yb.zero_(), yc.zero_()
    plan(ya)
    gather(yb)
    gather(yc)
    torch.cuda.synchronize()
    rel = float(((ya - yb).norm() / ya.norm()).item())
    print(f"gather vs plan: rel l2 {rel:.2e}; gather run-to-run bitwise equal: {bool(torch.equal(yb, yc))}")
    # against the serial loop order on the host (float64 only: the sums visit the entries in the same order)
    if a.cells <= 20:
        ref = np.zeros(nd, dtype=dt)
        dmh, dJ, cch, xh = mesh.dofmap[perm], pb["detJ"][perm], pb["cc"][perm], pb["x"]
        np.add.at(ref, dmh.reshape(-1), (xh[dmh] * dJ * cch[:, None]).reshape(-1))
        print(f"gather vs numpy add.at: max abs diff {np.abs(yb.cpu().numpy() - ref).max():.3e} (ref max {np.abs(ref).max():.3e})")

    def timeit(fn, y):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn(y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    variants = [int(v) for v in a.variants.split(",")]
    res = {"plan": []}
    res.update({f"gather:{v}": [] for v in variants})
    if sws is not None:
        res.update({f"static:{v}": [] for v in variants})
    for fn in (plan, gather):
        timeit(fn, ya)
    for _ in range(a.rounds):
        res["plan"].append(timeit(plan, ya))
        for v in variants:
            lib.set_tuning(lib.TUNE_MASS_VARIANT, v)
            res[f"gather:{v}"].append(timeit(gather, yb))
            if sws is not None:
                res[f"static:{v}"].append(timeit(static, yc))
        lib.set_tuning(lib.TUNE_MASS_VARIANT, 0)
    for k, v in res.items():
        med = float(np.median(v))
        print(f"{k:9s}: median {med:.4f} ms  min {min(v):.4f} ms  {alg / med / 1e6:7.0f} GB/s algorithmic ({100 * alg / med / 1e6 / 8000:.1f} % of 8 TB/s)  "
              f"{nd / med / 1e6:.2f} GDOF/s")


if __name__ == "__main__":
    main()
