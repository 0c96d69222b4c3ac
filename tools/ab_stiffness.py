#!/usr/bin/env python3
"""Interleaved A/B timing of stiffness-kernel builds in ONE process (guide rule 24):
N variants x M rounds on one workload; prints median / min per variant.

    python tools/ab_stiffness.py [--cells 54] [--degree 4] [--rounds 7] [--reps 10] [--dtype f64] cfg [cfg ...]

``cfg``:  ``plan``      shipped planned kernel, auto build
          ``plan:K``    planned kernel build K (FUS_TUNE_PLAN_VARIANT, csrc/fus_gpu.hip: 0 three cubes + own
                        buffer, 1 LDS-aliased, 2 LDS-aliased + G ring, 30 fp32 5-waves)
          ``raw:K``     the same reading the plan's dof lists instead of its run tables (FUS_TUNE_PLAN_RUNS = 0)
          ``runs:K``    the same reading the run tables whatever the dtype (FUS_TUNE_PLAN_RUNS = 2)
          ``geom``      geometry formed in the kernel from the 8 vertices (no G stream); ``geom:50`` / ``geom:51`` existed in the library of
                        commit 3e1ba31 only (P <= 5 with the flux formed inside the main loop at 4 / 3 waves per SIMD: profiles/r06j_*, r06l_*)
          ``col:V``     plan-free column kernel, workgroup variant V (0: ~256 threads, 1: ~128)
Add ``@x`` to a cfg to run it with the XCD remap on.  Two builds of the library are compared by
alternating runs of this tool with FUS_LIB_PATH=<other libfusgpu.so> on one box.  Results of the studies run with this tool: profiles/r0*_ab_*.log."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=54)
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--isolated", action="store_true", help="one event pair per launch (launches cannot overlap at their tails)")
    ap.add_argument("--order", default="lex", choices=["lex", "random", "morton"], help="cell order of the dofmap")
    ap.add_argument("configs", nargs="*", default=["plan", "plan:0", "plan:1", "plan:2"])
    a = ap.parse_args()
    import torch

    import bench
    import fusgpu_loader
    from conftest import build_problem

    lib = fusgpu_loader.submodule("_lib")
    ops = fusgpu_loader.submodule("operators")
    gll = fusgpu_loader.submodule("gll")
    dt = np.float64 if a.dtype == "f64" else np.float32
    pb = build_problem(a.degree, a.cells, dtype=dt, perturb=0.16)
    mesh = pb["mesh"]
    dev = torch.device("cuda", 0)
    perm = np.arange(mesh.ncells)
    if a.order == "random":
        perm = np.random.default_rng(7).permutation(mesh.ncells)
    x = torch.from_numpy(pb["x"]).to(dev)
    cc = torch.from_numpy(pb["cc"][perm]).to(dev)
    G = torch.from_numpy(pb["G"][perm]).to(dev)
    dm = torch.from_numpy(mesh.dofmap[perm]).to(dev)
    xd = torch.from_numpy(mesh.x_dofs[perm]).to(dev)
    xg = torch.from_numpy(mesh.x_g.astype(dt)).to(dev)
    y = torch.zeros(mesh.ndofs, dtype=x.dtype, device=dev)
    pts, wts, _ = gll.tabulate_1d(a.degree, dt)
    op = ops.stiffness_operator(a.degree, pb["D"].flatten(), dt)
    opg = ops.stiffness_operator(a.degree, pb["D"].flatten(), dt, geometry=(xd, xg, pts, wts))
    op(x, cc, y, G, dm)  # builds the plan (lists + run tables)
    torch.cuda.synchronize()

    def parse(c):
        remap = c.endswith("@x")
        c = c[:-2] if remap else c
        kind, _, k = c.partition(":")
        return kind, int(k) if k else -1, remap

    cfgs = [parse(c) for c in a.configs]
    times = {c: [] for c in a.configs}
    for rnd in range(a.rounds + 1):
        for name, (kind, k, remap) in zip(a.configs, cfgs):
            lib.set_tuning(lib.TUNE_XCD_REMAP, int(remap))
            ops.use_plan(kind != "col")
            fn, d_ = op, dm
            if kind == "col":
                lib.set_tuning(lib.TUNE_STIFFNESS_VARIANT, max(k, 0))
            elif kind == "geom":
                fn = opg
                lib.set_tuning(lib.TUNE_PLAN_VARIANT, k)  # geom:50 / geom:51: the flux formed in the main loop for P <= 5 too (dispatch_geometry.hip)
            else:
                lib.set_tuning(lib.TUNE_PLAN_VARIANT, k)
            lib.set_tuning(lib.TUNE_PLAN_RUNS, {"raw": 0, "runs": 2}.get(kind, 1))
            fn(x, cc, y, G, d_)
            if a.isolated:
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
                for e0, e1 in evs:
                    e0.record()
                    fn(x, cc, y, G, d_)
                    e1.record()
                torch.cuda.synchronize()
                tms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn(x, cc, y, G, d_)
                e1.record()
                torch.cuda.synchronize()
                tms = e0.elapsed_time(e1) / a.reps
            if rnd > 0:
                times[name].append(tms)
    lib.set_tuning(lib.TUNE_PLAN_VARIANT, -1)
    lib.set_tuning(lib.TUNE_PLAN_RUNS, 1)
    bpc = bench.stiffness_bytes_per_cell(a.degree, np.dtype(dt).itemsize)
    print(f"P={a.degree} cells={a.cells}^3 dtype={a.dtype} order={a.order} dofs={mesh.ndofs} lib={lib.LIB_PATH} "
          f"timing={'isolated launches' if a.isolated else 'back-to-back launches'}")
    for name in a.configs:
        t = np.array(times[name])
        gbs = mesh.ncells * bpc / (np.median(t) * 1e-3) / 1e9
        print(f"{name:12s}: median {np.median(t):.4f} ms  min {t.min():.4f} ms  {gbs:.0f} GB/s of general-G bytes "
              f"({100 * gbs / 8000:.1f}% of 8 TB/s)  {mesh.ndofs / np.median(t) / 1e6:.2f} GDOF/s")
        print(f"{'':12s}  rounds: " + " ".join(f"{v:.4f}" for v in t))


if __name__ == "__main__":
    main()
