#!/usr/bin/env python3
"""Interleaved A/B timing of stiffness-kernel tunings in ONE process (guide rule 24):
N variants x M rounds on the BASELINE config-3 workload; prints median / min per variant.

    python tools/ab_stiffness.py [--cells 54] [--degree 4] [--rounds 7] [--reps 10] v:r [v:r ...]
where each ``v:r`` is (kernel):(xcd remap).  Kernels: ``0``/``1`` = plan-free column kernel with
~256 / ~128-thread workgroups; ``p`` = planned kernel on a run-length coded plan, ``r`` = planned
kernel on a raw plan; ``100 + k`` = build k of the planned kernel (FUS_TUNE_PLAN_VARIANT, see
csrc/fus_gpu.hip: 0 default, 1 LDS-aliased, 2/3 occupancy hints, 30 fp32 5-waves build; k >= 4
are the experimental builds -- SoA G, ablations, volatile LDS reads, G streaming, 128-thread
batches, persistent kernel -- and need ``make -C fenicsx-fus-gpu_amd/csrc EXPERIMENTS=1``).
Results of the studies run with this tool: profiles/r01*_ab_*.log, r01d_ablation_and_experiments.log."""
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
    ap.add_argument("configs", nargs="*", default=["p:1", "p:0", "0:1", "1:0"])
    a = ap.parse_args()
    import torch

    import bench
    import fusgpu_loader
    from conftest import build_problem

    lib = fusgpu_loader.submodule("_lib")
    ops = fusgpu_loader.submodule("operators")
    dt = np.float64 if a.dtype == "f64" else np.float32
    pb = build_problem(a.degree, a.cells, dtype=dt, perturb=0.16)
    mesh = pb["mesh"]
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(pb["x"]).to(dev)
    cc = torch.from_numpy(pb["cc"]).to(dev)
    G = torch.from_numpy(pb["G"]).to(dev)
    dm = torch.from_numpy(mesh.dofmap).to(dev)
    y = torch.zeros(mesh.ndofs, dtype=x.dtype, device=dev)
    op = ops.stiffness_operator(a.degree, pb["D"].flatten(), dt)
    # "p" = planned (run-length plan), "r" = planned with a raw (uncompressed) plan
    cfgs = [tuple({"p": -1, "r": -2}.get(v, None) if v in ("p", "r") else int(v) for v in c.split(":")) for c in a.configs]
    G_soa = G.permute(0, 2, 1).contiguous().reshape(G.shape)  # experiment 104: [cell][6][n^3] bytes in a [cell][n^3][6]-shaped tensor
    dm_raw = dm.clone()  # a second dofmap array => its own cached plan, built with runs disabled
    lib.set_tuning(lib.TUNE_PLAN_RUNS, 0)
    ops.use_plan(True)
    op(x, cc, y, G, dm_raw)
    torch.cuda.synchronize()
    lib.set_tuning(lib.TUNE_PLAN_RUNS, 2)  # "p" = run-length plan
    op(x, cc, y, G, dm)
    torch.cuda.synchronize()
    lib.set_tuning(lib.TUNE_PLAN_RUNS, 1)
    dm_128 = dm.clone()  # a third dofmap array: plan cut for 128-thread workgroups (builds 110 / 111)
    lib.set_tuning(lib.TUNE_PLAN_THREADS, 128)
    lib.set_tuning(lib.TUNE_PLAN_VARIANT, 10)
    op(x, cc, y, G, dm_128)
    torch.cuda.synchronize()
    lib.set_tuning(lib.TUNE_PLAN_THREADS, 256)
    lib.set_tuning(lib.TUNE_PLAN_VARIANT, -1)
    times = {c: [] for c in cfgs}
    for rnd in range(a.rounds + 1):
        for c in cfgs:
            ops.use_plan(c[0] < 0 or c[0] >= 100)
            if c[0] >= 100:  # 100 + k = planned kernel build k
                lib.set_tuning(lib.TUNE_PLAN_VARIANT, c[0] - 100)
            elif c[0] >= 0:
                lib.set_tuning(lib.TUNE_STIFFNESS_VARIANT, c[0])
            else:
                lib.set_tuning(lib.TUNE_PLAN_VARIANT, 0)
            lib.set_tuning(lib.TUNE_XCD_REMAP, c[1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d_ = dm_raw if (c[0] == -2 or c[0] >= 120) else (dm_128 if c[0] in (110, 111) else dm)  # persistent builds (120+) read raw plans
            G_ = G_soa if c[0] == 104 else G
            op(x, cc, y, G_, d_)
            e0.record()
            for _ in range(a.reps):
                op(x, cc, y, G_, d_)
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[c].append(e0.elapsed_time(e1) / a.reps)
    bpc = bench.stiffness_bytes_per_cell(a.degree, np.dtype(dt).itemsize)
    for c in cfgs:
        t = np.array(times[c])
        gbs = mesh.ncells * bpc / (np.median(t) * 1e-3) / 1e9
        print(f"variant {c[0]} remap {c[1]}: median {np.median(t):.4f} ms  min {t.min():.4f} ms  "
              f"{gbs:.0f} GB/s  ({100 * gbs / 8000:.1f}% of 8 TB/s)  {mesh.ndofs / np.median(t) / 1e6:.2f} GDOF/s")


if __name__ == "__main__":
    main()
