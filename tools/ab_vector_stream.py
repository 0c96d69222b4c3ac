#!/usr/bin/env python3
"""FUS_TUNE_VECTOR_STREAM policies on one box: 0 cached, 3 non-temporal stores only, 1 non-temporal loads + stores (default) --
the fused RK4 step (general G), the cached-diagonal mass apply (3 vectors of 82 MB: fit the 256 MB Infinity Cache together) and
copy / axpy on 1 GiB operands."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fusgpu_loader  # noqa: E402

lib_mod, ops, boxmesh, ls = (fusgpu_loader.submodule(m) for m in ("_lib", "operators", "boxmesh", "linear_solver"))
torch.cuda.set_device(0)
L = 0.12
mesh = boxmesh.BoxMesh(4, 54, length=L)
h = ls.time_step_parameters(mesh, 4, 1500.0, 0.5e6, L)
dt, tf, _ = ls.snap_time_step(h, 4, 1500.0, 0.5e6, L)
s = ls.LinearSpectral3D(mesh, np.float64, fused=True, affine=False)
s.init()
s.rk4(0.0, tf, dt, max_steps=3)
n = mesh.ndofs
w, x, y = (torch.randn(n, dtype=torch.float64, device="cuda") for _ in range(3))
big, big2 = torch.ones((1 << 30) // 8, dtype=torch.float64, device="cuda"), torch.ones((1 << 30) // 8, dtype=torch.float64, device="cuda")
fn_muladd = lib_mod.load().fus_muladd_f64


def timed(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rnd in range(2):
    for mode, name in ((0, "cached accesses"), (3, "non-temporal stores only"), (1, "non-temporal loads + stores (default)")):
        lib_mod.set_tuning(lib_mod.TUNE_VECTOR_STREAM, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.rk4(3 * dt, tf, dt, max_steps=40)
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 40 * 1e3
        ma = timed(lambda: fn_muladd(w.data_ptr(), x.data_ptr(), y.data_ptr(), n, lib_mod.stream_ptr()), 50)
        cp = timed(lambda: ops.copy(big, big2), 10)
        ax = timed(lambda: ops.axpy(big.numel())(0.5, big, big2), 10)
        print(f"round {rnd}: {name:32s} RK4 step {step:.3f} ms | muladd (3 x 82 MB) {ma:6.1f} us = {3 * 8 * n / ma / 1e6:.2f} TB/s | copy 1 GiB {2 * (1 << 30) / cp / 1e6:.2f} TB/s | "
              f"axpy 1 GiB {3 * (1 << 30) / ax / 1e6:.2f} TB/s", flush=True)
