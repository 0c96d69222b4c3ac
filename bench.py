#!/usr/bin/env python3
"""
Headline benchmark (BASELINE.json): DOF/s of the fp64 stiffness-operator apply,
P = 4 hexahedral box, on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one apply ``y += K x`` over the whole (partitioned) mesh:
forward halo of x, stiffness kernel over all local cells, reverse halo of y
(the halo legs exist only for N > 1).  ``y`` is zeroed outside the timed region
and accumulated into, as in the reference's protocol
(numba-cpu/time_operators.py:227-233, cuda/time_operators.py:272-282).

Workload at N = 1: BASELINE config 3 -- P = 4, 54^3 = 157 464 perturbed
(non-affine) cells, 10 218 313 dofs, general per-quadrature-point G
(945 MB).  N > 1: weak scaling, 54^3 cells per GPU in 2x1x1 / 2x2x1 / 2x2x2
blocks (config 4 at N = 8: 108^3 cells, 81.2 M dofs).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: ``roofline`` (dominant kernel vs the HBM roofline, live HIP-event
timing) and ``cpu_baseline`` (the oracle's C restatement of the reference's
numba-cpu operator timed on this box's host cores; rank 0, N = 1 only).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def stiffness_bytes_per_cell(P, T):
    """Algorithmic HBM bytes per cell: G + dofmap + x once + y RMW + constant."""
    n = P + 1
    nd = n**3
    return 6 * nd * T + 4 * nd + T * P**3 + 2 * T * P**3 + T


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(P, pb, reps_omp=5, reps_serial=2):
    """Time the oracle (C restatement of numba-cpu/operators.py:71-227) on the host
    cores of this box, on the same mesh the GPU ran.  Reported, never shipped."""
    from oracle import oracle_c

    try:
        oracle_c.build(native=True)  # -march=native on the box that does the timing
        O = oracle_c.OracleLib(native=True)
    except Exception as e:  # no compiler on the box: fall back to the portable build
        log(f"native oracle build failed ({e}); using the portable build")
        O = oracle_c.OracleLib()
    mesh = pb["mesh"]
    ncores = min(O.max_threads(), host_cores())
    # oversubscription hurts a quota-limited box: take the best of a short thread sweep
    best = None
    for th in sorted({max(1, ncores // 2), ncores, min(2 * ncores, len(os.sched_getaffinity(0)))}):
        ysw = np.zeros(pb["mesh"].ndofs)
        O.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], ysw, pb["G"], pb["mesh"].dofmap, threads=th)
        t0 = time.perf_counter()
        O.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], ysw, pb["G"], pb["mesh"].dofmap, threads=th)
        dtm = time.perf_counter() - t0
        if best is None or dtm < best[1]:
            best = (th, dtm)
    ncores = best[0]
    y = np.zeros(mesh.ndofs)
    # bounded sample for the serial leg: a contiguous slab of cells
    ns = min(mesh.ncells, 40000)
    res = {}
    for name, threads, ncell, reps in (("omp", ncores, mesh.ncells, reps_omp), ("serial", 1, ns, reps_serial)):
        O.stiffness_apply(P, pb["D"], pb["x"], pb["cc"][:ncell], y, pb["G"][:ncell], mesh.dofmap[:ncell], threads=threads)
        ts = []
        for _ in range(reps):
            y[:] = 0.0
            t0 = time.perf_counter()
            O.stiffness_apply(P, pb["D"], pb["x"], pb["cc"][:ncell], y, pb["G"][:ncell], mesh.dofmap[:ncell], threads=threads)
            ts.append(time.perf_counter() - t0)
        dofs = ncell * P**3  # asymptotic dofs per cell, so slabs compare with the full box
        res[name] = dict(t=float(np.mean(ts)), dof_per_s=dofs / float(np.mean(ts)), ncell=int(ncell), threads=int(threads))
    return {
        "value": res["omp"]["dof_per_s"],
        "unit": "DOF/s",
        "cores": res["omp"]["threads"],
        "kind": "port",
        "sample": f"full workload ({res['omp']['ncell']} cells), {reps_omp} reps, OpenMP over {res['omp']['threads']} threads; "
        f"serial leg: {res['serial']['ncell']} cells x {reps_serial} reps",
        "single_thread_value": res["serial"]["dof_per_s"],
        "ms_per_apply": res["omp"]["t"] * 1e3,
        "impl": "oracle/fus_oracle.c (C restatement of numba-cpu/operators.py, -O3 -ffast-math -march=native)",
    }


def load_traffic(P, ncell):
    """Per-launch HBM bytes from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if int(t.get("P", -1)) == P and int(t.get("ncell", -1)) == ncell:
            return float(t["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def bench_rk4(args, rank, world, device):
    """Auxiliary metric (not the headline): full RK4 steps of the linear wave solver
    (BASELINE config 3: demo_linear_box, P = 4, ~10 M dofs per GPU), fused stage kernels."""
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    boxmesh, ls = fusgpu_loader.submodule("boxmesh"), fusgpu_loader.submodule("linear_solver")
    scat = fusgpu_loader.submodule("scatterer")
    P, L = args.degree, 0.12
    dt_np = np.float64 if args.dtype == "f64" else np.float32
    grid = boxmesh.default_grid(world)
    gcells = tuple(args.cells * g for g in grid)
    mesh = boxmesh.BoxMesh(P, gcells, grid=grid, rank=rank, length=tuple(L * g for g in grid), dtype=dt_np)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dts, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    comm = scat.TorchComm() if world > 1 else None
    if args.mode == "westervelt":  # BASELINE config 5 shape: Westervelt, bowl-warped trilinear cells
        nls = fusgpu_loader.submodule("nonlinear_solver")
        Lx = L * grid[0]

        def bowl(xg):
            out = xg.copy()
            yy, zz = xg[:, 1] / (L * grid[1]) - 0.5, xg[:, 2] / (L * grid[2]) - 0.5
            out[:, 0] = xg[:, 0] + 0.15 * (L / args.cells) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / Lx)
            return out

        mesh = boxmesh.BoxMesh(P, gcells, grid=grid, rank=rank, length=tuple(L * g for g in grid), dtype=dt_np, warp=bowl)
        solver = nls.WesterveltSpectral3D(mesh, dt_np, speed_of_sound=1500.0, source_frequency=0.5e6, comm=comm, fused=True)
        solver.affine = False
    else:
        solver = ls.LinearSpectral3D(mesh, dt_np, comm=comm, fused=True)
    solver.init()
    solver.rk4(0.0, tf, dts, max_steps=max(1, args.warmup))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.rk4(args.warmup * dts, tf, dts, max_steps=args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    out = {
        "metric": "rk4_step_dof_per_s" if args.mode == "rk4" else "westervelt_rk4_step_dof_per_s", "value": mesh.ndofs_global * args.steps / el, "unit": "DOF*steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("linear wave RK4 step (4 stages: stiffness + 2 facet mass + fused vector update + halo), "
                                if args.mode == "rk4" else
                                "Westervelt RK4 step (4 stages: fused cell pass [2 stiffness + 2 mass] + 2 facet mass + fused vector update + halo), ") +
                               f""
                               f"P={P}, {gcells[0]}x{gcells[1]}x{gcells[2]} cells, {mesh.ndofs_global} dofs",
                   "steps_to_final_time": nstep, "dt": dts,
                   "geometry": "affine box: constant-G fast path (opt-in, checked at set-up)" if solver.affine
                   else "general per-quadrature-point G"},
        "roofline": None, "cpu_baseline": None,
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--cells", type=int, default=54, help="cells per direction PER GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--xcd-remap", type=int, default=None)
    ap.add_argument("--no-plan", action="store_true", help="plan-free kernel (reads dofmap directly)")
    ap.add_argument("--mode", default="stiffness", choices=["stiffness", "rk4", "westervelt"],
                    help="stiffness: the headline metric; rk4 / westervelt: one full RK4 time step of the linear / "
                         "Westervelt solver per 'step' (auxiliary metrics)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import fusgpu_loader

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    ndev = max(torch.cuda.device_count(), 1)
    local_rank = local_rank % ndev  # a launcher that masks devices per rank leaves one visible device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("FUS_BENCH_FORCE_DIST", "0") == "1"  # exercise the N > 1 code path in a 1-rank world
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        opts = None
        try:  # comm kernels must get CUs while a chip-filling operator kernel runs
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:
            pass
        dist.init_process_group("nccl", device_id=device, pg_options=opts)

    lib = fusgpu_loader.submodule("_lib")
    ops = fusgpu_loader.submodule("operators")
    boxmesh = fusgpu_loader.submodule("boxmesh")
    gll = fusgpu_loader.submodule("gll")
    pre = fusgpu_loader.submodule("precompute")
    lib.load()
    if args.variant is not None:
        lib.set_tuning(lib.TUNE_STIFFNESS_VARIANT, args.variant)
    if args.no_plan:
        ops.use_plan(False)
    if args.xcd_remap is not None:
        lib.set_tuning(lib.TUNE_XCD_REMAP, args.xcd_remap)

    if args.mode in ("rk4", "westervelt"):
        return bench_rk4(args, rank, world, device)

    P = args.degree
    n = P + 1
    dt = np.float64 if args.dtype == "f64" else np.float32
    T = np.dtype(dt).itemsize
    grid = boxmesh.default_grid(world)
    gcells = tuple(args.cells * g for g in grid)

    t0 = time.time()
    mesh = boxmesh.BoxMesh(P, gcells, grid=grid, rank=rank, perturb=0.16, seed=0, dtype=dt)
    pts, wts, D = gll.tabulate_1d(P, dt)
    wts3 = gll.tensor_weights_3d(wts).astype(dt)
    dphi_g = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts), dt)
    xyz = mesh.dof_coordinates()
    x = (100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])).astype(dt)
    del xyz
    cc = np.random.default_rng(1234).standard_normal(mesh.ncells).astype(dt)

    x_d = torch.from_numpy(x).to(device)
    y_d = torch.zeros(mesh.ndofs, dtype=x_d.dtype, device=device)
    cc_d = torch.from_numpy(cc).to(device)
    dm_d = torch.from_numpy(mesh.dofmap).to(device)
    # geometry factors on the device (csrc/geometry.hpp; parity with the reference's precompute.py
    # is tested on the golden vectors): general per-quadrature-point G, no affine shortcut
    G_d = torch.empty((mesh.ncells, n**3, 6), dtype=x_d.dtype, device=device)
    pre.compute_scaled_geometrical_factor_device(
        G_d, (torch.from_numpy(mesh.x_dofs).to(device), torch.from_numpy(mesh.x_g).to(device)), mesh.ncells,
        torch.from_numpy(dphi_g).to(device), torch.from_numpy(wts3).to(device))
    torch.cuda.synchronize()
    if rank == 0:
        log(f"setup {time.time() - t0:.1f}s: P={P} cells/GPU={mesh.ncells} local dofs={mesh.ndofs} "
            f"global dofs={mesh.ndofs_global} grid={grid} G={G_d.numel() * T / 1e6:.0f} MB")
    op = ops.stiffness_operator(P, D.flatten(), dt)

    halo = None
    if use_dist:
        scat = fusgpu_loader.submodule("scatterer")
        halo = scat.HaloApply(mesh, op, scat.TorchComm(), dt, overlap=os.environ.get("FUS_HALO_OVERLAP", "1") != "0")

    def step():
        if halo is None:
            op(x_d, cc_d, y_d, G_d, dm_d)
        else:
            halo.apply(x_d, cc_d, y_d, G_d, dm_d)

    # set-up outside every step: batch plans, communicator bring-up
    if halo is None:
        op.prepare(dm_d)
    else:
        halo.prepare(x_d, cc_d, G_d, dm_d)
    for _ in range(args.warmup):
        step()
    y_d.zero_()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        step()
        ev1[i].record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    ev_ms = np.array([a.elapsed_time(b) for a, b in zip(ev0, ev1)])

    if halo is not None:
        # kernel time at N > 1: the three sub-launches (interior / boundary / interior) without
        # any exchange, timed after (outside) the timed region
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        halo.apply_local_only(x_d, cc_d, y_d, G_d, dm_d)
        e0.record()
        for _ in range(reps):
            halo.apply_local_only(x_d, cc_d, y_d, G_d, dm_d)
        e1.record()
        torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / reps
    else:
        kern_ms = float(ev_ms.mean())  # N = 1: the step IS the stiffness kernel launch

    # measured streaming ceiling of THIS device (outside the timed region): copy of 1 GiB -> 1 GiB
    # with the library's copy kernel (working set far beyond the 256 MiB Infinity Cache)
    copy_gbs = read_gbs = None
    try:
        nel = (1 << 30) // 8
        src = torch.empty(nel, dtype=torch.float64, device=device).fill_(1.0)
        dst = torch.empty_like(src)
        ops.copy(src, dst)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            ops.copy(src, dst)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * nel * 8 * 5 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        # read-only stream (the stiffness kernel is ~90 % reads): torch's reduction over 1 GiB
        src.sum()
        c0.record()
        for _ in range(5):
            src.sum()
        c1.record()
        torch.cuda.synchronize()
        read_gbs = nel * 8 * 5 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst
    except Exception as e:  # never let the side measurement break the bench line
        log(f"copy ceiling measurement failed: {e!r}")

    ndofs_global = mesh.ndofs_global
    value = ndofs_global / (elapsed / args.steps)
    bpc = stiffness_bytes_per_cell(P, T)
    achieved = mesh.ncells * bpc / (kern_ms * 1e-3) / 1e9

    out = {
        "metric": "stiffness_apply_dof_per_s",
        "value": value,
        "unit": "DOF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"stiffness apply y+=Kx, P={P} GLL hex box, {gcells[0]}x{gcells[1]}x{gcells[2]} perturbed cells, "
            f"{ndofs_global} dofs" + (" (BASELINE config 3)" if (world == 1 and P == 4 and args.cells == 54) else ""),
            "degree": P,
            "cells_per_gpu": mesh.ncells,
            "global_dofs": ndofs_global,
            "partition": f"{grid[0]}x{grid[1]}x{grid[2]} blocks",
            "geometry": "general per-quadrature-point G[ncell][n^3][6] (no affine shortcut)",
            "stiffness_kernel": "planned (batch plan, LDS pre-reduction)" if ops._USE_PLAN else f"plan-free variant {lib.get_tuning(lib.TUNE_STIFFNESS_VARIANT)}",
            "xcd_remap": lib.get_tuning(lib.TUNE_XCD_REMAP),
            "halo": None if halo is None else ("overlapped" if halo.overlap else "sequential"),
            "halo_exposed_ms": None if halo is None else max(0.0, ms_per_step - kern_ms),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": load_traffic(P, mesh.ncells),
            "kernel": "fus::stiffness_plan_kernel" if ops._USE_PLAN else "fus::stiffness_col_kernel",
            "kernel_ms": kern_ms,
            "step_ms_min": float(ev_ms.min()),
            "step_ms_std": float(ev_ms.std()),
            "algorithmic_bytes_per_cell": bpc,
            "cells_per_launch": mesh.ncells,
            "pct_of_hbm_roofline_dofs": 100.0 * achieved / HBM_PEAK_GBS,
            "measured_copy_gbs": copy_gbs,   # 1 GiB -> 1 GiB with fus_copy (read + write bytes)
            "measured_read_gbs": read_gbs,   # 1 GiB read-only reduction (torch.sum)
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            G = G_d.cpu().numpy()  # the CPU baseline streams the same G the GPU did
            pb = dict(mesh=mesh, D=D, x=x.astype(np.float64), cc=cc.astype(np.float64), G=G.astype(np.float64))
            if dt != np.float64:
                D = D.astype(np.float64)
                pb["D"] = D
            try:
                out["cpu_baseline"] = cpu_baseline(P, pb)
            except Exception as e:
                log(f"cpu_baseline failed: {e!r}")
                out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
