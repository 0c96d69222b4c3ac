"""The time-step modes: full fused RK4 steps of the linear (BASELINE config 3) and Westervelt (config 5 shape) solvers.
Since round 6 every line carries (i) a CHECK of the field the GPU stepped against the oracle's time loop on the same mesh
(N = 1; cuda/test_operators.py:248,312 keeps its checks next to its timings the same way), (ii) at N > 1 the per-rank
device time (min / max over ranks) and the step's exposed halo cost: the same steps once more with the exchange switched
off (one launch over all local cells, no scatter) on every rank."""
import time

import numpy as np

from .common import HBM_PEAK_GBS, coll_device, emit, host_cores, lib_built_from_tree, lib_sha, log, rehearsal
from .roofline import aux_traffic, rk4_step_bytes, rk4_step_traffic
from .cpu_legs import compare_with_oracle, cpu_baseline_rk4, oracle_step_field
from .transports import first_comm


def measure_rk4(args, rank, world, device, mode, perturbed, in_kernel_geometry, steps, warmup, comm=None, cpu_leg=False, single_gather=False, check=True):
    """Full RK4 steps of the linear (BASELINE config 3: demo_linear_box, P = 4, ~10 M dofs per GPU) or Westervelt
    (config 5 shape) solver, fused stage kernels; returns the bench line as a dict."""
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    boxmesh, ls = fusgpu_loader.submodule("boxmesh"), fusgpu_loader.submodule("linear_solver")
    P, L = args.degree, 0.12
    dt_np = np.float64 if args.dtype == "f64" else np.float32
    T = np.dtype(dt_np).itemsize
    grid = boxmesh.default_grid(world)
    gcells = tuple(args.cells * g for g in grid)
    mesh = boxmesh.BoxMesh(P, gcells, grid=grid, rank=rank, length=tuple(L * g for g in grid), dtype=dt_np)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L * grid[0])
    if world > 1:  # comm.Allreduce(hmin, mesh_size, op=MPI.MIN), cuda/demo_linear_box.py:108
        hm = torch.tensor([h], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(hm, op=dist.ReduceOp.MIN)
        h = float(hm.item())
    dts, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L * grid[0])  # the wave crosses the whole (partitioned) box
    if warmup + steps > nstep:
        raise SystemExit(f"--warmup + --steps = {warmup + steps} exceeds the {nstep} steps to the final time")
    want_single_gather, single_gather = bool(single_gather), False
    if mode == "westervelt":  # BASELINE config 5 shape: Westervelt, bowl-warped trilinear cells
        nls = fusgpu_loader.submodule("nonlinear_solver")
        Lx = L * grid[0]

        def bowl(xg):
            out = xg.copy()
            yy, zz = xg[:, 1] / (L * grid[1]) - 0.5, xg[:, 2] / (L * grid[2]) - 0.5
            out[:, 0] = xg[:, 0] + 0.15 * (L / args.cells) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / Lx)
            return out

        mesh = boxmesh.BoxMesh(P, gcells, grid=grid, rank=rank, length=tuple(L * g for g in grid), dtype=dt_np, warp=bowl)
        # default: the two-gather cell pass (what every medium takes since round 5; a heterogeneous one has no choice);
        # single_gather: the form a uniform c4 / c3 allows (the vector pass writes w = u_n + kappa v_n, the cell pass is a plain apply)
        solver = nls.WesterveltSpectral3D(mesh, dt_np, speed_of_sound=1500.0, source_frequency=0.5e6, comm=comm, fused=True,
                                          in_kernel_geometry=in_kernel_geometry, uniform_ratio=True if want_single_gather else "auto")
        solver.affine = False
        single_gather = solver.kappa is not None
    else:
        if perturbed:  # non-affine cells: general G, or G formed in the kernel
            mesh = boxmesh.BoxMesh(P, gcells, grid=grid, rank=rank, length=tuple(L * g for g in grid), dtype=dt_np,
                                   perturb=0.16, seed=0)
        solver = ls.LinearSpectral3D(mesh, dt_np, comm=comm, fused=True, in_kernel_geometry=in_kernel_geometry)
    solver.init()
    halo_check = None
    if world > 1 and getattr(solver, "halo", None) is not None:
        # the exchange this solver will use, checked before anything is timed: every ghost must come back from a forward
        # scatter holding its owner's value (the global lexicographic id, exact in floating point), and the reverse
        # scatter of "1 in every ghost" must leave on every owned dof the number of ranks that ghost it, whose global sum is
        # the global number of ghosts
        tdt = torch.float64 if dt_np == np.float64 else torch.float32
        lex = torch.from_numpy(mesh.global_lexicographic_ids().astype(dt_np)).to(device) % 8191.0  # exact in fp32 too
        v = lex.clone()
        v[mesh.nlocal:] = -1.0
        solver.halo.fwd(v)
        bad = float((v != lex).sum().item())
        w = torch.zeros(mesh.ndofs, dtype=tdt, device=device)
        w[mesh.nlocal:] = 1.0
        solver.halo.rev(w)
        sums = torch.tensor([bad, float(w[: mesh.nlocal].sum().item()), float(mesh.ndofs - mesh.nlocal), float(solver.halo.health())],
                            dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(sums)
        halo_check = {"forward_wrong_ghosts": int(sums[0].item()), "reverse_sum": float(sums[1].item()), "global_ghosts": float(sums[2].item()),
                      "device_wait_timeouts": int(sums[3].item())}
        halo_check["ok"] = bool(sums[0].item() == 0 and sums[1].item() == sums[2].item() and sums[3].item() == 0)
        if not halo_check["ok"]:
            raise SystemExit(f"rank {rank}: halo check failed: {halo_check}")
    nwarm = max(1, warmup)
    solver.rk4(0.0, tf, dts, max_steps=nwarm)
    # the field after the warm-up steps (from u = v = 0 at t = 0): what the oracle's time loop is compared with below
    u_warm = solver.u[: mesh.nlocal].clone() if (check and world == 1) else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    _, steps_done = solver.rk4(warmup * dts, tf, dts, max_steps=steps)
    e1.record()
    assert steps_done == steps, (steps_done, steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    if world > 1 and getattr(solver, "halo", None) is not None:
        # the solver's exchanges are not re-derived here; a device-side wait that gave up invalidates the run
        late = torch.tensor([float(solver.halo.health())], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(late)
        if float(late.item()) != 0.0:
            raise SystemExit(f"rank {rank}: {int(late.item())} device-side halo wait(s) timed out: the run is invalid")
    dev_ms = e0.elapsed_time(e1) / steps  # device time of the region on the launch stream
    ranks_ms = local_ms = None
    if world > 1:
        every = [torch.zeros(1, dtype=torch.float64, device=coll_device(device)) for _ in range(world)]
        dist.all_gather(every, torch.tensor([dev_ms], dtype=torch.float64, device=coll_device(device)))
        per = [float(t.item()) for t in every]
        ranks_ms = {"min": min(per), "max": max(per), "per_rank": per, "what": "HIP-event time of the K steps on each rank's launch stream / K"}
        if getattr(solver, "halo", None) is not None:
            # the same steps with the exchange switched off: ONE launch over all local cells per stage, no scatter -- the
            # N = 1 step on this rank's cells (the field is not a solution any more; it is not looked at again)
            h, solver.halo = solver.halo, None
            try:
                solver.rk4(0.0, tf, dts, max_steps=1)
                torch.cuda.synchronize()
                e0.record()
                solver.rk4(0.0, tf, dts, max_steps=steps)
                e1.record()
                torch.cuda.synchronize()
                lt = torch.tensor([e0.elapsed_time(e1) / steps], dtype=torch.float64, device=coll_device(device))
                dist.all_reduce(lt, op=dist.ReduceOp.MAX)
                local_ms = float(lt.item())
            finally:
                solver.halo = h
    geo_kernel = bool(getattr(solver, "in_kernel_geometry", False))
    model = rk4_step_bytes(P, T, mesh.ncells, mesh.ndofs, int(solver.fdm1.shape[0]), int(solver.fdm2.shape[0]), mode,
                           bool(solver.affine), geo_kernel, single_gather, lean=bool(getattr(solver, "lean_stages", False)))
    achieved = model["bytes_per_step"] / (dev_ms * 1e-3) / 1e9
    if mode == "rk4" and perturbed and world == 1:
        traffic, traffic_source = rk4_step_traffic(P, mesh.ncells, args.dtype, geo_kernel)
    elif mode == "westervelt" and world == 1:
        traffic, traffic_source = aux_traffic("westervelt_step" + ("_in_kernel_geometry" if geo_kernel else "") + ("_single_gather" if single_gather else ""),
                                              P, mesh.ncells, args.dtype)
    else:
        traffic, traffic_source = None, "no PMC passes replayed for this configuration of the step"
    cpu = None
    if cpu_leg and mode == "rk4" and world == 1 and not geo_kernel and dt_np == np.float64:
        try:
            cpu = cpu_baseline_rk4(P, mesh, solver, dts, steps=nwarm)  # its field is kept: the check below reads it
        except Exception as e:  # noqa: BLE001
            log(f"rk4 cpu_baseline failed: {e!r}")
    chk = None
    if u_warm is not None:
        try:
            u_ref = oracle_step_field(mode, P, mesh, solver, nwarm, dts, device)
            chk = compare_with_oracle(u_warm.cpu().numpy(), u_ref[: mesh.nlocal], args.dtype,
                                      f"u after the {nwarm} warm-up steps of this run (from u = v = 0)  vs  the oracle's time loop on the same mesh, {mesh.nlocal} dofs",
                                      tol=(1e-11, 1e-10) if args.dtype == "f64" else (1e-4, 1e-3))
            chk["oracle"] = ("oracle/rk4_oracle.py solve" if mode == "rk4" else "oracle/rk4_oracle.py solve_westervelt") + " over oracle/fus_oracle.c (pinned: tests/golden/rk4*.npz)"
        except Exception as e:  # noqa: BLE001
            log(f"{mode} step check could not run: {e!r}")
            chk = {"ok": False, "error": repr(e), "rel_l2": float("nan")}
        del u_warm
    out = {
        "metric": "rk4_step_dof_per_s" if mode == "rk4" else "westervelt_rk4_step_dof_per_s", "value": mesh.ndofs_global * steps / el, "unit": "DOF*steps/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("linear wave RK4 step (4 stages: stiffness + 2 facet mass + fused vector update + halo), "
                                if mode == "rk4" else
                                "Westervelt RK4 step (4 stages: cell pass [stiffness part; mass terms are diagonal products in the vector pass] + 2 facet mass + fused vector update + halo), ") +
                               f"P={P}, {gcells[0]}x{gcells[1]}x{gcells[2]} cells, {mesh.ndofs_global} dofs",
                   "degree": P, "cells_per_gpu": mesh.ncells, "global_dofs": mesh.ndofs_global,
                   "steps_to_final_time": nstep, "dt": dts,
                   "geometry": "affine box: constant-G fast path (opt-in, checked at set-up)" if solver.affine
                   else ("G (and detJ) formed in the cell kernel from the vertices (the solvers' default on non-affine cells of degree >= 3)" if geo_kernel else "general per-quadrature-point G"),
                   "halo_check": halo_check, "halo_schedule": getattr(getattr(solver, "halo", None), "schedule_kind", None),
                   # N > 1: the step against the same steps without any exchange (one launch over all local cells per stage)
                   "local_step_ms": local_ms, "halo_exposed_ms": None if local_ms is None else el / steps * 1e3 - local_ms,
                   "halo_exposed_frac": None if local_ms is None else (el / steps * 1e3 - local_ms) / local_ms, "ranks_ms": ranks_ms,
                   "lib_sha": lib_sha(), "lib_built_from_tree": lib_built_from_tree()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "whole fused RK4 step: 4 x (cell pass + facet_terms_kernel + rk4_stage kernel" + (", LEAN stage kinds 4-7)" if getattr(solver, "lean_stages", False) else ")"),
                     "kernel_ms": dev_ms, "kernel_ms_how": "one HIP-event pair around the K steps of the timed region / K",
                     "algorithmic_bytes_per_step": model["bytes_per_step"], "cell_pass_bytes_per_cell": model["cell_pass_bytes_per_cell"],
                     "vector_touches_per_step": model["vector_touches_per_step"], "cells_per_launch": mesh.ncells},
        "cpu_baseline": cpu, "check": chk,
    }
    if chk is not None and not chk["ok"]:
        out["valid"] = False
    if rehearsal():
        out.update(valid=False, rehearsal="ranks share the visible GPU(s): NOT a measurement")
    if world > 1:  # nobody frees an arena a neighbour may still write into
        torch.cuda.synchronize()
        dist.barrier()
    del solver
    return out


def bench_rk4(args, rank, world, device):
    """Auxiliary metric (not the headline): ``--mode rk4`` / ``--mode westervelt``."""
    import torch.distributed as dist

    import fusgpu_loader

    scat = fusgpu_loader.submodule("scatterer")
    comm = first_comm(args, scat, world, device)[0] if world > 1 else None
    out = measure_rk4(args, rank, world, device, args.mode, args.perturbed, args.in_kernel_geometry, args.steps, args.warmup, comm,
                      cpu_leg=not args.no_cpu_baseline, single_gather=args.single_gather, check=not args.no_check)
    if rank == 0:
        emit(out)
    if world > 1:
        dist.destroy_process_group()
    if out.get("check") is not None and not out["check"]["ok"]:
        log(f"RESULT CHECK FAILED: {out['check']}")
        raise SystemExit(3)
