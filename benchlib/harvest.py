"""N > 1, after the headline's timed region: the other lines of the hot path on the SAME partition and communicator, so that the
ONE command the driver runs at N = 2 / 4 / 8 (``bench.py --gpus N --steps K --warmup W``, no extra flags) answers every question
a multi-GPU node can answer:

  mass            partitioned cell mass apply (row split: atomic-free kernel, numba-cpu/operators.py:19-68 + the scatters)
  rk4_geom        fused linear RK4 step, in-kernel geometry (the solver's default; cuda/demo_linear_box.py:537-553 per stage)
  rk4             the same step on the reference's G stream
  westervelt_geom Westervelt P = 6 step, 2/3 of the cells per direction (BASELINE config 5; cuda/demo_nonlinear_bowl.py:603-632)

each with ms per step (max over ranks, barrier + synchronise on both sides), the per-rank device time (min / max), its exposed halo
cost against the same work without any exchange, and its own halo check (+ the oracle check of the partitioned mass apply).  A
wall-clock budget (FUS_BENCH_HARVEST_BUDGET_S, default 150 s, decided by rank 0 for all ranks) skips what does not fit and says so."""
import argparse
import os
import time

import numpy as np

from .common import HBM_PEAK_GBS, coll_device, host_cores, log, rehearsal
from .cpu_legs import oracle_apply
from .roofline import mass_bytes_per_cell
from .steps import measure_rk4


def owned_dofs_check(y_d, y_ref_d, nl, health, device, dtype, what):
    """N > 1 form of the result check: the owned dofs of every rank against the oracle's apply over each rank's cells
    reverse-scattered (``y_ref_d``); collective, the same verdict on every rank."""
    import torch
    import torch.distributed as dist

    dd = (y_d[:nl].double() - y_ref_d[:nl].double())
    sums = torch.stack([(dd * dd).sum(), (y_ref_d[:nl].double() ** 2).sum(), y_d[:nl].double().sum(), y_ref_d[:nl].double().sum()]).to(coll_device(device))
    dist.all_reduce(sums)
    mx = torch.stack([dd.abs().max() if nl else dd.new_zeros(()), y_ref_d[:nl].double().abs().max() if nl else dd.new_zeros(()),
                      torch.tensor(float(health), dtype=torch.float64, device=device)]).to(coll_device(device))
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    rel_l2 = float(sums[0].sqrt().item()) / max(float(sums[1].sqrt().item()), 1e-300)
    rel_max = float(mx[0].item()) / max(float(mx[1].item()), 1e-300)
    tol_l2, tol_max = (1e-12, 1e-11) if dtype == "f64" else (1e-5, 1e-4)
    return {"rel_l2": rel_l2, "rel_max": rel_max, "sum_y": float(sums[2].item()), "sum_y_oracle": float(sums[3].item()),
            "norm_y_oracle": float(sums[1].sqrt().item()), "tol_rel_l2": tol_l2, "tol_rel_max": tol_max,
            "ok": bool(np.isfinite(rel_l2) and rel_l2 <= tol_l2 and rel_max <= tol_max and float(sums[1].item()) > 0 and float(mx[2].item()) == 0.0),
            "what": what, "oracle": "oracle/fus_oracle.c"}


def _agree(flag, device):
    """Rank 0's decision for all ranks (budget checks must not split the ranks before a collective)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=coll_device(device))
    dist.broadcast(t, src=0)
    return bool(t.item() == 1.0)


def partitioned_mass(args, rank, world, device, comm, mesh, dt, x_d, cc_d, y_d, dm_d, dphi_g, wts3, x, cc, ops, pre):
    """y += M(c) x over the partitioned mesh through HaloApply's row split (set A next to the exchanges, set B between them)."""
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    scat = fusgpu_loader.submodule("scatterer")
    P = args.degree
    n, T = P + 1, np.dtype(dt).itemsize
    detJ = torch.empty((mesh.ncells, n**3), dtype=x_d.dtype, device=device)
    pre.compute_scaled_jacobian_determinant_device(
        detJ, (torch.from_numpy(mesh.x_dofs).to(device), torch.from_numpy(mesh.x_g).to(device)), mesh.ncells,
        torch.from_numpy(dphi_g).to(device), torch.from_numpy(wts3).to(device))
    op = ops.mass_operator(n**3, dt)
    halo = scat.HaloApply(mesh, op, comm, dt, overlap=os.environ.get("FUS_HALO_OVERLAP", "1") != "0")
    halo.prepare(x_d, cc_d, detJ, dm_d)
    nl = mesh.nlocal
    # halo check of THIS line's closures: poisoned ghosts come back as their owners' values; the owned sum of one apply equals what the
    # cells of all ranks contribute, sum_c sum_i x detJ c, only if every ghost contribution reached its owner
    expect = x_d[nl:].clone()
    x_d[nl:] = -777.0
    halo.fwd(x_d)
    fwd_err = float((x_d[nl:] - expect).abs().max().item()) if expect.numel() else 0.0
    x_d[nl:] = expect
    ops.fill(0.0, y_d)
    halo.apply(x_d, cc_d, y_d, detJ, dm_d)
    ref = (x_d[dm_d.long()] * detJ * cc_d[:, None]).sum()
    sums = torch.stack([y_d[:nl].sum(), y_d[:nl].abs().sum(), ref, torch.tensor(float(halo.health()), dtype=x_d.dtype, device=device),
                        torch.tensor(fwd_err, dtype=x_d.dtype, device=device)]).to(coll_device(device))
    dist.all_reduce(sums)
    rel = abs(float(sums[0].item()) - float(sums[2].item())) / max(float(sums[1].item()), 1e-300)
    halo_check = {"forward_max_abs_err": float(sums[4].item()), "owned_sum_defect_over_sum_abs": rel, "device_wait_timeouts": int(sums[3].item()),
                  "ok": bool(float(sums[4].item()) == 0.0 and rel < (1e-9 if args.dtype == "f64" else 1e-3) and float(sums[3].item()) == 0.0)}
    # oracle check: one apply into a zeroed y (it is in y_d now), owned dofs against the oracle's apply over this rank's cells, reverse-scattered
    check = None
    if not args.no_check:
        y_loc = oracle_apply(P, mesh, None, x.astype(np.float64), cc.astype(np.float64), detJ.cpu().numpy().astype(np.float64), True, portable=True,
                             threads=max(1, host_cores() // max(1, world)))
        y_ref_d = torch.from_numpy(y_loc.astype(dt)).to(device)
        halo.rev(y_ref_d)
        torch.cuda.synchronize()
        check = owned_dofs_check(y_d, y_ref_d, nl, halo.health(), device, args.dtype,
                                 "one partitioned mass apply into a zeroed y, owned dofs of all ranks  vs  the oracle's apply over each rank's cells, reverse-scattered")
    K = max(1, args.steps)
    for _ in range(max(1, args.warmup)):
        halo.apply(x_d, cc_d, y_d, detJ, dm_d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K):
        halo.apply(x_d, cc_d, y_d, detJ, dm_d)
    e1.record()
    torch.cuda.synchronize()
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_device(device))
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms = float(el.item()) / K * 1e3
    every = [torch.zeros(1, dtype=torch.float64, device=coll_device(device)) for _ in range(world)]
    dist.all_gather(every, torch.tensor([e0.elapsed_time(e1) / K], dtype=torch.float64, device=coll_device(device)))
    per = [float(t.item()) for t in every]
    # ONE launch over all local rows, no exchange: the N = 1 kernel on this rank's cells
    op.prepare(dm_d) if hasattr(op, "prepare") else None
    op(x_d, cc_d, y_d, detJ, dm_d)
    e0.record()
    for _ in range(10):
        op(x_d, cc_d, y_d, detJ, dm_d)
    e1.record()
    torch.cuda.synchronize()
    kt = torch.tensor([e0.elapsed_time(e1) / 10], dtype=torch.float64, device=coll_device(device))
    dist.all_reduce(kt, op=dist.ReduceOp.MAX)
    kern_ms = float(kt.item())
    late = torch.tensor([float(halo.health())], dtype=torch.float64, device=coll_device(device))
    dist.all_reduce(late)
    rows = halo.row_split(dm_d, mesh.ndofs) is not None
    bpc = mass_bytes_per_cell(P, T)
    out = {"ms": ms, "rank_ms": [min(per), max(per)], "kernel_ms": kern_ms, "halo_exposed_ms": ms - kern_ms, "halo_exposed_frac": (ms - kern_ms) / kern_ms,
           "frac": mesh.ncells * bpc / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "value_dof_per_s": mesh.ndofs_global / (ms * 1e-3),
           "kernel": ops.mass_kernel_name(dm_d, mesh.ndofs, atomic=not rows), "split": "by dof (row sets A / B)" if rows else "by cell (float-atomic twin)",
           "schedule": halo.schedule_kind, "halo_check": halo_check, "check": check, "failed_waits_all_ranks": int(late.item()), "steps": K}
    torch.cuda.synchronize()
    dist.barrier()  # nobody frees an arena a neighbour may still write into
    halo.fwd.close(), halo.rev.close()
    del detJ
    return out


def _step_line(r):
    """The scalars of one measure_rk4 line the harvest keeps."""
    cfg, rf = r["config"], r["roofline"]
    rk = cfg.get("ranks_ms") or {}
    return {"ms": r["ms_per_step"], "rank_ms": [rk.get("min"), rk.get("max")], "local_ms": cfg.get("local_step_ms"),
            "halo_exposed_ms": cfg.get("halo_exposed_ms"), "halo_exposed_frac": cfg.get("halo_exposed_frac"), "frac": rf["frac"],
            "value_dof_steps_per_s": r["value"], "halo_check": cfg.get("halo_check"), "schedule": cfg.get("halo_schedule"), "steps": r["steps"],
            "degree": cfg["degree"], "cells_per_gpu": cfg["cells_per_gpu"], "global_dofs": cfg["global_dofs"], "geometry": cfg["geometry"]}


def harvest(args, rank, world, device, comm, mesh, dt, x_d, cc_d, y_d, dm_d, dphi_g, wts3, x, cc, ops, pre, budget_s=150.0, into=None):
    """The secondary lines of an N > 1 run; returns ``{name: line | {"skipped": reason} | {"error": ...}, "seconds": ...}`` (filled line by
    line into ``into`` where given: a run whose optional phases are cut short keeps what was finished)."""
    import torch
    import torch.distributed as dist

    t_start = time.perf_counter()
    out = into if into is not None else {}
    steps = max(1, min(args.steps, 20))
    wargs = argparse.Namespace(**{**vars(args), "degree": 6, "cells": max(4, round(args.cells * 2 / 3))})  # 54 -> 36: the same dof count per GPU

    def line_mass():
        return partitioned_mass(args, rank, world, device, comm, mesh, dt, x_d, cc_d, y_d, dm_d, dphi_g, wts3, x, cc, ops, pre)

    def line_step(a, mode, geo):
        return lambda: _step_line(measure_rk4(a, rank, world, device, mode, True, geo, steps, 2, comm=comm, cpu_leg=False, check=False))

    # most wanted first: config 5's step has no other driver-run command at N > 1
    plan = (("westervelt_geom", line_step(wargs, "westervelt", True), 45.0), ("rk4_geom", line_step(args, "rk4", True), 30.0),
            ("mass", line_mass, 15.0), ("rk4", line_step(args, "rk4", False), 30.0))
    for name, fn, need_s in plan:
        used = time.perf_counter() - t_start
        if not _agree(used + need_s <= budget_s or os.environ.get("FUS_BENCH_HARVEST_ALL") == "1", device):
            out[name] = {"skipped": f"budget: {used:.0f} s of {budget_s:.0f} s used, this line is allowed {need_s:.0f} s"}
            continue
        t0 = time.perf_counter()
        err = None
        try:
            out[name] = fn()
        except (Exception, SystemExit) as e:  # noqa: BLE001  (a failed halo check of one line ends that line, not the run)
            err = repr(e)
            log(f"rank {rank}: harvest line {name!r} failed: {err}")
        # a line that failed on ONE rank leaves the others inside its collectives: say so on every rank and stop harvesting
        bad = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(bad)
        if float(bad.item()) != 0.0:
            out[name] = {"error": err or "failed on another rank"}
            out["stopped_after"] = name
            break
        out[name]["seconds"] = time.perf_counter() - t0
        if rank == 0:
            o = out[name]
            log(f"harvest {name}: {o['ms']:.4f} ms/step (ranks {o['rank_ms'][0]:.4f} .. {o['rank_ms'][1]:.4f}), exposed halo "
                f"{(o.get('halo_exposed_ms') or float('nan')) * 1e3:+.1f} us, halo check {'ok' if (o.get('halo_check') or {}).get('ok') else 'FAILED'}")
    out["seconds"] = time.perf_counter() - t_start
    out["budget_s"] = budget_s
    if rehearsal():
        out["rehearsal"] = "ranks share the visible GPU(s): NOT a measurement"
    return out
