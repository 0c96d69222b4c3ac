"""The auxiliary lines of the default N = 1 run (``aux`` + their scalars in ``roofline.secondary``): cell mass apply (SURVEY 8d's
second operator line), the stiffness apply with G formed in the kernel, both fused RK4 steps, the Westervelt P = 6 steps, the
sustained headline, the halo proxy, the stand-alone scatters.  Each on its own bytes contract, each after the headline's timed
region -- and since round 6 each with a CHECK of what its own timed launches computed against the oracle (the reference keeps
cuda/test_operators.py:213-312 next to cuda/time_operators.py:204-290 on the same operators)."""
import argparse
import glob
import json
import os
import time

import numpy as np

from .common import HBM_PEAK_GBS, ROOT, kernel_src_sha, lib_built_from_tree, log
from .cpu_legs import compare_with_oracle, cpu_baseline_mass, oracle_apply
from .roofline import aux_traffic, geom_bytes_per_cell, mass_bytes_per_cell
from .steps import measure_rk4
from .transports import measure_halo_proxy, measure_scatter

AUX_STEADY_LAUNCHES, AUX_STEADY_WARM = 200, 100


def aux_tol(dtype):
    """Accumulated launches (hundreds of applies summed into one y): fp64 keeps SURVEY 8d's bar, fp32 gets the rounding of the sum."""
    return (1e-12, 1e-11) if dtype == "f64" else (3e-5, 3e-4)


def timed_steady(fn, burst):
    """(ms per launch in the steady state, ms per launch of the first burst, launches issued): ``burst`` back-to-back launches after 3
    untimed ones (what the aux lines timed until round 4), then AUX_STEADY_WARM untimed + AUX_STEADY_LAUNCHES timed launches between one
    HIP-event pair.  A burst of 20 launches of a 0.15 ms kernel is over in 3 ms -- inside the ramp of the device's clocks after the idle
    gap before it: the in-kernel-geometry kernel (VALU / LDS heavy) reads 0.16-0.185 ms in such a burst and 0.155 ms from the 50th launch
    on, whatever ran before (profiles/r05d_geom_variance_probe.log); a time loop runs in the steady state."""
    import torch

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(burst):
        fn()
    e1.record()
    torch.cuda.synchronize()
    first = e0.elapsed_time(e1) / burst
    for _ in range(AUX_STEADY_WARM):
        fn()
    e0.record()
    for _ in range(AUX_STEADY_LAUNCHES):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / AUX_STEADY_LAUNCHES, first, 3 + burst + AUX_STEADY_WARM + AUX_STEADY_LAUNCHES


def aux_mass(args, P, T, dt, mesh, x_d, cc_d, y_d, dm_d, dphi_g, wts3, device, ops, pre, x_host, cc_host, cpu_leg=True):
    """The cell mass apply y += M(c) x on the headline's mesh (numba-cpu/operators.py:19-68; shares the stiffness
    operator's batch plan): K back-to-back launches between one HIP-event pair, 3 044 B/cell at P = 4 / fp64.  Every form
    timed here (default, static-detJ, float-atomic twin) is checked: y is zeroed before its launches and y / launches is
    compared with ONE oracle apply."""
    import torch

    n = P + 1
    detJ = torch.empty((mesh.ncells, n**3), dtype=x_d.dtype, device=device)
    pre.compute_scaled_jacobian_determinant_device(
        detJ, (torch.from_numpy(mesh.x_dofs).to(device), torch.from_numpy(mesh.x_g).to(device)), mesh.ncells,
        torch.from_numpy(dphi_g).to(device), torch.from_numpy(wts3).to(device))
    mop = ops.mass_operator(n**3, dt)
    kname = ops.mass_kernel_name(dm_d, mesh.ndofs)
    K = AUX_STEADY_LAUNCHES  # steady state, like the other aux kernels (timed_steady)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    detJ_h = detJ.cpu().numpy().astype(np.float64)
    y_ref = None
    if not args.no_check:
        y_ref = oracle_apply(P, mesh, None, x_host.astype(np.float64), cc_host.astype(np.float64), detJ_h, True)

    def timed(fn, what):
        ops.fill(0.0, y_d)
        for _ in range(AUX_STEADY_WARM):
            fn(x_d, cc_d, y_d, detJ, dm_d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K):
            fn(x_d, cc_d, y_d, detJ, dm_d)
        e1.record()
        torch.cuda.synchronize()
        wall, ms = (time.perf_counter() - t0) / K * 1e3, e0.elapsed_time(e1) / K
        chk = None
        if y_ref is not None:
            count = AUX_STEADY_WARM + K
            chk = compare_with_oracle(y_d.cpu().numpy() / count, y_ref, args.dtype,
                                      f"y of the {count} launches of {what} (zeroed before them) / {count}  vs  one oracle mass apply, all {mesh.ndofs} dofs",
                                      tol=aux_tol(args.dtype))
        return wall, ms, chk

    # the float-atomic batch-plan kernel beside it (what the sub-launches of a partitioned apply use)
    _, atomic_ms, atomic_chk = timed(mop.atomic, "the float-atomic twin")
    # opt-in: detJ declared constant across applies -> streamed from a row-ordered copy instead of gathered through the entry ids
    static_ms, static_chk = None, None
    try:
        _, static_ms, static_chk = timed(ops.mass_operator(n**3, dt, static_detJ=True), "the static-detJ form")
    except Exception as e:  # noqa: BLE001
        log(f"aux mass, static-detJ form failed: {e!r}")
    wall_ms, ms, chk = timed(mop, "the default operator")
    bpc = mass_bytes_per_cell(P, T)
    achieved = mesh.ncells * bpc / (ms * 1e-3) / 1e9
    traffic, traffic_source = None, "no PMC pass of the mass kernel in profiles/traffic_latest.json"
    try:  # replayed like the headline's: only if the mass kernel's sources and flags are the profiled ones
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            tm = json.load(f).get("aux", {}).get("mass")
        if tm and int(tm["P"]) == P and int(tm["ncell"]) == mesh.ncells and tm.get("dtype", "f64") == args.dtype:
            files = tuple(tm.get("kernel_src_files", ("plan.hpp", "mass.hpp")))
            if tm.get("kernel", "fus::mass_plan_kernel") != kname:
                traffic_source = f"the profiled kernel was {tm.get('kernel', 'fus::mass_plan_kernel')}, this run launches {kname}"
            elif tm.get("kernel_src_sha") == kernel_src_sha(files) and lib_built_from_tree():
                traffic = float(tm["hbm_bytes_per_launch"])
                traffic_source = f"replayed from {tm['source']} (rocprofv3 --pmc; same kernel sources and compile flags)"
            else:
                traffic_source = "the mass kernel's sources differ from the profiled ones"
    except Exception:
        pass
    checks = [c for c in (chk, static_chk, atomic_chk) if c is not None]
    out = {"metric": "mass_apply_dof_per_s", "value": mesh.ndofs_global / (wall_ms * 1e-3), "unit": "DOF/s", "ms_per_step": wall_ms, "steps": K,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_source, "kernel": kname, "kernel_ms": ms,
                        "kernel_ms_how": f"one HIP-event pair around {AUX_STEADY_LAUNCHES} back-to-back launches after {AUX_STEADY_WARM} untimed ones",
                        "algorithmic_bytes_per_cell": bpc, "cells_per_launch": mesh.ncells,
                        "atomic_kernel": "fus::mass_plan_kernel", "atomic_kernel_ms": atomic_ms,
                        "atomic_kernel_frac": mesh.ncells * bpc / (atomic_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        # mass_operator(N, T, static_detJ=True): same operator, same sums, priced on the SAME algorithmic bytes although
                        # it reads fewer (2 index bytes per entry instead of 4): opt-in, the caller promises a constant detJ
                        "static_detJ_kernel_ms": static_ms,
                        "static_detJ_frac": None if not static_ms else mesh.ncells * bpc / (static_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "cpu_baseline": None,
           # the default operator's check; the other two forms beside it; ``ok`` = every form timed here
           "check": None if chk is None else dict(chk, ok=all(c["ok"] for c in checks), static_detJ_rel_l2=None if static_chk is None else static_chk["rel_l2"],
                                                  atomic_rel_l2=None if atomic_chk is None else atomic_chk["rel_l2"])}
    if cpu_leg:
        out["cpu_baseline"] = cpu_baseline_mass(P, mesh, x_host.astype(np.float64), cc_host.astype(np.float64), detJ_h)
    # the same operator in cached-diagonal form (opt-in, own contract: 3 vector touches per dof)
    dmo = ops.diagonal_mass_operator(cc_d, detJ, dm_d, mesh.ndofs, dt)
    ops.fill(0.0, y_d)
    for _ in range(3):
        dmo(x_d, y_d)
    e0.record()
    for _ in range(K):
        dmo(x_d, y_d)
    e1.record()
    torch.cuda.synchronize()
    msd = e0.elapsed_time(e1) / K
    dchk = None
    if y_ref is not None:
        dchk = compare_with_oracle(y_d.cpu().numpy() / (K + 3), y_ref, args.dtype,
                                   f"y of the {K + 3} launches of the cached-diagonal form / {K + 3}  vs  one oracle mass apply", tol=aux_tol(args.dtype))
    ach = 3 * T * mesh.ndofs / (msd * 1e-3) / 1e9
    diag = {"metric": "mass_apply_cached_diagonal_dof_per_s", "value": mesh.ndofs_global / (msd * 1e-3), "unit": "DOF/s", "ms_per_step": msd, "steps": K,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "fus::muladd_kernel", "kernel_ms": msd, "algorithmic_bytes_per_launch": 3 * T * mesh.ndofs,
                         "bytes_contract": "y += (M(c) 1) (.) x with w = M(c) 1 assembled once: 3 vector touches per dof (opt-in; not the reference's gather-scale-scatter)"},
            "cpu_baseline": None, "check": dchk}
    return out, diag


def measure_sustained(step_fn, alg_bytes, total=2500, windows=10):
    """>= 0.5 s of back-to-back headline applies: ms per apply overall and per sub-window (one HIP event between two
    windows), device clocks before / after where sysfs shows them."""
    import torch

    def clocks():
        # sysfs, read in-process: rocm-smi is a '#!/usr/bin/env python3' script, and starting it from a process that has
        # initialised the GPU (or under a profiler's preloaded library) is the exec hop this pool forbids
        out = {}
        try:
            for card in sorted(glob.glob("/sys/class/drm/card*/device")):
                for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
                    try:
                        with open(os.path.join(card, name)) as f:
                            cur = [ln.split(":", 1)[1].strip().rstrip("*").strip() for ln in f.read().splitlines() if ln.rstrip().endswith("*")]
                    except OSError:
                        continue
                    if cur:
                        out[name[7:]] = cur[0]
                if out:
                    break
        except Exception:  # noqa: BLE001
            return None
        return out or None

    per = max(1, total // windows)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(windows + 1)]
    c0 = clocks()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for w in range(windows):
        for _ in range(per):
            step_fn()
        ev[w + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c1 = clocks()
    win = [ev[i].elapsed_time(ev[i + 1]) / per for i in range(windows)]
    ms = ev[0].elapsed_time(ev[windows]) / (per * windows)
    return {"applies": per * windows, "seconds": wall, "ms_per_apply": ms, "window_applies": per, "window_ms_per_apply_min": float(min(win)),
            "window_ms_per_apply_max": float(max(win)), "window_ms_per_apply": [float(v) for v in win],
            "frac_of_hbm_roofline": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "clocks_before": c0, "clocks_after": c1}


def aux_geom(args, c, ops, y_oracle):
    """the headline's apply with G formed in the kernel -- own bytes contract, own line; checked against the oracle's
    K x (the oracle reads the reference's G array: the first DIRECT oracle comparison of the solvers' default kernel at 10 M dofs)."""
    P, T, mesh = c.P, c.T, c.mesh
    gop = ops.stiffness_operator(P, c.D.flatten(), c.dt, geometry=(mesh.x_dofs, mesh.x_g, c.pts, c.wts))
    gop.prepare(c.dm_d) if hasattr(gop, "prepare") else None
    K = AUX_STEADY_LAUNCHES
    ops.fill(0.0, c.y_d)
    gms, gms_burst, count = timed_steady(lambda: gop(c.x_d, c.cc_d, c.y_d, None, c.dm_d), max(1, args.steps))
    chk = None
    if y_oracle is not None:
        chk = compare_with_oracle(c.y_d.cpu().numpy() / count, y_oracle, args.dtype,
                                  f"y of the {count} launches of this line (zeroed before them) / {count}  vs  one oracle apply on the reference's G array, all {mesh.ndofs} dofs",
                                  tol=aux_tol(args.dtype))
    gb = geom_bytes_per_cell(P, T)
    gach = mesh.ncells * gb / (gms * 1e-3) / 1e9
    gtr, gtr_src = aux_traffic("stiffness_in_kernel_geometry", P, mesh.ncells, args.dtype)
    return {"metric": "stiffness_apply_in_kernel_geometry_dof_per_s", "value": mesh.ndofs_global / (gms * 1e-3), "unit": "DOF/s", "ms_per_step": gms, "steps": K,
            "roofline": {"bound": "hbm", "achieved": gach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gach / HBM_PEAK_GBS, "traffic": gtr, "traffic_source": gtr_src,
                         "kernel": "fus::stiffness_plan_geom_kernel", "kernel_ms": gms,
                         "kernel_ms_how": f"one HIP-event pair around {AUX_STEADY_LAUNCHES} back-to-back launches after {AUX_STEADY_WARM} untimed ones (steady state)",
                         "kernel_ms_first_burst": gms_burst, "first_burst_launches": max(1, args.steps),
                         "algorithmic_bytes_per_cell": gb, "cells_per_launch": mesh.ncells,
                         "bytes_contract": "no G array: dofmap + x once + y RMW + constant + 8 vertex ids + one vertex per cell (DESIGN 3.2); NOT the headline contract",
                         "bound_note": "latency + vector ALU issue + float-atomic request rate together, not HBM bytes (DESIGN 3.2)"},
            "cpu_baseline": None, "check": chk}


def aux_lines(args, rank, world, device, c, ops, pre, y_oracle):
    """``aux`` of the default N = 1 line.  ``c``: the headline's context (mesh, device arrays, tables, ``step``, ``alg_bytes``);
    ``y_oracle``: the oracle's K x on the headline's inputs (None with --no-check).  An auxiliary line never breaks the headline:
    a failure is logged and its entry is None; a line whose CHECK fails stays in the output with ``check.ok = false`` (the caller
    marks the run invalid and exits non-zero)."""
    P, mesh = c.P, c.mesh
    aux = {}
    try:
        aux["mass"], aux["mass_cached_diagonal"] = aux_mass(args, P, c.T, c.dt, mesh, c.x_d, c.cc_d, c.y_d, c.dm_d, c.dphi_g, c.wts3, device, ops, pre,
                                                            c.x, c.cc, cpu_leg=not args.no_cpu_baseline)
    except Exception as e:  # noqa: BLE001
        log(f"aux mass line failed: {e!r}")
        aux["mass"] = None
    try:  # >= 0.5 s of back-to-back headline applies (the timed region is K launches: milliseconds)
        aux["sustained"] = measure_sustained(c.step, c.alg_bytes)
    except Exception as e:  # noqa: BLE001
        log(f"aux sustained line failed: {e!r}")
        aux["sustained"] = None
    try:  # north_star's "< 5 % halo overhead" on the one-GPU proxy: paired single launch | schedule | schedule + exchanges
        if mesh.ncells == args.cells**3:
            aux["halo_proxy"] = measure_halo_proxy(c.op, mesh, c.cc_d, c.G_d, c.dm_d, c.y_d, device, c.dt, P, args.cells)
    except Exception as e:  # noqa: BLE001
        log(f"aux halo_proxy line failed: {e!r}")
        aux["halo_proxy"] = None
    keys = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "cpu_baseline", "check")
    try:
        aux["stiffness_in_kernel_geometry"] = aux_geom(args, c, ops, y_oracle)
    except Exception as e:  # noqa: BLE001
        log(f"aux in-kernel-geometry line failed: {e!r}")
        aux["stiffness_in_kernel_geometry"] = None
    for name, geo_k in (("rk4_step", False), ("rk4_step_in_kernel_geometry", True)):
        try:
            r = measure_rk4(args, rank, world, device, "rk4", True, geo_k, max(1, min(args.steps, 20)), 2, cpu_leg=not args.no_cpu_baseline,
                            check=not args.no_check)
            aux[name] = {k: r.get(k) for k in keys}
        except Exception as e:  # noqa: BLE001
            log(f"aux {name} line failed: {e!r}")
            aux[name] = None
    try:  # BASELINE config 5's step on one GPU: Westervelt, P = 6, 36^3 bowl-warped cells (10.2 M dofs), fused stage
        wargs = argparse.Namespace(**{**vars(args), "degree": 6, "cells": max(4, round(args.cells * 2 / 3))})  # 54 -> 36: the same dof count
        # the reference's G stream | the solver's default (in-kernel geometry, two-gather cell pass: any medium) | the single-gather form
        for name, geo_k, single in (("westervelt_step", False, False), ("westervelt_step_in_kernel_geometry", True, False),
                                    ("westervelt_step_single_gather", True, True)):
            try:
                r = measure_rk4(wargs, rank, world, device, "westervelt", True, geo_k, max(1, min(args.steps, 20)), 2, cpu_leg=False, single_gather=single,
                                check=not args.no_check)
                aux[name] = {k: r.get(k) for k in keys}
            except Exception as e:  # noqa: BLE001
                log(f"aux {name} line failed: {e!r}")
                aux[name] = None
    except Exception as e:  # noqa: BLE001
        log(f"aux westervelt_step lines failed: {e!r}")
        aux["westervelt_step"] = None
    try:  # the reference's third timing script (numba-cpu/time_scatterer.py), self-neighbour with config-4 messages
        aux["scatter"] = measure_scatter(device, c.dt, reps=100, P=P, cells=args.cells)
    except Exception as e:  # noqa: BLE001
        log(f"aux scatter line failed: {e!r}")
        aux["scatter"] = None
    return aux


def failed_aux_checks(aux):
    """Names of the aux lines whose own check did not pass."""
    return [k for k, v in (aux or {}).items() if isinstance(v, dict) and isinstance(v.get("check"), dict) and not v["check"].get("ok", False)]
