"""The CPU legs: the oracle (oracle/: C restatement of numba-cpu/operators.py, test infrastructure) timed on the box's host
cores as ``cpu_baseline``, and used as the CHECKER of what the timed regions computed.  Never the thing measured or shipped."""
import os
import time

import numpy as np

from .common import host_cores, log


def cpu_baseline(P, pb, reps_omp=60, reps_serial=10):
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
    # threads = the cores this process may really use (cgroup quota, affinity mask): no oversubscription -- a box with a
    # CPU quota throttles the whole group for the rest of the period once the quota is spent, which is what made
    # this number jump between boxes.  Threads are pinned (OMP_PROC_BIND=close, OMP_PLACES=cores, set in main()
    # before any OpenMP runtime is loaded).
    quota, affinity = host_cores(), len(os.sched_getaffinity(0))
    ncores = max(1, min(O.max_threads(), quota))
    y = np.zeros(mesh.ndofs)
    # bounded sample (a few seconds of CPU work in all): whole workload for both legs up to config-3 size,
    # a contiguous slab of cells beyond
    ns = min(mesh.ncells, 160000)
    res = {}
    for name, threads, ncell, reps in (("omp", ncores, mesh.ncells, reps_omp), ("serial", 1, ns, reps_serial)):
        for _ in range(2):
            O.stiffness_apply(P, pb["D"], pb["x"], pb["cc"][:ncell], y, pb["G"][:ncell], mesh.dofmap[:ncell], threads=threads)
        ts = []
        for _ in range(reps):
            y[:] = 0.0
            t0 = time.perf_counter()
            O.stiffness_apply(P, pb["D"], pb["x"], pb["cc"][:ncell], y, pb["G"][:ncell], mesh.dofmap[:ncell], threads=threads)
            ts.append(time.perf_counter() - t0)
        if name == "omp":
            pb["y_oracle"] = y.copy()  # K x of the whole mesh (y is zeroed before every rep): what result_check compares with
        dofs = ncell * P**3  # asymptotic dofs per cell, so slabs compare with the full box
        res[name] = dict(t=float(np.median(ts)), tmin=float(np.min(ts)), mean=float(np.mean(ts)), std=float(np.std(ts)),
                         dof_per_s=dofs / float(np.median(ts)), ncell=int(ncell), threads=int(threads), dofs=dofs)
    noisy = res["omp"]["std"] > 0.3 * res["omp"]["t"]
    return {
        "value": res["omp"]["dof_per_s"],
        "unit": "DOF/s",
        "cores": res["omp"]["threads"],
        "kind": "port",
        "sample": f"full workload ({res['omp']['ncell']} cells), median of {reps_omp} reps, OpenMP over {res['omp']['threads']} pinned threads; "
        f"serial leg: {res['serial']['ncell']} cells x {reps_serial} reps",
        # what the reference's njit loop (no parallel=True) and its serial C++ loop actually are: ONE thread.  This is the
        # stated reference-equivalent baseline; the OpenMP figure above is more than the reference does.
        "single_thread_value": res["serial"]["dof_per_s"],
        "single_thread_ms_per_apply": res["serial"]["t"] * 1e3,
        "value_best_rep": res["omp"]["dofs"] / res["omp"]["tmin"],
        "ms_per_apply": res["omp"]["t"] * 1e3,
        "ms_per_apply_min": res["omp"]["tmin"] * 1e3,
        "ms_per_apply_mean": res["omp"]["mean"] * 1e3,
        "ms_per_apply_std": res["omp"]["std"] * 1e3,
        "noisy": bool(noisy),  # std / median > 0.3: the OpenMP figure of this box is not to be trusted to better than that
        "quota_cores": quota,
        "affinity_cores": affinity,
        "omp_proc_bind": os.environ.get("OMP_PROC_BIND"),
        "impl": "oracle/fus_oracle.c (C restatement of numba-cpu/operators.py, -O3 -ffast-math -march=native)",
    }


def cpu_baseline_mass(P, mesh, x, cc, detJ, reps=5):
    """The oracle's cell mass apply (C restatement of numba-cpu/operators.py:19-68), serial as the
    reference runs it, on the whole workload."""
    from oracle import oracle_c

    try:
        oracle_c.build(native=True)
        O = oracle_c.OracleLib(native=True)
    except Exception as e:
        log(f"native oracle build failed ({e}); using the portable build")
        O = oracle_c.OracleLib()
    y = np.zeros(mesh.ndofs)
    O.mass_apply(x, cc, y, detJ, mesh.dofmap)
    ts = []
    for _ in range(reps):
        y[:] = 0.0
        t0 = time.perf_counter()
        O.mass_apply(x, cc, y, detJ, mesh.dofmap)
        ts.append(time.perf_counter() - t0)
    t = float(np.mean(ts))
    return {"value": mesh.ndofs / t, "unit": "DOF/s", "cores": 1, "kind": "port",
            "sample": f"full workload ({mesh.ncells} cells), {reps} reps, one thread (the reference's njit loop is serial)",
            "ms_per_apply": t * 1e3, "impl": "oracle/fus_oracle.c oracle_mass_apply_f64"}


_STEP_FIELDS = {}  # (mode, P, cells, dofs, steps, dt) -> the oracle's u after ``steps`` steps from rest


def _step_key(mode, P, mesh, steps, dts):
    return (mode, int(P), int(mesh.ncells), int(mesh.ndofs), int(steps), float(dts))


def _oracle_lib():
    from oracle import oracle_c

    try:
        oracle_c.build(native=True)
        return oracle_c.OracleLib(native=True)
    except Exception as e:  # noqa: BLE001
        log(f"native oracle build failed ({e}); using the portable build")
        return oracle_c.OracleLib()


def oracle_step_field(mode, P, mesh, solver, steps, dts, device):
    """u after ``steps`` RK4 steps from rest, by the ORACLE's time loop (oracle/rk4_oracle.py: numba-cpu/demo_linear_box.py:302-455 /
    cuda/demo_nonlinear_bowl.py:540-650 restated over oracle/fus_oracle.c) on the mesh and material the GPU solver stepped: the
    checker of the RK4 / Westervelt step lines.  One field per (mode, mesh, steps, dt) and process: the general-G and
    in-kernel-geometry lines of one mesh share it (the oracle always reads the reference's G array, formed here on the device
    by csrc/geometry.hpp -- pinned against numba-cpu/precompute.py to 1e-13 by tests/test_precompute.py)."""
    import torch

    import fusgpu_loader
    from oracle import rk4_oracle

    key = _step_key(mode, P, mesh, steps, dts)
    if key in _STEP_FIELDS:
        return _STEP_FIELDS[key]
    O = _oracle_lib()
    threads = max(1, min(O.max_threads(), host_cores()))
    pre, gll = fusgpu_loader.submodule("precompute"), fusgpu_loader.submodule("gll")
    n = P + 1
    G = getattr(solver, "G_array", None) if mode == "rk4" else getattr(solver, "G", None)
    if G is None or G.dim() != 3:  # the solver dropped the array (in-kernel geometry): form it once more for the checker
        pts, wts, _ = gll.tabulate_1d(P, np.float64)
        G = torch.empty((mesh.ncells, n**3, 6), dtype=torch.float64, device=device)
        pre.compute_scaled_geometrical_factor_device(
            G, (torch.from_numpy(mesh.x_dofs).to(device), torch.from_numpy(mesh.x_g.astype(np.float64)).to(device)), mesh.ncells,
            torch.from_numpy(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts), np.float64)).to(device),
            torch.from_numpy(gll.tensor_weights_3d(wts).astype(np.float64)).to(device))
    h = lambda t: np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float64))  # noqa: E731
    if mode == "rk4":
        geo = (h(G), h(solver.detJ), h(solver.detJ_f1), h(solver.detJ_f2))
        u, _ = rk4_oracle.solve(mesh, steps, dts, c0=solver.c0, rho0=solver.rho0, f0=solver.f0, p0=solver.p0, oracle_c=O, threads=threads, geometry=geo)
    else:
        geo = (h(G), h(solver.detJ), h(solver.dF1), h(solver.dF2))
        u, _ = rk4_oracle.solve_westervelt(mesh, steps, dts, c0=solver.c0, rho0=solver.rho0, f0=solver.f0, p0=solver.p0, oracle_c=O, threads=threads,
                                           geometry=geo)
    del G, geo
    _STEP_FIELDS[key] = u
    return u


def cpu_baseline_rk4(P, mesh, solver, dts, steps=2):
    """The oracle's RK4 loop (oracle/rk4_oracle.py: numba-cpu/demo_linear_box.py:302-455 restated; pinned by
    tests/golden/rk4_*.npz) on the SAME mesh and geometry factors the GPU stepped, ``steps`` steps serial (what the
    reference's njit loop is) and ``steps`` steps with the OpenMP stiffness apply.  "Solve time per step" -> DOF*steps/s."""
    from oracle import oracle_c, rk4_oracle

    try:
        oracle_c.build(native=True)
        O = oracle_c.OracleLib(native=True)
    except Exception as e:  # noqa: BLE001
        log(f"native oracle build failed ({e}); using the portable build")
        O = oracle_c.OracleLib()
    geo = tuple(np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float64)) for t in (solver.G, solver.detJ, solver.detJ_f1, solver.detJ_f2))
    ncores = max(1, min(O.max_threads(), host_cores()))
    res = {}
    for name, threads in (("serial", 1), ("omp", ncores)):
        tm = {}
        u, _ = rk4_oracle.solve(mesh, steps, dts, oracle_c=O, threads=threads, timing=tm, geometry=geo)
        res[name] = tm["seconds_per_step"]
    _STEP_FIELDS[_step_key("rk4", P, mesh, steps, dts)] = u  # the field of the timed loop IS the checker's (oracle_step_field)
    return {"value": mesh.ndofs / res["omp"], "unit": "DOF*steps/s", "cores": ncores, "kind": "port",
            "sample": f"full workload ({mesh.ncells} cells, {mesh.ndofs} dofs), {steps} RK4 steps per leg (time loop only, set-up excluded): OpenMP stiffness "
                      f"apply over {ncores} pinned threads; serial leg: the same loop on one thread",
            "single_thread_value": mesh.ndofs / res["serial"], "s_per_step": res["omp"], "single_thread_s_per_step": res["serial"],
            "impl": "oracle/rk4_oracle.py over oracle/fus_oracle.c (stiffness, facet mass) + numpy vector updates; the reference prints this as "
                    "'Solve time per step' (numba-cpu/demo_linear_box.py:472-473)"}


def oracle_apply(P, mesh, D, x, cc, geo, mass, portable=False, threads=None):
    """One apply of the oracle (oracle/fus_oracle.c: numba-cpu/operators.py:71-227 / :19-68 restated) on this rank's cells:
    the checker of ``result_check``, never the thing measured.  ``portable``: the prebuilt x86-64-v3 library, nothing compiled
    (N > 1: several ranks must not run the -march=native build into one file at the same time)."""
    from oracle import oracle_c

    if portable:
        O = oracle_c.OracleLib()
    else:
        try:
            oracle_c.build(native=True)
            O = oracle_c.OracleLib(native=True)
        except Exception as e:  # noqa: BLE001
            log(f"native oracle build failed ({e}); using the portable build")
            O = oracle_c.OracleLib()
    threads = max(1, min(O.max_threads(), host_cores() if threads is None else threads))
    y = np.zeros(mesh.ndofs)
    if mass:
        O.mass_apply(x, cc, y, geo, mesh.dofmap)
    else:
        O.stiffness_apply(P, D, x, cc, y, geo, mesh.dofmap, threads=threads)
    return y


def compare_with_oracle(y_gpu, y_ref, dtype, what, tol=None):
    """{rel_l2, rel_max, sum_y, ...}: the GPU result of the timed run against the oracle's on the same inputs.  Tolerance:
    SURVEY 8d (fp64 rel l2 <= 1e-12, max-abs / max <= 1e-11; fp32 1e-5 / 1e-4) unless ``tol = (rel_l2, rel_max)`` is given
    (time loops: 1e-11 / 1e-10, the bar of tests/test_solver_gpu.py)."""
    y_gpu = np.asarray(y_gpu, dtype=np.float64)
    d = y_gpu - y_ref
    nrm, mx = float(np.linalg.norm(y_ref)), float(np.max(np.abs(y_ref))) if y_ref.size else 0.0
    rel_l2 = float(np.linalg.norm(d)) / max(nrm, 1e-300)
    rel_max = (float(np.max(np.abs(d))) if d.size else 0.0) / max(mx, 1e-300)
    tol_l2, tol_max = tol if tol is not None else ((1e-12, 1e-11) if dtype == "f64" else (1e-5, 1e-4))
    return {"rel_l2": rel_l2, "rel_max": rel_max, "sum_y": float(y_gpu.sum()), "sum_y_oracle": float(y_ref.sum()), "norm_y_oracle": nrm,
            "tol_rel_l2": tol_l2, "tol_rel_max": tol_max, "ok": bool(np.isfinite(rel_l2) and rel_l2 <= tol_l2 and rel_max <= tol_max and nrm > 0.0),
            "what": what, "oracle": "oracle/fus_oracle.c (C restatement of numba-cpu/operators.py), same x, constants, geometry factors, dofmap"}
