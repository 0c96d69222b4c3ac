"""The apply modes (``--mode stiffness`` = the headline, ``stiffness_geom``, ``mass``, ``mass_diag``): one step = one apply
y += K x (or M x) over the whole (partitioned) mesh -- forward halo of x, the kernel over all local cells, reverse halo of y.
Set-up (mesh, tables, geometry factors on the device, batch plans, at N > 1 the choice of the halo transport by the run's own
halo check), the timed region, the roofline figures, the CPU leg, the result check, and then either the auxiliary lines of the
N = 1 default run (aux_lines.py) or the N > 1 harvest (harvest.py)."""
import os
import time
import types

import numpy as np

from .aux_lines import aux_lines, failed_aux_checks
from .common import HBM_PEAK_GBS, coll_device, emit, host_cores, kernel_src_sha, lib_built_from_tree, lib_sha, log, rehearsal
from .cpu_legs import compare_with_oracle, cpu_baseline, cpu_baseline_mass, oracle_apply
from .harvest import harvest, owned_dofs_check
from .roofline import geom_bytes_per_cell, load_traffic, mass_bytes_per_cell, secondary_summary, stiffness_bytes_per_cell
from .transports import apply_variant_env, compare_transports, first_contact_report, gather_verdicts, make_comm, transport_candidates, transport_text


def run_apply(args, rank, world, device, use_dist, lib, ops, boxmesh, gll, pre):
    import torch
    import torch.distributed as dist

    import fusgpu_loader

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
    geom = args.mode == "stiffness_geom"
    mass_diag = args.mode == "mass_diag"  # cached-diagonal form of the cell mass apply: opt-in, own bytes contract, own line
    mass = args.mode == "mass" or mass_diag
    if mass_diag and use_dist:
        raise SystemExit("--mode mass_diag is a single-GPU line")
    if mass:
        # the operand in G's position is the scaled Jacobian determinant detJ[ncell][n^3] (numba-cpu/operators.py:19-68)
        del G_d
        G_d = torch.empty((mesh.ncells, n**3), dtype=x_d.dtype, device=device)
        pre.compute_scaled_jacobian_determinant_device(
            G_d, (torch.from_numpy(mesh.x_dofs).to(device), torch.from_numpy(mesh.x_g).to(device)), mesh.ncells,
            torch.from_numpy(dphi_g).to(device), torch.from_numpy(wts3).to(device))
        torch.cuda.synchronize()
    if geom and use_dist:
        raise SystemExit("--mode stiffness_geom is a single-GPU line")
    if geom:
        # own bytes contract: no G array exists for this operator
        op = ops.stiffness_operator(P, D.flatten(), dt, geometry=(mesh.x_dofs, mesh.x_g, pts, wts))
        del G_d
        G_d = None
    elif mass_diag:
        dmo = ops.diagonal_mass_operator(cc_d, G_d, dm_d, mesh.ndofs, dt)  # w = M(c) 1 assembled once, outside every step

        def op(x_, cc_, y_, detJ_, dm_):
            dmo(x_, y_)
    elif mass:
        op = ops.mass_operator(n**3, dt, exclusive=args.exclusive, atomic=args.mass_atomic or args.exclusive, static_detJ=args.mass_static)
    else:
        op = ops.stiffness_operator(P, D.flatten(), dt)

    halo, transport, halo_check, tried, first_contact, halo_compare = None, None, None, [], None, None

    def step():
        if halo is None:
            op(x_d, cc_d, y_d, G_d, dm_d)
        else:
            halo.apply(x_d, cc_d, y_d, G_d, dm_d)

    def check_halo():
        """The exchanges of THIS run, checked before anything is timed: (1) poisoned ghost entries of x come back
        from a forward scatter as their owners' values (x is an analytic field, the ghosts were filled from the
        same formula); (2) the sum of y over the OWNED dofs of all ranks equals 1^T K x = 0 (K 1 = 0, K symmetric) --
        for the mass operator: what the cells of all ranks contribute -- only if every ghost contribution reached
        its owner; (3) no device-side wait of the PEER transport timed out.  Collective: same verdict on every rank."""
        nl = mesh.nlocal
        expect = x_d[nl:].clone()
        x_d[nl:] = -777.0
        halo.fwd(x_d)
        fwd_err = float((x_d[nl:] - expect).abs().max().item()) if expect.numel() else 0.0
        x_d[nl:] = expect  # whatever the exchange did, the timed region starts from the right ghosts
        y_d.zero_()
        step()
        # mass operator: the owned sum equals what the cells of all ranks contribute, sum_c sum_i x detJ c
        ref = (x_d[dm_d.long()] * G_d * cc_d[:, None]).sum() if mass else torch.zeros((), dtype=x_d.dtype, device=device)
        timeouts = torch.tensor(float(halo.health()), dtype=x_d.dtype, device=device)
        sums = torch.stack([y_d[:nl].sum(), y_d[:nl].abs().sum(), ref, timeouts,
                            torch.tensor(fwd_err, dtype=x_d.dtype, device=device)]).to(coll_device(device))
        dist.all_reduce(sums)
        rel = abs(float(sums[0].item()) - float(sums[2].item())) / max(float(sums[1].item()), 1e-300)
        ok = float(sums[4].item()) == 0.0 and rel < (1e-9 if args.dtype == "f64" else 1e-3) and float(sums[3].item()) == 0.0
        return {"forward_max_abs_err": fwd_err, "owned_sum_defect_over_sum_abs": rel, "device_wait_timeouts": int(sums[3].item()),
                "ok": bool(ok)}

    # set-up outside every step: batch plans, communicator bring-up (and, at N > 1, the choice of the transport)
    if not use_dist:
        if hasattr(op, "prepare"):
            op.prepare(dm_d)
        for _ in range(args.warmup):
            step()
    else:
        scat = fusgpu_loader.submodule("scatterer")
        os.environ.setdefault("FUS_IPC_SPIN_SECONDS", "10")  # a transport that does not deliver fails its check in seconds
        first_contact = first_contact_report(rank, world, device)
        peer_failed_on_data = False
        for kind in transport_candidates(args):
            base, _, variant = kind.partition(":")
            if variant == "fenced" and not peer_failed_on_data:
                continue  # the fenced rung answers a DATA failure of the fence-free protocol, nothing else
            apply_variant_env(kind)
            comm, why = make_comm(base, scat, world, device)
            if comm is None:
                tried.append({"transport": kind, "result": f"did not come up: {why}"})
                log(f"halo transport {kind!r} did not come up ({why}); trying the next one")
                continue
            verdict = None
            try:
                halo = scat.HaloApply(mesh, op, comm, dt, overlap=os.environ.get("FUS_HALO_OVERLAP", "1") != "0")
                halo.prepare(x_d, cc_d, G_d, dm_d)
                for _ in range(args.warmup):
                    step()
                err = None
            except Exception as e:  # noqa: BLE001
                err = repr(e)
                log(f"rank {rank}: halo transport {kind!r} failed during bring-up: {err}")
            arena = None
            try:
                st_ = halo.fwd.status() if (err is None and base == "peer" and hasattr(halo.fwd, "status")) else {}
                arena = st_.get("arena_memory")
                if arena is not None and st_.get("fenced"):
                    arena += ", fenced"
            except Exception:  # noqa: BLE001
                pass
            bring_up = gather_verdicts(rank, world, {"error": err, "arena_memory": arena})
            failed_ranks = [v["rank"] for v in bring_up if v["error"] is not None]
            entry = {"transport": kind, "bring_up_failed_on_ranks": failed_ranks,
                     "arena_memory_by_rank": [v["arena_memory"] for v in bring_up] if base == "peer" else None}
            if not failed_ranks:
                verdict = check_halo()
                per_rank = gather_verdicts(rank, world, {"forward_max_abs_err": verdict["forward_max_abs_err"], "device_wait_timeouts": int(halo.health())})
                entry["check_failed_on_ranks"] = [v["rank"] for v in per_rank if v["forward_max_abs_err"] != 0.0 or v["device_wait_timeouts"] != 0]
                rejects = os.environ.get("FUS_BENCH_TEST_REJECT", "").split(",")  # test hook: exercise the fall-back path ("=kind": that exact rung only)
                if base in rejects or ("=" + kind) in rejects:
                    verdict = dict(verdict, ok=False, rejected_by="FUS_BENCH_TEST_REJECT")
            else:
                entry["bring_up_errors"] = {v["rank"]: v["error"] for v in bring_up if v["error"] is not None}
            if verdict is not None and verdict["ok"]:
                transport, halo_check = kind, verdict
                tried.append(dict(entry, result="ok"))
                if rank == 0:
                    log(f"halo transport {kind!r}: came up on all {world} ranks, halo check passed ({verdict}); arena memory by rank: {entry['arena_memory_by_rank']}; CHOSEN")
                break
            if kind == "peer" and verdict is not None:
                peer_failed_on_data = True  # it came up everywhere and its exchanges did not deliver: try the fenced form of the same protocol next
            tried.append(dict(entry, result=f"rejected: {verdict if verdict is not None else 'bring-up failed on rank(s) ' + str(failed_ranks)}"))
            if rank == 0:
                log(f"halo transport {kind!r} rejected: {tried[-1]}; trying the next one")
            try:
                torch.cuda.synchronize()
                dist.barrier()  # nobody frees an arena a neighbour may still write into
                if halo is not None:
                    halo.fwd.close(), halo.rev.close()
                comm.close() if hasattr(comm, "close") else None
            except Exception as e:  # noqa: BLE001
                log(f"rank {rank}: tearing down {kind!r}: {e!r}")
            halo = None
        if halo is None:
            raise SystemExit(f"no halo transport passed the halo check on rank {rank}: {tried}")
    ops.fill(0.0, y_d)  # the library's fill: streaming stores, no dirty lines left in the memory-side cache for the first timed launches to write back
    # Timed region: EXACTLY K steps issued back to back, bracketed by barrier + device synchronise on
    # both sides (wall clock -> value) and by ONE HIP-event pair on the launch stream (device time of
    # the region -> average launch duration -> roofline).  Nothing else is enqueued inside the region:
    # per-step events would serialise consecutive launches (each would have to drain before the next
    # starts) and measure the isolated launch instead; that figure is taken AFTER the region below.
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    r0.record()
    for i in range(args.steps):
        step()
    r1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    region_ms = r0.elapsed_time(r1) / args.steps
    # what the timed region left in y: K accumulated applies (y was zeroed right before it) -- kept for result_check below
    y_region = y_d.clone() if (not use_dist and not args.no_check) else None
    if halo is not None:
        # a device-side wait that gave up inside the timed region means an exchange did not deliver: no line then
        late = torch.tensor([float(halo.health())], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(late)
        if float(late.item()) != 0.0:
            raise SystemExit(f"rank {rank}: {int(late.item())} device-side halo wait(s) timed out during the timed region: the run is invalid")

    # isolated launches (outside the timed region): one event pair per step, as the reference's
    # protocol times one apply at a time (cuda/time_operators.py:272-282); agrees with the per-dispatch
    # durations of rocprofv3 --kernel-trace
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    for i in range(args.steps):
        ev0[i].record()
        step()
        ev1[i].record()
    torch.cuda.synchronize()
    ev_ms = np.array([a.elapsed_time(b) for a, b in zip(ev0, ev1)])

    def timed_launches(fn, reps=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    sched_ms = None
    if halo is not None:
        # kernel time at N > 1 (after, outside the timed region): ONE launch over all local cells -- the kernel the
        # N = 1 line times, and what the halo overhead is measured against -- and the apply's own launch schedule
        # (sub-ranges, streams, events) with no exchange in it: what cutting the launch costs by itself
        if hasattr(op, "prepare"):
            op.prepare(dm_d)
        kern_ms = timed_launches(lambda: op(x_d, cc_d, y_d, G_d, dm_d))
        sched_ms = timed_launches(lambda: halo.apply_no_exchange(x_d, cc_d, y_d, G_d, dm_d))
    else:
        kern_ms = region_ms  # N = 1: the step IS the stiffness kernel launch

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
    bpc = geom_bytes_per_cell(P, T) if geom else (mass_bytes_per_cell(P, T) if mass else stiffness_bytes_per_cell(P, T))
    alg_bytes = 3 * T * mesh.ndofs if mass_diag else mesh.ncells * bpc  # cached diagonal: w, x read, y read-modify-write
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    sha = lib_sha()
    traffic, traffic_source = (None, "not profiled for this mode") if (geom or mass) else load_traffic(P, mesh.ncells, sha, args.dtype)
    if geom:
        kname = "fus::stiffness_plan_geom_kernel"
    elif mass_diag:
        kname = "fus::muladd_kernel"
    elif mass:
        rows = halo is not None and halo.row_split(dm_d, mesh.ndofs) is not None  # partitioned: split by dof, atomic-free kernel
        kname = ops.mass_kernel_name(dm_d, mesh.ndofs, atomic=args.mass_atomic or args.exclusive or (halo is not None and not rows), static=args.mass_static)
    else:
        kname = "fus::stiffness_plan_kernel" if ops._USE_PLAN else "fus::stiffness_col_kernel"

    out = {
        "metric": "stiffness_apply_in_kernel_geometry_dof_per_s" if geom else (
            "mass_apply_cached_diagonal_dof_per_s" if mass_diag else ("mass_apply_dof_per_s" if mass else "stiffness_apply_dof_per_s")),
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
            "workload": ("cell mass apply y+=Mx" if mass else "stiffness apply y+=Kx") + f", P={P} GLL hex box, {gcells[0]}x{gcells[1]}x{gcells[2]} perturbed cells, "
            f"{ndofs_global} dofs" + (" (BASELINE config 3)" if (world == 1 and P == 4 and args.cells == 54) else ""),
            "degree": P,
            "cells_per_gpu": mesh.ncells,
            "global_dofs": ndofs_global,
            "partition": f"{grid[0]}x{grid[1]}x{grid[2]} blocks",
            "geometry": ("formed in the kernel from the 8 vertices of each trilinear cell (no G array; NOT the headline "
                         "bytes contract)") if geom else ("scaled Jacobian determinant detJ[ncell][n^3]" if mass else
                                                           "general per-quadrature-point G[ncell][n^3][6] (no affine shortcut)"),
            "stiffness_kernel": None if mass else ("planned (batch plan, LDS pre-reduction)" if ops._USE_PLAN else f"plan-free variant {lib.get_tuning(lib.TUNE_STIFFNESS_VARIANT)}"),
            "xcd_remap": lib.get_tuning(lib.TUNE_XCD_REMAP),
            "halo": None if halo is None else ("overlapped" if halo.overlap else "sequential"),
            "halo_schedule": None if halo is None else halo.schedule_kind,
            "halo_lead_cells": None if halo is None else halo.lead_cells,
            "halo_check": halo_check,
            "halo_transport": None if halo is None else transport_text(transport),
            "halo_transports_tried": tried or None,
            "first_contact": first_contact,
            "halo_compare": halo_compare,
            # the step against ONE launch over all local cells (the kernel of the N = 1 line) ...
            "halo_exposed_ms": None if halo is None else ms_per_step - kern_ms,
            "halo_exposed_frac": None if halo is None else (ms_per_step - kern_ms) / kern_ms,
            # ... of which: cutting that launch into the schedule's sub-launches (no exchange), and the exchanges
            "halo_split_cost_ms": None if halo is None else sched_ms - kern_ms,
            "halo_exchange_exposed_ms": None if halo is None else ms_per_step - sched_ms,
            "schedule_launches_ms": sched_ms,
            "ranks": world,
            "lib_sha": sha,
            "lib_source_hash": lib.load().fus_source_hash().decode(),
            "kernel_src_sha": kernel_src_sha(),
            "lib_built_from_tree": lib_built_from_tree(),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": kname,
            "kernel_ms": kern_ms,
            "kernel_ms_how": ("one HIP-event pair around the K back-to-back launches of the timed region / K" if halo is None
                              else "event pair around 10 back-to-back launches over ALL local cells (one launch each, no exchange), after the timed region"),
            "isolated_launch_ms_mean": float(ev_ms.mean()),  # one event pair per launch, outside the timed region
            "isolated_launch_ms_min": float(ev_ms.min()),
            "isolated_launch_ms_std": float(ev_ms.std()),
            "isolated_frac": None if halo is not None else alg_bytes / (float(ev_ms.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_cell": None if mass_diag else bpc,
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_contract": ("cached diagonal: y += (M(c) 1) (.) x, 3 vector touches per dof; NOT the reference's gather-scale-scatter contract "
                               "(that is --mode mass)") if mass_diag else None,
            "cells_per_launch": mesh.ncells,
            "pct_of_hbm_roofline_dofs": 100.0 * achieved / HBM_PEAK_GBS,
            "measured_copy_gbs": copy_gbs,   # 1 GiB -> 1 GiB with fus_copy (read + write bytes)
            "measured_read_gbs": read_gbs,   # 1 GiB read-only reduction (torch.sum)
        },
    }
    if rehearsal():
        out.update(valid=False, rehearsal="ranks share the visible GPU(s) (bootstrap over gloo): NOT a measurement")
    out["cpu_baseline"] = None
    pb = None
    if rank == 0 and world == 1 and not use_dist:
        if not args.no_cpu_baseline and mass:
            try:
                out["cpu_baseline"] = cpu_baseline_mass(P, mesh, x.astype(np.float64), cc.astype(np.float64), G_d.cpu().numpy().astype(np.float64))
            except Exception as e:  # noqa: BLE001
                log(f"cpu_baseline failed: {e!r}")
        elif not args.no_cpu_baseline and not geom:
            G = G_d.cpu().numpy()  # the CPU baseline streams the same G the GPU did
            pb = dict(mesh=mesh, D=D.astype(np.float64), x=x.astype(np.float64), cc=cc.astype(np.float64), G=np.asarray(G, dtype=np.float64))
            try:
                out["cpu_baseline"] = cpu_baseline(P, pb)
            except Exception as e:  # noqa: BLE001
                log(f"cpu_baseline failed: {e!r}")
    y_oracle_keep = None
    # ---- result check bound to the timed run (the reference keeps cuda/test_operators.py:213-312 next to cuda/time_operators.py:204-290
    # on the same operators): what the timed region left in y, against the oracle's apply on the same inputs
    check = None
    if not args.no_check:
        try:
            x64, cc64, D64 = x.astype(np.float64), cc.astype(np.float64), D.astype(np.float64)
            if mass:
                geo_h = G_d.cpu().numpy().astype(np.float64)
            elif pb is not None:
                geo_h = pb["G"]
            else:
                Gt = G_d
                if Gt is None:  # in-kernel geometry: the oracle still takes the reference's G array (numba-cpu/precompute.py:115-163)
                    Gt = torch.empty((mesh.ncells, n**3, 6), dtype=x_d.dtype, device=device)
                    pre.compute_scaled_geometrical_factor_device(
                        Gt, (torch.from_numpy(mesh.x_dofs).to(device), torch.from_numpy(mesh.x_g).to(device)), mesh.ncells,
                        torch.from_numpy(dphi_g).to(device), torch.from_numpy(wts3).to(device))
                geo_h = Gt.cpu().numpy().astype(np.float64)
                del Gt
            if os.environ.get("FUS_BENCH_TEST_BREAK_CHECK") == "1":  # test hook: the checker sees other constants than the GPU did
                cc64 = cc64 * (1.0 + 1e-3)
                if pb is not None:
                    pb.pop("y_oracle", None)
            if not use_dist:
                y_ref = pb.get("y_oracle") if pb is not None else None
                if y_ref is None:
                    y_ref = oracle_apply(P, mesh, D64, x64, cc64, geo_h, mass)
                y_oracle_keep = None if mass else y_ref  # K x of the headline's inputs: aux_lines checks the in-kernel-geometry apply with it
                check = compare_with_oracle(y_region.cpu().numpy() / args.steps, y_ref, args.dtype,
                                            f"y of the timed region (zeroed before it, {args.steps} accumulated applies) / {args.steps}  vs  one oracle apply, all {mesh.ndofs} dofs")
                y_region = None
            else:
                # N > 1: the ghost block of y keeps its partial sums from step to step, so the region's y is not K steps of one
                # operator; one more apply into a zeroed y (same halo objects, same kernels), owned dofs of every rank against the
                # oracle's apply over this rank's cells reverse-scattered through the transport the halo check passed
                if rank == 0:  # the portable oracle library travels prebuilt; should it be missing, ONE rank builds it
                    from oracle import oracle_c

                    oracle_c.OracleLib()
                dist.barrier()
                y_loc = oracle_apply(P, mesh, D64, x64, cc64, geo_h, mass, portable=True, threads=max(1, host_cores() // max(1, world)))
                y_ref_d = torch.from_numpy(y_loc.astype(dt)).to(device)
                halo.rev(y_ref_d)
                ops.fill(0.0, y_d)
                step()
                torch.cuda.synchronize()
                check = owned_dofs_check(y_d, y_ref_d, mesh.nlocal, halo.health(), device, args.dtype,
                                         "one apply after the timed region (same halo objects and kernels) into a zeroed y, owned dofs of all ranks  vs  the "
                                         "oracle's apply over each rank's cells, reverse-scattered")
        except Exception as e:  # noqa: BLE001
            log(f"result check could not run: {e!r}")
            check = {"ok": False, "error": repr(e), "rel_l2": float("nan"), "rel_max": float("nan"), "sum_y": float("nan")}
        out["check"] = check
        out["config"]["check"] = check  # the driver's record keeps ``config`` verbatim
        if not check["ok"]:
            out["valid"] = False
    if pb is not None:
        pb.pop("G", None)  # 945 MB of host memory the auxiliary lines do not need
    bad_aux = []
    if world == 1 and not use_dist and not (geom or mass) and not args.no_aux and not args.no_plan:
        # SURVEY 8d asks for the mass apply next to the stiffness apply, north_star names the RK4 step: all after the timed region
        # and the check of the headline, same mesh, each with its own bytes contract and its own check (reference:
        # numba-cpu/time_operators.py:176-268 times the operators in one script)
        c = types.SimpleNamespace(P=P, T=T, dt=dt, mesh=mesh, x_d=x_d, cc_d=cc_d, y_d=y_d, dm_d=dm_d, G_d=G_d, dphi_g=dphi_g, wts3=wts3, pts=pts, wts=wts,
                                  D=D, x=x, cc=cc, op=op, step=step, alg_bytes=alg_bytes)
        out["aux"] = aux_lines(args, rank, world, device, c, ops, pre, y_oracle_keep)
        bad_aux = failed_aux_checks(out["aux"])
        if bad_aux:
            out["valid"] = False
    # ---- N > 1 extras: everything from here on is OPTIONAL next to the headline, which is measured and checked by now.  A phase that hangs
    # (a transport's bring-up on a platform it has never seen, a collective whose peer is gone) must not take the line with it: a timer emits what
    # the run has -- the headline with its check, and what the extras finished -- and leaves (every rank has its own; rank 0 prints)
    extras_timer, extras_phase = None, ["halo_compare"]
    if use_dist and halo is not None:
        import threading

        budget = float(os.environ.get("FUS_BENCH_EXTRAS_TIMEOUT_S", "420"))

        def bail():
            # (runs on a timer thread: the main thread is normally blocked by now, but it may only be SLOW and still filling ``out`` -- serialise a
            # snapshot, retry if a dict changes underneath, and leave whatever happens: a bail-out that raises would be a hang again)
            code = 0 if (check is None or check["ok"]) else 3
            try:
                log(f"rank {rank}: the optional phases did not finish in {budget:.0f} s (in: {extras_phase[0]}): the line is emitted without the rest")
                if rank == 0:
                    import copy

                    snap = None
                    for _ in range(5):
                        try:
                            snap = copy.deepcopy(out)
                            break
                        except RuntimeError:
                            time.sleep(0.05)
                    if snap is None:  # the headline alone
                        snap = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                                        "vs_baseline", "dtype", "data", "valid", "check")}
                        snap.update(config={"workload": out["config"]["workload"]}, roofline={k: v for k, v in out["roofline"].items() if k != "secondary"},
                                    cpu_baseline=None)
                    snap["extras_timed_out"] = {"seconds": budget, "phase": extras_phase[0]}
                    try:
                        snap["roofline"]["secondary"] = secondary_summary(snap)
                    except Exception:  # noqa: BLE001
                        pass
                    emit(snap)
            finally:
                os._exit(code)

        if budget > 0:
            extras_timer = threading.Timer(budget, bail)
            extras_timer.daemon = True
            extras_timer.start()
        if os.environ.get("FUS_BENCH_TEST_HANG_EXTRAS") == str(rank):  # test hook: this rank never reaches the optional phases
            time.sleep(3600)
    if use_dist and halo is not None and args.halo_compare:
        try:
            halo_compare = compare_transports(args, rank, world, device, scat, mesh, op, dt, x_d, cc_d, y_d, G_d, dm_d, transport, halo, kern_ms)
        except Exception as e:  # noqa: BLE001
            log(f"rank {rank}: halo compare failed: {e!r}")
            halo_compare = {"error": repr(e)}
        out["config"]["halo_compare"] = halo_compare
    extras_phase[0] = "harvest"
    harvested = None
    if use_dist and halo is not None and not (geom or mass) and not args.no_harvest and not args.no_plan:
        # N > 1, the driver's fixed command (no extra flags): the other lines of the path on the SAME partition and communicator, after
        # the headline's timed region -- partitioned mass apply, fused RK4 steps, Westervelt P = 6 step (BASELINE config 5)
        harvested = out["harvest"] = {}
        try:
            harvest(args, rank, world, device, halo.comm, mesh, dt, x_d, cc_d, y_d, dm_d, dphi_g, wts3, x, cc, ops, pre,
                    budget_s=float(os.environ.get("FUS_BENCH_HARVEST_BUDGET_S", "150")), into=harvested)
        except Exception as e:  # noqa: BLE001
            log(f"rank {rank}: harvest failed: {e!r}")
            harvested["error"] = repr(e)
    if out.get("aux") is not None or check is not None or harvested is not None:
        try:
            out["roofline"]["secondary"] = secondary_summary(out)
        except Exception as e:  # noqa: BLE001
            log(f"roofline.secondary failed: {e!r}")
    if extras_timer is not None:
        extras_timer.cancel()
    if rank == 0:
        emit(out)
    failed = (check is not None and not check["ok"]) or bool(bad_aux)
    if failed and rank == 0:
        log(f"RESULT CHECK FAILED: headline {check}; auxiliary lines with a failed check: {bad_aux}")
    if use_dist:
        try:
            torch.cuda.synchronize()
            dist.barrier()
        except Exception:  # noqa: BLE001
            pass
        dist.destroy_process_group()
    if failed:
        raise SystemExit(3)

