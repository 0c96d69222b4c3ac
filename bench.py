#!/usr/bin/env python3
"""
Headline benchmark (BASELINE.json): DOF/s of the fp64 stiffness-operator apply,
P = 4 hexahedral box, on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started without a launcher (no RANK in the environment), this
script starts its own N ranks -- one child process per GPU, before the parent makes any GPU
call -- relays rank 0's JSON line and exits non-zero if any rank failed.

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

The code lives in ``benchlib/`` (modes, transports, roofline, CPU legs); this file is the entry the driver runs and hashes:
argument parsing, process / device bring-up, dispatch.
"""

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import (HBM_PEAK_GBS, emit, kernel_src_sha, lib_built_from_tree, lib_sha, log, protect_stdout,  # noqa: E402,F401
                             rehearsal, start_watchdog)
from benchlib.launch import dry_run, spawn_ranks  # noqa: E402
from benchlib.roofline import geom_bytes_per_cell, mass_bytes_per_cell, rk4_step_bytes, stiffness_bytes_per_cell  # noqa: E402,F401


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
    ap.add_argument("--mode", default="stiffness", choices=["stiffness", "stiffness_geom", "mass", "mass_diag", "rk4", "westervelt", "scatter"],
                    help="stiffness: the headline metric; mass: the cell mass apply (SURVEY 8d's second operator line); stiffness_geom: the same apply with G formed in the kernel "
                         "from the cell vertices (own bytes contract, separate line); rk4 / westervelt: one full RK4 "
                         "time step of the linear / Westervelt solver per 'step' (auxiliary metrics); scatter: scatter_forward / scatter_reverse alone "
                         "(the reference's numba-cpu/time_scatterer.py), one rank that is its own neighbour with config-4 messages")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the N-rank path (gloo): launcher, partition, halo plan and exchange; no "
                         "GPU, no operator, the printed line is marked invalid")
    ap.add_argument("--in-kernel-geometry", action="store_true",
                    help="--mode westervelt / rk4: the cell pass forms G (and detJ) from the cell vertices")
    ap.add_argument("--perturbed", action="store_true", help="--mode rk4: perturbed (non-affine) cells instead of the affine box")
    ap.add_argument("--halo", default=os.environ.get("FUS_HALO", "peer"), choices=["peer", "native", "torch"],
                    help="N > 1 transport, first choice: peer = peer-mapped arenas + send / receive kernels issued by "
                         "libfusgpu.so (default); native = grouped ncclSend/ncclRecv issued by libfusgpu.so; torch = "
                         "torch.distributed all_to_all_single.  A transport that does not come up on every rank or "
                         "fails the run's halo check is replaced by the next one (peer -> native -> torch)")
    ap.add_argument("--halo-compare", dest="halo_compare", action="store_true", default=True,
                    help="N > 1 (default since round 6: the driver's fixed command carries no extra flags): after the timed region, time the apply over "
                         "EVERY transport that comes up (peer, peer-fenced, native = RCCL) in alternating rounds in this one process and put each one's "
                         "exposed cost in config.halo_compare")
    ap.add_argument("--no-halo-compare", dest="halo_compare", action="store_false")
    ap.add_argument("--no-harvest", action="store_true",
                    help="default mode at N > 1: skip the secondary lines (partitioned mass apply, fused RK4 steps, Westervelt P = 6 step) that follow "
                         "the headline's timed region on the same partition and communicator (roofline.secondary)")
    ap.add_argument("--single-gather", action="store_true",
                    help="--mode westervelt: the single-gather cell pass (uniform c4 / c3: the vector pass writes w = u_n + kappa v_n) instead of the default two-gather pass")
    ap.add_argument("--mass-static", action="store_true",
                    help="--mode mass: mass_operator(N, T, static_detJ=True) -- detJ streamed from a row-ordered copy (opt-in: the caller promises a constant detJ)")
    ap.add_argument("--mass-atomic", action="store_true",
                    help="--mode mass: the float-atomic batch-plan kernel instead of the atomic-free transposed-dofmap kernel (a partitioned "
                         "apply keeps the atomic-free kernel too: HaloApply splits it by dof, not by cell)")
    ap.add_argument("--exclusive", action="store_true",
                    help="--mode mass: the batch plan carries exclusive-dof marks (plain load + store instead of an atomic for dofs "
                         "one batch touches alone; opt-in, measured slower from P = 4 up: profiles/r04d_ab_mass_exclusive_marks.log)")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the comparison of the timed run's y with the oracle's K x (stiffness / mass modes; done by default, the "
                         "run exits non-zero when it fails)")
    ap.add_argument("--no-aux", action="store_true", help="default mode at N = 1: skip the mass and RK4-step lines of 'aux'")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        # no launcher: become one.  Nothing above imported torch or touched the GPU.
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:], os.path.abspath(__file__)))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # CPU-baseline threads are pinned; must be in the environment before torch loads an OpenMP runtime
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    protect_stdout()
    start_watchdog(float(os.environ.get("FUS_BENCH_WATCHDOG_S", "1500")), rank)
    if args.dry_run:
        raise SystemExit(dry_run(args, rank, world))

    import torch
    import torch.distributed as dist

    import fusgpu_loader

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
        if rehearsal():
            dist.init_process_group("gloo")
        else:
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
        from benchlib.steps import bench_rk4

        return bench_rk4(args, rank, world, device)
    if args.mode == "scatter":
        if world != 1:
            raise SystemExit("--mode scatter is the N = 1 self-neighbour line (at N > 1 the exchange is inside every other mode's step)")
        from benchlib.transports import measure_scatter

        sc = measure_scatter(device, np.float64 if args.dtype == "f64" else np.float32, reps=max(1, args.steps), P=args.degree, cells=args.cells)
        first = next((v for v in sc["transports"].values() if "scatter_forward" in v), None)
        emit({"metric": "scatter_forward_reverse_us", "value": None if first is None else first["scatter_forward"]["us_per_call_stream"], "unit": "us",
              "n_gpus": 1, "steps": args.steps, "warmup": 3, "ms_per_step": None if first is None else first["scatter_forward"]["us_per_call_stream"] * 1e-3,
              "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
              "config": {"workload": sc["workload"], "lib_sha": lib_sha(), "lib_built_from_tree": lib_built_from_tree()},
              "scatter": sc, "roofline": None, "cpu_baseline": sc.get("cpu_baseline")})
        return

    from benchlib.apply import run_apply

    return run_apply(args, rank, world, device, use_dist, lib, ops, boxmesh, gll, pre)


if __name__ == "__main__":
    main()
