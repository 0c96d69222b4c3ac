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


def mass_bytes_per_cell(P, T):
    """Algorithmic HBM bytes per cell of the cell mass apply: detJ + dofmap + x once + y RMW + constant."""
    nd = (P + 1) ** 3
    return nd * T + 4 * nd + 3 * T * P**3 + T


def geom_bytes_per_cell(P, T):
    """Algorithmic HBM bytes per cell of the in-kernel-geometry apply: dofmap + x once + y RMW +
    constant + the cell's vertex ids (8 int32) + vertex coordinates, each vertex read once
    (3 T per cell asymptotically).  DESIGN.md 3.3."""
    nd = (P + 1) ** 3
    return 4 * nd + 3 * T * P**3 + T + 32 + 3 * T


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


_JSON_FD = None


def protect_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries loaded below write there too (RCCL
    prints a version banner on stdout when a communicator is created with ncclCommInitRank), so file
    descriptor 1 is pointed at stderr for the life of the process and the JSON line is written to a
    private duplicate of the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def start_watchdog(seconds, rank):
    """A rank that is still running after ``seconds`` is taken to be hung (a collective whose peer never
    arrived, a kernel that never drains): say so and leave with a non-zero code, so that the launcher
    stops the other ranks and the caller sees a failure instead of a job that never ends."""
    import threading

    if seconds <= 0:
        return

    def fire():
        log(f"rank {rank}: watchdog: still running after {seconds:.0f} s -- giving up (FUS_BENCH_WATCHDOG_S=0 disables)")
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()


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


def lib_sha():
    """Short hash of the libfusgpu.so this run loads (ties a bench line to the profiled binary)."""
    import hashlib

    import fusgpu_loader

    path = fusgpu_loader.submodule("_lib").LIB_PATH
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:12]
    except OSError:
        return None


def kernel_src_sha(files=("plan.hpp", "stiffness.hpp", "stiffness_plan.hpp")):
    """Hash of what defines a kernel's code (its sources and the compile flags; default: the HEADLINE kernel): a PMC pass
    stays valid for a library that differs from the profiled one only elsewhere (halo transport, ABI glue)."""
    import hashlib

    csrc = os.path.join(ROOT, "fenicsx-fus-gpu_amd", "csrc")
    h = hashlib.sha256()
    try:
        for n in ("Makefile",) + tuple(files):
            with open(os.path.join(csrc, n), "rb") as f:
                data = f.read()
            if n == "Makefile":  # only the compile flags: the header list changes with every new file
                data = b"\n".join(line for line in data.split(b"\n") if line.startswith((b"CXXFLAGS", b"ARCH", b"           -f")))
            h.update(data)
    except OSError:
        return None
    return h.hexdigest()[:12]


def lib_built_from_tree():
    """True if the loaded libfusgpu.so was built from exactly the sources in this tree (fus_source_hash())."""
    import fusgpu_loader

    try:
        return bool(fusgpu_loader.submodule("_lib").built_from_tree())
    except Exception:
        return None


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """``bench.py --gpus N`` started without a launcher: run N copies of this script, one rank per
    GPU, rendezvous on 127.0.0.1.  The parent never touches the GPU (no HIP call, no torch import);
    rank 0's stdout is relayed, every rank's stderr is inherited.  Returns the exit code."""
    import subprocess

    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    out0 = None
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            try:
                if r == 0 and out0 is None:
                    out0, _ = procs[0].communicate(timeout=0.5)
                else:
                    procs[r].wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if procs[r].returncode != 0 and rc == 0:
                rc = procs[r].returncode or 1
                log(f"rank {r} exited with code {procs[r].returncode}: stopping the other ranks")
                for q in pending:  # they would hang in the next collective
                    procs[q].terminate()
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    return rc


class _DryRunKernels:
    """pack / unpack with plain torch indexing -- ONLY for ``--dry-run`` (launcher / rendezvous /
    halo-plan rehearsal on CPU under gloo; nothing is measured and no operator is applied)."""

    def index_tensor(self, idx_np):
        import torch

        return torch.from_numpy(np.ascontiguousarray(idx_np, dtype=np.int64))

    def buffer(self, n):
        import torch

        return torch.empty(int(n), dtype=torch.float64)

    def pack_fwd(self, in_, out, index):
        out.copy_(in_[index])

    def unpack_fwd(self, in_, out, index, N):
        out[index + N] = in_

    def pack_rev(self, in_, out, index, N):
        out.copy_(in_[index + N])

    def unpack_rev(self, in_, out, index):
        out.index_add_(0, index, in_)


def dry_run(args, rank, world):
    """Rehearsal of the N-rank path without a GPU: spawn / rendezvous (gloo), partition, halo plan
    exchange, forward + reverse all-to-all-v with the real per-neighbour counts, barrier + max-over-ranks
    timing, one JSON line.  The line is marked invalid: nothing here is a measurement."""
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    if os.environ.get("FUS_BENCH_TEST_FAIL_RANK") == str(rank):  # launcher test: one rank dies after rendezvous
        os._exit(3)
    if os.environ.get("FUS_BENCH_TEST_HANG_RANK") == str(rank):  # launcher test: one rank never reaches the collectives
        time.sleep(3600)
    boxmesh, scat, utils = (fusgpu_loader.submodule(m) for m in ("boxmesh", "scatterer", "utils"))
    P = args.degree
    grid = boxmesh.default_grid(world)
    cells = min(args.cells, 4)
    mesh = boxmesh.BoxMesh(P, tuple(cells * g for g in grid), grid=grid, rank=rank)
    comm = scat.TorchComm()
    od, gd = utils.compute_scatterer_data_flat(mesh.index_map, comm if world > 1 else None)
    k = _DryRunKernels()
    fwd = scat.scatter_forward(comm, od, gd, mesh.nlocal, np.float64, kernels=k)
    rev = scat.scatter_reverse(comm, od, gd, mesh.nlocal, np.float64, kernels=k)
    lex = torch.from_numpy(mesh.global_lexicographic_ids().astype(np.float64))
    x = lex.clone()
    x[mesh.nlocal:] = -1.0
    for _ in range(args.warmup):
        fwd(x)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd(x)
        rev(torch.zeros_like(x))
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = torch.tensor([float(torch.equal(x, lex))])  # every ghost now holds its owner's value
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        emit({
            "metric": "stiffness_apply_dof_per_s", "value": None, "unit": "DOF/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": float(el.item()) / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "dry_run": True,
            "valid": False, "halo_ok": bool(ok.item() == 1.0), "ranks": dist.get_world_size(), "backend": "gloo",
            "config": {"workload": f"DRY RUN (CPU, gloo): halo exchange only, P={P}, {cells}^3 cells per rank",
                       "partition": f"{grid[0]}x{grid[1]}x{grid[2]} blocks", "global_dofs": mesh.ndofs_global},
            "roofline": None, "cpu_baseline": None})
    dist.destroy_process_group()
    return 0 if ok.item() == 1.0 else 1


def coll_device(device):
    """Where the tensors of bench.py's own collectives (barrier flags, max-over-ranks time) live."""
    import torch

    return torch.device("cpu") if rehearsal() else device


def rehearsal():
    """FUS_BENCH_REHEARSAL=1: the N-rank code path of THIS script with real HIP kernels and real processes
    where only one GPU exists -- every rank on the visible GPU(s) modulo their count, torch.distributed over
    gloo, the exchange staged through the host.  The line is marked invalid: not a measurement."""
    return os.environ.get("FUS_BENCH_REHEARSAL", "0") == "1"


class _StagedGlooComm:
    """Exchange of device tensors over gloo, staged through the host -- rehearsal only."""

    def __init__(self, inner):
        self.inner = inner
        self.rank, self.size, self.backend = inner.rank, inner.size, inner.backend

    def alltoallv(self, send, send_counts, recv, recv_counts, async_op=False):
        import torch

        torch.cuda.synchronize()
        s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        self.inner.alltoallv(s, send_counts, r, recv_counts)
        recv.copy_(r)
        return None

    def alltoallv_int64(self, *a):
        return self.inner.alltoallv_int64(*a)

    def barrier(self):
        self.inner.barrier()


TRANSPORT_TEXT = {
    "peer": "libfusgpu.so PEER transport: peer-mapped arenas (HIP IPC), send / receive kernels with sequence flags, no RCCL kernel",
    "native": "libfusgpu.so: grouped ncclSend/ncclRecv on a library-owned stream",
    "torch": "torch.distributed.all_to_all_single (RCCL)",
}


def transport_candidates(args):
    """Transports this run may use, in order of preference; the first one that comes up on EVERY rank and passes the
    run's own halo check is used (decided collectively, recorded in the line)."""
    # "peer:finegrained": the PEER transport once more with its receive arenas in fine-grained instead of uncached device memory
    # (FUS_IPC_MEMORY) -- export / open of UNCACHED memory between two different devices has never run on this pool (one GPU per box),
    # and falling straight back to RCCL would cost 22 % per apply where another memory kind might cost nothing
    return {"peer": ["peer", "peer:finegrained", "native", "torch"], "native": ["native", "torch"], "torch": ["torch"]}[args.halo]


def make_comm(kind, scat, world, device):
    """One candidate transport, created on all ranks or on none: returns (comm, None) or (None, reason)."""
    import torch
    import torch.distributed as dist

    if kind == "torch":
        return (_StagedGlooComm(scat.TorchComm()) if rehearsal() else scat.TorchComm()), None
    if kind == "native" and rehearsal():
        return None, "RCCL refuses two ranks on one device (rehearsal)"
    comm, err = None, None
    try:
        comm = scat.NativeComm(transport="peer" if kind == "peer" else "rccl")
    except Exception as e:  # noqa: BLE001  (NativeComm itself fails on all ranks or on none; this is the belt to its braces)
        err = repr(e)
        log(f"{kind} communicator failed on rank {dist.get_rank() if dist.is_initialized() else 0}: {err}")
    ok = 1.0 if err is None else 0.0
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = float(flag.item())
    if ok == 1.0:
        return comm, None
    if comm is not None:
        comm.close()
    return None, err or "failed on another rank"


AUX_STEADY_LAUNCHES, AUX_STEADY_WARM = 200, 100


def timed_steady(fn, burst):
    """(ms per launch in the steady state, ms per launch of the first burst): ``burst`` back-to-back launches after 3 untimed ones
    (what the aux lines timed until round 4), then AUX_STEADY_WARM untimed + AUX_STEADY_LAUNCHES timed launches between one HIP-event pair.
    A burst of 20 launches of a 0.15 ms kernel is over in 3 ms -- inside the ramp of the device's clocks after the idle gap before
    it: the in-kernel-geometry kernel (VALU / LDS heavy) reads 0.16-0.185 ms in such a burst and 0.155 ms from the 50th launch on,
    whatever ran before (profiles/r05d_geom_variance_probe.log); a time loop runs in the steady state."""
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
    return e0.elapsed_time(e1) / AUX_STEADY_LAUNCHES, first


def aux_mass(args, P, T, dt, mesh, x_d, cc_d, y_d, dm_d, dphi_g, wts3, device, ops, pre, x_host, cc_host):
    """The cell mass apply y += M(c) x on the headline's mesh (numba-cpu/operators.py:19-68; shares the stiffness
    operator's batch plan): K back-to-back launches between one HIP-event pair, 3 044 B/cell at P = 4 / fp64."""
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

    def timed(fn):
        for _ in range(AUX_STEADY_WARM):
            fn(x_d, cc_d, y_d, detJ, dm_d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K):
            fn(x_d, cc_d, y_d, detJ, dm_d)
        e1.record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3, e0.elapsed_time(e1) / K

    # the float-atomic batch-plan kernel beside it (what the sub-launches of a partitioned apply use)
    _, atomic_ms = timed(mop.atomic)
    # opt-in: detJ declared constant across applies -> streamed from a row-ordered copy instead of gathered through the entry ids
    static_ms = None
    try:
        _, static_ms = timed(ops.mass_operator(n**3, dt, static_detJ=True))
    except Exception as e:  # noqa: BLE001
        log(f"aux mass, static-detJ form failed: {e!r}")
    wall_ms, ms = timed(mop)
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
           "cpu_baseline": None}
    if x_host is not None:
        out["cpu_baseline"] = cpu_baseline_mass(P, mesh, x_host.astype(np.float64), cc_host.astype(np.float64), detJ.cpu().numpy().astype(np.float64))
    # the same operator in cached-diagonal form (opt-in, own contract: 3 vector touches per dof)
    dmo = ops.diagonal_mass_operator(cc_d, detJ, dm_d, mesh.ndofs, dt)
    for _ in range(3):
        dmo(x_d, y_d)
    e0.record()
    for _ in range(K):
        dmo(x_d, y_d)
    e1.record()
    torch.cuda.synchronize()
    msd = e0.elapsed_time(e1) / K
    ach = 3 * T * mesh.ndofs / (msd * 1e-3) / 1e9
    diag = {"metric": "mass_apply_cached_diagonal_dof_per_s", "value": mesh.ndofs_global / (msd * 1e-3), "unit": "DOF/s", "ms_per_step": msd, "steps": K,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "fus::muladd_kernel", "kernel_ms": msd, "algorithmic_bytes_per_launch": 3 * T * mesh.ndofs,
                         "bytes_contract": "y += (M(c) 1) (.) x with w = M(c) 1 assembled once: 3 vector touches per dof (opt-in; not the reference's gather-scale-scatter)"},
            "cpu_baseline": None}
    return out, diag


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
        rk4_oracle.solve(mesh, steps, dts, oracle_c=O, threads=threads, timing=tm, geometry=geo)
        res[name] = tm["seconds_per_step"]
    return {"value": mesh.ndofs / res["omp"], "unit": "DOF*steps/s", "cores": ncores, "kind": "port",
            "sample": f"full workload ({mesh.ncells} cells, {mesh.ndofs} dofs), {steps} RK4 steps per leg (time loop only, set-up excluded): OpenMP stiffness "
                      f"apply over {ncores} pinned threads; serial leg: the same loop on one thread",
            "single_thread_value": mesh.ndofs / res["serial"], "s_per_step": res["omp"], "single_thread_s_per_step": res["serial"],
            "impl": "oracle/rk4_oracle.py over oracle/fus_oracle.c (stiffness, facet mass) + numpy vector updates; the reference prints this as "
                    "'Solve time per step' (numba-cpu/demo_linear_box.py:472-473)"}


def first_contact_report(rank, world, device):
    """N > 1, before anything is exchanged: what the transports will find, one block on rank 0's stderr and the same facts in the
    line (``config.first_contact``) -- the devices by PCI bus id (ordinals are process-local), which of the devices visible to a
    rank it can reach peer-to-peer, the IPC mode of the environment.  The first run on a real 8-GPU node must explain itself."""
    import torch
    import torch.distributed as dist

    def pci(d):
        p = torch.cuda.get_device_properties(d)
        try:
            return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except AttributeError:
            return f"ordinal-{d}"

    me = {"rank": rank, "pid": os.getpid(), "device_ordinal": device.index, "pci_bus_id": pci(device.index),
          "name": torch.cuda.get_device_properties(device.index).name, "visible_devices": torch.cuda.device_count(), "peer_access": {}}
    for d in range(torch.cuda.device_count()):
        if d != device.index:
            try:
                me["peer_access"][pci(d)] = bool(torch.cuda.can_device_access_peer(device.index, d))
            except Exception as e:  # noqa: BLE001
                me["peer_access"][pci(d)] = f"error: {e!r}"
    every = [None] * world
    dist.all_gather_object(every, me)
    report = {"ranks": every, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "FUS_IPC_MEMORY": os.environ.get("FUS_IPC_MEMORY"),
              "rehearsal": rehearsal()}
    if rank == 0:
        log("first contact: " + ", ".join(f"rank {r['rank']} -> {r['pci_bus_id']} ({r['name']}, ordinal {r['device_ordinal']} of {r['visible_devices']})" for r in every))
        shared = len({r["pci_bus_id"] for r in every}) < world
        if shared:
            log("first contact: several ranks share one device (rehearsal): peer access is not the question here")
        for r in every:
            no = [k for k, v in r["peer_access"].items() if v is not True]
            log(f"first contact: rank {r['rank']} peer access to the other visible devices: " + ("all" if not no else f"NOT to {no}") + f" ({len(r['peer_access'])} checked)")
        log(f"first contact: HSA_ENABLE_IPC_MODE_LEGACY={report['HSA_ENABLE_IPC_MODE_LEGACY']!r} (must be '0': dmabuf IPC), FUS_IPC_MEMORY={report['FUS_IPC_MEMORY']!r}")
    return report


def compare_transports(args, rank, world, device, scat, mesh, op, dt, x_d, cc_d, y_d, G_d, dm_d, chosen_kind, chosen_halo, kern_ms, rounds=5):
    """``--halo-compare``: the apply over every transport that comes up (the chosen one + the other of peer / native), timed in
    ALTERNATING rounds of K steps in this one process (barrier + synchronise on both sides, max over ranks), each one's exposed cost
    against ONE launch over all local cells; the result of one apply through each extra transport is compared with the chosen
    transport's.  One ``bench.py --gpus 8 --halo-compare`` run answers "PEER or RCCL, and by how much" (VERDICT r4 item 7)."""
    import torch
    import torch.distributed as dist

    ops_mod = __import__("fusgpu_loader").submodule("operators")
    halos, comms, notes = {chosen_kind: chosen_halo}, {}, {}
    for kind in ("peer", "native"):
        if kind in halos:
            continue
        comm, why = make_comm(kind, scat, world, device)
        if comm is None:
            notes[kind] = f"did not come up: {why}"
            continue
        err, h = None, None
        try:
            h = scat.HaloApply(mesh, op, comm, dt, overlap=os.environ.get("FUS_HALO_OVERLAP", "1") != "0")
            h.prepare(x_d, cc_d, G_d, dm_d)
            h.apply(x_d, cc_d, y_d, G_d, dm_d)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        every = gather_verdicts(rank, world, {"error": err})
        if any(v["error"] for v in every):
            notes[kind] = f"bring-up failed on rank(s) {[v['rank'] for v in every if v['error']]}: {[v['error'] for v in every if v['error']][:2]}"
            try:
                dist.barrier()
                if h is not None:
                    h.fwd.close(), h.rev.close()
                comm.close()
            except Exception:  # noqa: BLE001
                pass
            continue
        halos[kind], comms[kind] = h, comm
    # one apply through each transport into a zeroed y: the extra transports against the chosen one
    ref, diffs = None, {}
    for kind, h in halos.items():
        ops_mod.fill(0.0, y_d)
        h.apply(x_d, cc_d, y_d, G_d, dm_d)
        torch.cuda.synchronize()
        owned = y_d[: mesh.nlocal].clone()
        if ref is None:
            ref = owned
        else:
            t = torch.stack([(owned - ref).abs().max() if owned.numel() else owned.new_zeros(()), ref.abs().max() if ref.numel() else ref.new_zeros(())]).double().to(coll_device(device))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            diffs[kind] = float(t[0].item()) / max(float(t[1].item()), 1e-300)
    times = {k: [] for k in halos}
    for _ in range(rounds):
        for kind, h in halos.items():
            h.apply(x_d, cc_d, y_d, G_d, dm_d)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                h.apply(x_d, cc_d, y_d, G_d, dm_d)
            torch.cuda.synchronize()
            dist.barrier()
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_device(device))
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            times[kind].append(float(el.item()) / args.steps * 1e3)
    out = {"rounds": rounds, "steps_per_round": args.steps, "one_launch_ms": kern_ms, "chosen": chosen_kind, "transports": {}, "not_compared": notes or None}
    for kind, h in halos.items():
        med = float(np.median(times[kind]))
        late = torch.tensor([float(h.health())], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(late)
        out["transports"][kind] = {"transport": TRANSPORT_TEXT[kind], "schedule": h.schedule_kind, "ms_per_step_median": med,
                                   "ms_per_step_rounds": times[kind], "exposed_ms": med - kern_ms, "exposed_frac": (med - kern_ms) / kern_ms,
                                   "failed_waits_all_ranks": int(late.item()), "max_rel_diff_vs_chosen": diffs.get(kind)}
    if rank == 0:
        log("halo compare: " + "; ".join(f"{k}: {v['ms_per_step_median']:.4f} ms/step = one launch {v['exposed_ms'] * 1e3:+.1f} us ({100 * v['exposed_frac']:+.1f} %)"
                                         for k, v in out["transports"].items()) + (f"; not compared: {notes}" if notes else ""))
    try:
        torch.cuda.synchronize()
        dist.barrier()
        for kind, comm in comms.items():
            halos[kind].fwd.close(), halos[kind].rev.close()
            comm.close()
    except Exception as e:  # noqa: BLE001
        log(f"rank {rank}: halo compare teardown: {e!r}")
    return out


def gather_verdicts(rank, world, mine):
    """Every rank's view of one bring-up / check step, so that the line and the log name the rank that failed."""
    import torch.distributed as dist

    every = [None] * world
    dist.all_gather_object(every, dict(mine, rank=rank))
    return every


def aux_traffic(key, P, ncell, dtype):
    """(HBM bytes per launch / per step of the ``aux.<key>`` entry of profiles/traffic_latest.json, source) or (None, reason):
    a REPLAYED figure of separate rocprofv3 --pmc passes, reported only when the workload is the profiled one and every
    kernel source it names (and the compile flags) are the profiled ones -- the rule of the headline's ``roofline.traffic``."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            t = json.load(f).get("aux", {}).get(key)
    except Exception:
        return None, "no profiles/traffic_latest.json"
    if not t:
        return None, f"no PMC passes in profiles/traffic_latest.json (aux.{key})"
    if int(t.get("P", -1)) != P or int(t.get("ncell", -1)) != ncell or t.get("dtype", "f64") != dtype:
        return None, "profiled workload differs from this run"
    files = tuple(t.get("kernel_src_files", ()))
    if not files or t.get("kernel_src_sha") != kernel_src_sha(files) or not lib_built_from_tree():
        return None, "the kernel sources differ from the profiled ones"
    val = t.get("hbm_bytes_per_step", t.get("hbm_bytes_per_launch"))
    src = f"(2 FETCH_SIZE + WRITE_SIZE) x 1024 of separate rocprofv3 --pmc passes ({t.get('source')}"
    if t.get("breakdown"):
        src += f"; per launch: {t['breakdown']}, 4 launches of each per step"
    return float(val), src + "); same kernel sources and compile flags"


def rk4_step_traffic(P, ncell, dtype, in_kernel_geometry):
    """(HBM bytes per fused RK4 step from the committed per-kernel PMC passes, source) or (None, reason): the sum over the
    step's launches of each kernel's per-launch bytes, replayed only when every kernel's sources and the compile flags are
    the profiled ones (as the headline's traffic)."""
    key = "rk4_step_in_kernel_geometry" if in_kernel_geometry else "rk4_step"
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            t = json.load(f).get("aux", {}).get(key)
    except Exception:
        return None, "no profiles/traffic_latest.json"
    if not t:
        return None, f"no PMC passes of the step's kernels in profiles/traffic_latest.json (aux.{key})"
    if int(t.get("P", -1)) != P or int(t.get("ncell", -1)) != ncell or t.get("dtype", "f64") != dtype:
        return None, "profiled workload differs from this run"
    files = tuple(t.get("kernel_src_files", ()))
    if not files or t.get("kernel_src_sha") != kernel_src_sha(files) or not lib_built_from_tree():
        return None, "the step's kernel sources differ from the profiled ones"
    return float(t["hbm_bytes_per_step"]), (f"sum over the step's launches of the per-launch (2 FETCH_SIZE + WRITE_SIZE) x 1024 of separate rocprofv3 --pmc "
                                            f"passes ({t.get('source')}): {t.get('breakdown')}; same kernel sources and compile flags")


def config4_self_plan(n1, permuted=False, seed=0):
    """``utils.config4_self_plan`` of the package (the halo plan of one config-4 rank that is its own neighbour)."""
    import fusgpu_loader

    return fusgpu_loader.submodule("utils").config4_self_plan(n1, permuted, seed)


def measure_scatter(device, dtype_np, kinds=("peer", "native", "torch"), reps=100, P=4, cells=54):
    """The reference's third timing script (numba-cpu/time_scatterer.py:126-210: scatter_reverse / scatter_forward alone, one
    call at a time between two clock reads) at N = 1: a rank that is its own neighbour with config-4 message sizes.  Per
    transport and direction: ``us_per_call_sync`` = mean / std of host clock around call + device synchronise (the reference's
    protocol -- its closures block), ``us_per_call_stream`` = ``reps`` calls back to back between one HIP-event pair."""
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    scat = fusgpu_loader.submodule("scatterer")
    n1 = P * cells + 1
    od, gd, N = config4_self_plan(n1)
    ng = int(od[1][0])
    tdt = torch.float64 if np.dtype(dtype_np) == np.float64 else torch.float32
    buf = torch.randn(N + ng, dtype=tdt, device=device)
    out = {"workload": f"one rank, its own neighbour, config-4 messages ({ng} elements = {ng * np.dtype(dtype_np).itemsize / 1e6:.2f} MB per direction: "
                       f"3 faces of {n1 * n1}, 3 edges of {n1}, 1 corner), vector of {N + ng} dofs", "reps": reps, "transports": {}}
    own_pg = False
    for kind in kinds:
        comm = None
        try:
            if kind == "torch":
                if not dist.is_initialized():
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", str(_free_port()))
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
                    own_pg = True
                comm = scat.TorchComm()
            else:
                comm = scat.NativeComm(transport="peer" if kind == "peer" else "rccl")
            row = {}
            for dname, mk in (("scatter_forward", scat.scatter_forward), ("scatter_reverse", scat.scatter_reverse)):
                sc = mk(comm, od, gd, N, dtype_np)
                for _ in range(3):
                    sc(buf)
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    sc(buf)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    sc(buf)
                e1.record()
                torch.cuda.synchronize()
                row[dname] = {"us_per_call_sync_mean": float(np.mean(ts)) * 1e6, "us_per_call_sync_std": float(np.std(ts)) * 1e6,
                              "us_per_call_sync_min": float(np.min(ts)) * 1e6, "us_per_call_stream": e0.elapsed_time(e1) / reps * 1e3}
                if hasattr(sc, "status"):
                    row[dname]["failed_waits"] = int(sc.status().get("failures", 0))
                if hasattr(sc, "close"):
                    sc.close()
                buf.normal_()  # reverse adds: keep the values bounded
            row["transport"] = TRANSPORT_TEXT[kind]
            out["transports"][kind] = row
        except Exception as e:  # noqa: BLE001
            out["transports"][kind] = {"error": repr(e)}
            log(f"scatter timing, transport {kind!r}: {e!r}")
        finally:
            if comm is not None and hasattr(comm, "close"):
                try:
                    comm.close()
                except Exception:  # noqa: BLE001
                    pass
    if own_pg:
        dist.destroy_process_group()
    # CPU beside it: the oracle's numpy restatement of the reference's closures (pack, copy, unpack), same plan
    try:
        from oracle import oracle_np

        h = np.random.default_rng(0).standard_normal(N + ng)
        cpu = {}
        for dname, fn in (("scatter_forward", oracle_np.scatter_forward_all), ("scatter_reverse", oracle_np.scatter_reverse_all)):
            fn([h], [od], [gd], [N])
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                fn([h], [od], [gd], [N])
                ts.append(time.perf_counter() - t0)
            cpu[dname] = {"us_per_call_mean": float(np.mean(ts)) * 1e6, "us_per_call_std": float(np.std(ts)) * 1e6}
        out["cpu_baseline"] = dict(cpu, kind="port", cores=1, impl="oracle/oracle_np.py (numba-cpu/scatterer.py:78-207 restated, no MPI: one rank)",
                                   sample="the same plan, 10 calls per direction")
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline"] = None
        log(f"scatter cpu leg failed: {e!r}")
    return out


def measure_sustained(step_fn, alg_bytes, total=2500, windows=10):
    """>= 0.5 s of back-to-back headline applies: ms per apply overall and per sub-window (one HIP event between two
    windows), device clocks before / after where sysfs shows them."""
    import glob

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


def first_comm(args, scat, world, device):
    """(comm, kind) of the first candidate transport that comes up on every rank."""
    for kind in transport_candidates(args):
        if ":" in kind:  # arena-memory variants of a transport are retried by the apply modes' own halo check, not here
            continue
        comm, why = make_comm(kind, scat, world, device)
        if comm is not None:
            return comm, kind
        log(f"halo transport {kind!r} not available ({why}); trying the next one")
    raise SystemExit("no halo transport came up")


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


def compare_with_oracle(y_gpu, y_ref, dtype, what):
    """{rel_l2, rel_max, sum_y, ...}: the GPU result of the timed run against the oracle's on the same inputs.  Tolerance:
    SURVEY 8d (fp64 rel l2 <= 1e-12, max-abs / max <= 1e-11; fp32 1e-5 / 1e-4)."""
    y_gpu = np.asarray(y_gpu, dtype=np.float64)
    d = y_gpu - y_ref
    nrm, mx = float(np.linalg.norm(y_ref)), float(np.max(np.abs(y_ref))) if y_ref.size else 0.0
    rel_l2 = float(np.linalg.norm(d)) / max(nrm, 1e-300)
    rel_max = (float(np.max(np.abs(d))) if d.size else 0.0) / max(mx, 1e-300)
    tol_l2, tol_max = (1e-12, 1e-11) if dtype == "f64" else (1e-5, 1e-4)
    return {"rel_l2": rel_l2, "rel_max": rel_max, "sum_y": float(y_gpu.sum()), "sum_y_oracle": float(y_ref.sum()), "norm_y_oracle": nrm,
            "tol_rel_l2": tol_l2, "tol_rel_max": tol_max, "ok": bool(np.isfinite(rel_l2) and rel_l2 <= tol_l2 and rel_max <= tol_max and nrm > 0.0),
            "what": what, "oracle": "oracle/fus_oracle.c (C restatement of numba-cpu/operators.py), same x, constants, geometry factors, dofmap"}


def measure_halo_proxy(op, mesh, cc_d, G_d, dm_d, y_d, device, dt_np, P, cells, kinds=("peer", "native"), rounds=5, reps=40):
    """north_star's "< 5 % halo-exchange overhead" on the only proxy a one-GPU box has (tools/overlap_probe.py --paired, the
    measurement DESIGN 4.3 quotes): ONE rank that is its own neighbour with the messages of a config-4 rank (3 faces + 3 edges
    + 1 corner of a 54^3-cell P = 4 block: 1.14 MB per direction; it sends AND receives every message -- the upper bound of what
    a rank of a 2x2x2 partition does), 8 590 boundary cells first.  ``rounds`` alternating rounds of ``reps`` applies each of
    (single launch over all cells | HaloApply's own launch schedule without exchange | the same with both exchanges);
    medians, and medians of the per-round differences.  Per transport: PEER (the default) and RCCL grouped send / recv."""
    import torch

    import fusgpu_loader

    scat = fusgpu_loader.submodule("scatterer")
    n1 = P * cells + 1
    od, gd, N = config4_self_plan(n1)
    ng = int(od[1][0])
    if N + ng != mesh.ndofs:
        raise ValueError("halo proxy: the self-neighbour plan is sized for the serial box")

    class _RankView:  # the attributes HaloApply reads from a mesh
        pass

    m = _RankView()
    nb = cells * cells + cells * (cells - 1) + (cells - 1) * (cells - 1)  # the cells on three faces of a cells^3 block ...
    nb = (nb + 9) // 10 * 10  # ... in whole batches of the plan (8 590 at config 4, as tools/overlap_probe.py)
    m.num_boundary_cells, m.ncells, m.nlocal, m.dofmap, m.index_map = nb, mesh.ncells, N, mesh.dofmap, None
    tdt = torch.float64 if np.dtype(dt_np) == np.float64 else torch.float32
    xg = torch.randn(mesh.ndofs, dtype=tdt, device=device)
    out = {"workload": f"one rank, its own neighbour, config-4 messages ({ng} elements per direction), {nb} boundary cells of {mesh.ncells}; "
                       f"{rounds} alternating rounds x {reps} applies, medians of per-round differences",
           "rounds": rounds, "reps": reps, "transports": {}}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for kind in kinds:
        comm = halo = None
        try:
            comm = scat.NativeComm(transport="peer" if kind == "peer" else "rccl")
            halo = scat.HaloApply(m, op, comm, dt_np, plan=(od, gd))
            halo.prepare(xg, cc_d, G_d, dm_d)
            fns = (("single", lambda: op(xg, cc_d, y_d, G_d, dm_d)),
                   ("schedule", lambda: halo.apply_no_exchange(xg, cc_d, y_d, G_d, dm_d)),
                   ("halo", lambda: halo.apply(xg, cc_d, y_d, G_d, dm_d)))
            res = {k: [] for k, _ in fns}
            for _ in range(rounds):
                for k, fn in fns:
                    res[k].append(timed(fn))
            a = {k: np.array(v) for k, v in res.items()}
            single = float(np.median(a["single"]))
            d_halo, d_split = float(np.median(a["halo"] - a["single"])), float(np.median(a["schedule"] - a["single"]))
            out["transports"][kind] = {
                "transport": TRANSPORT_TEXT[kind], "schedule": halo.schedule_kind, "lead_cells": halo.lead_cells,
                "single_launch_us": single, "schedule_without_exchange_us": float(np.median(a["schedule"])),
                "with_both_exchanges_us": float(np.median(a["halo"])), "exposed_us": d_halo, "exposed_pct": 100.0 * d_halo / single,
                "split_us": d_split, "exchanges_us": float(np.median(a["halo"] - a["schedule"])),
                "exposed_us_per_round": [float(v) for v in (a["halo"] - a["single"])], "failed_waits": int(halo.health())}
        except Exception as e:  # noqa: BLE001
            out["transports"][kind] = {"error": repr(e)}
            log(f"halo proxy, transport {kind!r}: {e!r}")
        finally:
            try:
                torch.cuda.synchronize()
                if halo is not None:
                    halo.fwd.close(), halo.rev.close()
                if comm is not None:
                    comm.close()
            except Exception:  # noqa: BLE001
                pass
    return out


def secondary_summary(out):
    """The scalars of ``aux`` that matter, mirrored into ``roofline.secondary`` (<= 1 kB): the driver's record keeps
    ``config``, ``roofline`` and ``cpu_baseline`` verbatim and only the NAME of ``aux``."""
    aux = out.get("aux") or {}
    r3, r1 = (lambda v: None if v is None else round(float(v), 3)), (lambda v: None if v is None else round(float(v), 1))
    sec = {}

    def line(key, name):
        a = aux.get(name)
        if not a:
            return
        rf = a.get("roofline") or {}
        alg = rf.get("algorithmic_bytes_per_step") or ((rf.get("algorithmic_bytes_per_cell") or 0) * (rf.get("cells_per_launch") or 0)) or rf.get("algorithmic_bytes_per_launch")
        tr = rf.get("traffic")
        sec[key] = {"ms": None if rf.get("kernel_ms") is None else round(float(rf["kernel_ms"]), 4), "frac": r3(rf.get("frac")),
                    "tr": r3(tr / alg) if (tr and alg) else None}

    line("mass", "mass")
    mrf = (aux.get("mass") or {}).get("roofline") or {}
    if mrf.get("static_detJ_kernel_ms"):
        sec["mass_static"] = {"ms": round(float(mrf["static_detJ_kernel_ms"]), 4), "frac": r3(mrf.get("static_detJ_frac"))}
    line("mass_diag", "mass_cached_diagonal")
    line("geom", "stiffness_in_kernel_geometry")
    line("rk4", "rk4_step")
    line("rk4_geom", "rk4_step_in_kernel_geometry")
    line("westervelt", "westervelt_step")
    line("westervelt_geom", "westervelt_step_in_kernel_geometry")
    line("westervelt_1g", "westervelt_step_single_gather")
    su = aux.get("sustained")
    if su:
        sec["sustained"] = {"ms": round(float(su["ms_per_apply"]), 4), "frac": r3(su["frac_of_hbm_roofline"])}
    hp = (aux.get("halo_proxy") or {}).get("transports") or {}
    if hp:
        sec["halo_proxy"] = {k: ({"us": r1(v.get("exposed_us")), "pct": r1(v.get("exposed_pct"))} if "exposed_us" in v else {"error": True})
                             for k, v in hp.items()}
    sc = (aux.get("scatter") or {}).get("transports") or {}
    if "peer" in sc and "scatter_forward" in sc["peer"]:
        sec["scatter_peer_us"] = [r1(sc["peer"]["scatter_forward"]["us_per_call_sync_mean"]), r1(sc["peer"]["scatter_reverse"]["us_per_call_sync_mean"])]
    ck = out.get("check")
    if ck:
        sec["check"] = {"rel_l2": float(f"{ck['rel_l2']:.2e}"), "ok": ck["ok"]}
    return sec


def load_traffic(P, ncell, sha, dtype="f64"):
    """(per-launch HBM bytes, source) from the committed rocprofv3 PMC passes (profiles/), or
    (None, reason).  PMC counters cannot be read from inside the run, so this is a REPLAYED figure:
    it is reported only when the profiled library is the one loaded now (same hash), and the
    line names its source."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except Exception:
        return None, "no profiles/traffic_latest.json"
    if int(t.get("P", -1)) != P or int(t.get("ncell", -1)) != ncell or t.get("dtype", "f64") != dtype:
        return None, "profiled workload differs from this run"
    if t.get("lib_sha") == sha:
        return float(t["hbm_bytes_per_launch"]), f"replayed from {t.get('source')} (rocprofv3 --pmc, same library hash)"
    ks = kernel_src_sha()
    if ks is not None and t.get("kernel_src_sha") == ks and lib_built_from_tree():
        return float(t["hbm_bytes_per_launch"]), (f"replayed from {t.get('source')} (rocprofv3 --pmc; library {t.get('lib_sha')} then, {sha} now: "
                                                  f"same kernel sources and compile flags {ks}, the library differs elsewhere)")
    return None, f"profiled library {t.get('lib_sha')} is not the loaded one ({sha}) and the kernel's sources differ"


def rk4_step_bytes(P, T, ncells, ndofs, nfacets_source, nfacets_absorbing, mode, affine, in_kernel_geometry, single_gather=False):
    """Algorithmic HBM bytes of ONE fused RK4 step (4 stages), DESIGN.md section 6:
    linear:      4 x [cell pass + facet terms] + 41 vector touches (csrc/rk4.hpp: FIRST 9 + MIDDLE 12 + MIDDLE 12 + LAST 8)
    Westervelt:  4 x [cell pass: stiffness part, two gathers unless c4/c3 is uniform] + 4 x 15 vector touches
                 (csrc/westervelt.hpp rk4_stage_nl2_kernel; + 1 per stage for w when the pass is single-gather)
    cell pass per cell: G (or the 48-byte affine record, or vertex ids + coordinates) + dofmap + x once per gather +
    y read-modify-write + constants;  facet terms per facet: detJ + dofmap + y RMW (+ x for the absorbing set)."""
    n = P + 1
    nd = n**3
    if mode == "rk4":
        if affine:
            cell = 48 + 4 * nd + 3 * T * P**3 + T
        elif in_kernel_geometry:
            cell = geom_bytes_per_cell(P, T)
        else:
            cell = stiffness_bytes_per_cell(P, T)
        touches = 41
    else:
        gathers = 1 if single_gather else 2
        geo = (32 + 3 * T) if in_kernel_geometry else 6 * nd * T
        cell = geo + 4 * nd + gathers * T * P**3 + 2 * T * P**3 + gathers * T
        touches = 4 * (15 + (1 if single_gather else 0))
    facets = nfacets_source * n * n * (T + 4 + 2 * T) + nfacets_absorbing * n * n * (T + 4 + 3 * T)
    return {"cell_pass_bytes_per_cell": cell, "vector_touches_per_step": touches,
            "bytes_per_step": 4 * (ncells * cell + facets) + touches * T * ndofs}


def measure_rk4(args, rank, world, device, mode, perturbed, in_kernel_geometry, steps, warmup, comm=None, cpu_leg=False, single_gather=False):
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
    solver.rk4(0.0, tf, dts, max_steps=max(1, warmup))
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
    geo_kernel = bool(getattr(solver, "in_kernel_geometry", False))
    model = rk4_step_bytes(P, T, mesh.ncells, mesh.ndofs, int(solver.fdm1.shape[0]), int(solver.fdm2.shape[0]), mode,
                           bool(solver.affine), geo_kernel, single_gather)
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
            cpu = cpu_baseline_rk4(P, mesh, solver, dts)
        except Exception as e:  # noqa: BLE001
            log(f"rk4 cpu_baseline failed: {e!r}")
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
                   "lib_sha": lib_sha(), "lib_built_from_tree": lib_built_from_tree()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "whole fused RK4 step: 4 x (cell pass + facet_terms_kernel + rk4_stage kernel)",
                     "kernel_ms": dev_ms, "kernel_ms_how": "one HIP-event pair around the K steps of the timed region / K",
                     "algorithmic_bytes_per_step": model["bytes_per_step"], "cell_pass_bytes_per_cell": model["cell_pass_bytes_per_cell"],
                     "vector_touches_per_step": model["vector_touches_per_step"], "cells_per_launch": mesh.ncells},
        "cpu_baseline": cpu,
    }
    if rehearsal():
        out.update(valid=False, rehearsal="ranks share the visible GPU(s): NOT a measurement")
    del solver
    return out


def bench_rk4(args, rank, world, device):
    """Auxiliary metric (not the headline): ``--mode rk4`` / ``--mode westervelt``."""
    import torch.distributed as dist

    import fusgpu_loader

    scat = fusgpu_loader.submodule("scatterer")
    comm = first_comm(args, scat, world, device)[0] if world > 1 else None
    out = measure_rk4(args, rank, world, device, args.mode, args.perturbed, args.in_kernel_geometry, args.steps, args.warmup, comm,
                      cpu_leg=not args.no_cpu_baseline, single_gather=args.single_gather)
    if rank == 0:
        emit(out)
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
    ap.add_argument("--halo-compare", action="store_true",
                    help="N > 1: after the timed region, time the apply over EVERY transport that comes up (peer, native = RCCL) in alternating "
                         "rounds in this one process and put each one's exposed cost in config.halo_compare")
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
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

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
        return bench_rk4(args, rank, world, device)
    if args.mode == "scatter":
        if world != 1:
            raise SystemExit("--mode scatter is the N = 1 self-neighbour line (at N > 1 the exchange is inside every other mode's step)")
        sc = measure_scatter(device, np.float64 if args.dtype == "f64" else np.float32, reps=max(1, args.steps), P=args.degree, cells=args.cells)
        first = next((v for v in sc["transports"].values() if "scatter_forward" in v), None)
        emit({"metric": "scatter_forward_reverse_us", "value": None if first is None else first["scatter_forward"]["us_per_call_stream"], "unit": "us",
              "n_gpus": 1, "steps": args.steps, "warmup": 3, "ms_per_step": None if first is None else first["scatter_forward"]["us_per_call_stream"] * 1e-3,
              "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
              "config": {"workload": sc["workload"], "lib_sha": lib_sha(), "lib_built_from_tree": lib_built_from_tree()},
              "scatter": sc, "roofline": None, "cpu_baseline": sc.get("cpu_baseline")})
        return

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
        ipc_memory_env = os.environ.get("FUS_IPC_MEMORY")
        for kind in transport_candidates(args):
            base, _, arena_kind = kind.partition(":")
            if arena_kind:
                os.environ["FUS_IPC_MEMORY"] = arena_kind
            elif ipc_memory_env is None:
                os.environ.pop("FUS_IPC_MEMORY", None)
            else:
                os.environ["FUS_IPC_MEMORY"] = ipc_memory_env
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
                arena = halo.fwd.status().get("arena_memory") if (err is None and base == "peer" and hasattr(halo.fwd, "status")) else None
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
                if base in os.environ.get("FUS_BENCH_TEST_REJECT", "").split(","):  # test hook: exercise the fall-back path
                    verdict = dict(verdict, ok=False, rejected_by="FUS_BENCH_TEST_REJECT")
            else:
                entry["bring_up_errors"] = {v["rank"]: v["error"] for v in bring_up if v["error"] is not None}
            if verdict is not None and verdict["ok"]:
                transport, halo_check = kind, verdict
                tried.append(dict(entry, result="ok"))
                if rank == 0:
                    log(f"halo transport {kind!r}: came up on all {world} ranks, halo check passed ({verdict}); arena memory by rank: {entry['arena_memory_by_rank']}; CHOSEN")
                break
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
        if args.halo_compare:
            try:
                halo_compare = compare_transports(args, rank, world, device, scat, mesh, op, dt, x_d, cc_d, y_d, G_d, dm_d, transport.partition(":")[0], halo, kern_ms)
            except Exception as e:  # noqa: BLE001
                log(f"rank {rank}: --halo-compare failed: {e!r}")
                halo_compare = {"error": repr(e)}
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
        kname = ops.mass_kernel_name(dm_d, mesh.ndofs, atomic=args.mass_atomic or args.exclusive or (halo is not None and not rows))
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
            "halo_transport": None if halo is None else TRANSPORT_TEXT[transport.partition(":")[0]] + (f" [arenas in {transport.partition(':')[2]} memory]" if ":" in transport else ""),
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
    if world == 1 and not use_dist and not (geom or mass) and not args.no_aux and not args.no_plan:
        # SURVEY 8d asks for the mass apply next to the stiffness apply, north_star names the RK4 step: both after the
        # timed region of the headline, same mesh, each with its own bytes contract (reference: numba-cpu/time_operators.py:176-268
        # times the operators in one script)
        out["aux"] = {}
        try:
            out["aux"]["mass"], out["aux"]["mass_cached_diagonal"] = aux_mass(
                args, P, T, dt, mesh, x_d, cc_d, y_d, dm_d, dphi_g, wts3, device, ops, pre, x if not args.no_cpu_baseline else None, cc)
        except Exception as e:  # noqa: BLE001  (an auxiliary line never breaks the headline)
            log(f"aux mass line failed: {e!r}")
            out["aux"]["mass"] = None
        try:  # >= 0.5 s of back-to-back headline applies (the timed region above is K launches: milliseconds)
            out["aux"]["sustained"] = measure_sustained(step, alg_bytes)
        except Exception as e:  # noqa: BLE001
            log(f"aux sustained line failed: {e!r}")
            out["aux"]["sustained"] = None
        try:  # north_star's "< 5 % halo overhead" on the one-GPU proxy: paired single launch | schedule | schedule + exchanges
            if mesh.ncells == args.cells**3:
                out["aux"]["halo_proxy"] = measure_halo_proxy(op, mesh, cc_d, G_d, dm_d, y_d, device, dt, P, args.cells)
        except Exception as e:  # noqa: BLE001
            log(f"aux halo_proxy line failed: {e!r}")
            out["aux"]["halo_proxy"] = None
        keys = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "cpu_baseline")
        try:  # the same apply with G formed in the kernel -- own bytes contract, own line
            gop = ops.stiffness_operator(P, D.flatten(), dt, geometry=(mesh.x_dofs, mesh.x_g, pts, wts))
            gop.prepare(dm_d) if hasattr(gop, "prepare") else None
            K = AUX_STEADY_LAUNCHES
            gms, gms_burst = timed_steady(lambda: gop(x_d, cc_d, y_d, None, dm_d), max(1, args.steps))
            gb = geom_bytes_per_cell(P, T)
            gach = mesh.ncells * gb / (gms * 1e-3) / 1e9
            gtr, gtr_src = aux_traffic("stiffness_in_kernel_geometry", P, mesh.ncells, args.dtype)
            out["aux"]["stiffness_in_kernel_geometry"] = {
                "metric": "stiffness_apply_in_kernel_geometry_dof_per_s", "value": mesh.ndofs_global / (gms * 1e-3), "unit": "DOF/s", "ms_per_step": gms, "steps": K,
                "roofline": {"bound": "hbm", "achieved": gach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gach / HBM_PEAK_GBS, "traffic": gtr, "traffic_source": gtr_src,
                             "kernel": "fus::stiffness_plan_geom_kernel", "kernel_ms": gms,
                             "kernel_ms_how": f"one HIP-event pair around {AUX_STEADY_LAUNCHES} back-to-back launches after {AUX_STEADY_WARM} untimed ones (steady state)",
                             "kernel_ms_first_burst": gms_burst, "first_burst_launches": max(1, args.steps),
                             "algorithmic_bytes_per_cell": gb, "cells_per_launch": mesh.ncells,
                             "bytes_contract": "no G array: dofmap + x once + y RMW + constant + 8 vertex ids + one vertex per cell (DESIGN 3.3); NOT the headline contract",
                             "bound_note": "float-atomic request rate of the flush, not HBM bytes (DESIGN 3.3 / 3.4)"},
                "cpu_baseline": None}
            del gop
        except Exception as e:  # noqa: BLE001
            log(f"aux in-kernel-geometry line failed: {e!r}")
            out["aux"]["stiffness_in_kernel_geometry"] = None
        for name, geo_k in (("rk4_step", False), ("rk4_step_in_kernel_geometry", True)):
            try:
                r = measure_rk4(args, rank, world, device, "rk4", True, geo_k, max(1, min(args.steps, 20)), 2, cpu_leg=not args.no_cpu_baseline)
                out["aux"][name] = {k: r[k] for k in keys}
            except Exception as e:  # noqa: BLE001
                log(f"aux {name} line failed: {e!r}")
                out["aux"][name] = None
        try:  # BASELINE config 5's step on one GPU: Westervelt, P = 6, 36^3 bowl-warped cells (10.2 M dofs), fused stage
            wargs = argparse.Namespace(**{**vars(args), "degree": 6, "cells": max(4, round(args.cells * 2 / 3))})  # 54 -> 36: the same dof count
            # the reference's G stream | the solver's default (in-kernel geometry, two-gather cell pass: any medium) | the single-gather form
            for name, geo_k, single in (("westervelt_step", False, False), ("westervelt_step_in_kernel_geometry", True, False),
                                        ("westervelt_step_single_gather", True, True)):
                try:
                    r = measure_rk4(wargs, rank, world, device, "westervelt", True, geo_k, max(1, min(args.steps, 20)), 2, cpu_leg=False, single_gather=single)
                    out["aux"][name] = {k: r[k] for k in keys}
                except Exception as e:  # noqa: BLE001
                    log(f"aux {name} line failed: {e!r}")
                    out["aux"][name] = None
        except Exception as e:  # noqa: BLE001
            log(f"aux westervelt_step lines failed: {e!r}")
            out["aux"]["westervelt_step"] = None
        try:  # the reference's third timing script (numba-cpu/time_scatterer.py), self-neighbour with config-4 messages
            out["aux"]["scatter"] = measure_scatter(device, dt, reps=100, P=P, cells=args.cells)
        except Exception as e:  # noqa: BLE001
            log(f"aux scatter line failed: {e!r}")
            out["aux"]["scatter"] = None
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
                nl = mesh.nlocal
                dd = (y_d[:nl].double() - y_ref_d[:nl].double())
                sums = torch.stack([(dd * dd).sum(), (y_ref_d[:nl].double() ** 2).sum(), y_d[:nl].double().sum(), y_ref_d[:nl].double().sum()]).to(coll_device(device))
                dist.all_reduce(sums)
                mx = torch.stack([dd.abs().max() if nl else dd.new_zeros(()), y_ref_d[:nl].double().abs().max() if nl else dd.new_zeros(()),
                                  torch.tensor(float(halo.health()), dtype=torch.float64, device=device)]).to(coll_device(device))
                dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                rel_l2 = float(sums[0].sqrt().item()) / max(float(sums[1].sqrt().item()), 1e-300)
                rel_max = float(mx[0].item()) / max(float(mx[1].item()), 1e-300)
                tol_l2, tol_max = (1e-12, 1e-11) if args.dtype == "f64" else (1e-5, 1e-4)
                check = {"rel_l2": rel_l2, "rel_max": rel_max, "sum_y": float(sums[2].item()), "sum_y_oracle": float(sums[3].item()),
                         "norm_y_oracle": float(sums[1].sqrt().item()), "tol_rel_l2": tol_l2, "tol_rel_max": tol_max,
                         "ok": bool(np.isfinite(rel_l2) and rel_l2 <= tol_l2 and rel_max <= tol_max and float(sums[1].item()) > 0 and float(mx[2].item()) == 0.0),
                         "what": "one apply after the timed region (same halo objects and kernels) into a zeroed y, owned dofs of all ranks  vs  the oracle's "
                                 "apply over each rank's cells, reverse-scattered", "oracle": "oracle/fus_oracle.c"}
        except Exception as e:  # noqa: BLE001
            log(f"result check could not run: {e!r}")
            check = {"ok": False, "error": repr(e), "rel_l2": float("nan"), "rel_max": float("nan"), "sum_y": float("nan")}
        out["check"] = check
        out["config"]["check"] = check  # the driver's record keeps ``config`` verbatim
        if not check["ok"]:
            out["valid"] = False
    if out.get("aux") is not None or check is not None:
        try:
            out["roofline"]["secondary"] = secondary_summary(out)
        except Exception as e:  # noqa: BLE001
            log(f"roofline.secondary failed: {e!r}")
    if rank == 0:
        emit(out)
    failed = check is not None and not check["ok"]
    if failed and rank == 0:
        log(f"RESULT CHECK FAILED: {check}")
    if use_dist:
        dist.destroy_process_group()
    if failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
