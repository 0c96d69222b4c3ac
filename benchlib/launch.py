"""Starting the ranks (``bench.py --gpus N`` without a launcher), the CPU dry run (gloo) and the one-GPU rehearsal's staged
communicator.  Nothing here touches the GPU before the children exist."""
import os
import sys
import time

import numpy as np

from .common import _free_port, emit, host_cores, log


def spawn_ranks(n, argv, script):
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
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script), *argv], env=env,
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
