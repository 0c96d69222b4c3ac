"""
The reference's communicator argument, accepted as it is.

The reference's drivers are started with ``mpirun`` and hand ``comm = MPI.COMM_WORLD`` (mpi4py) to
``scatter_forward / scatter_reverse(comm, ...)`` (cuda/scatterer.py:104-110, 191-197; cuda/demo_linear_box.py:41, 192,
206-207) and ``compute_scatterer_data`` hard-codes it (cuda/utils.py:62, 68).  MPI moves the halo DATA there; here the
data moves through libfusgpu.so's own transports (PEER: peer-mapped arenas + send / receive kernels; RCCL: grouped
send / recv), and a communicator is needed for the one-off BOOTSTRAP only:

  * PEER: an all-gather of the arena blobs (HIP IPC handles) per scatter closure, an "all ok?" vote around it;
  * RCCL: a broadcast of the 128-byte unique id, the same votes;
  * ``compute_scatterer_data``: one all-to-all of integer index lists.

``MpiBootstrap(comm)`` provides exactly that over anything that quacks like an ``mpi4py.MPI.Comm`` -- ``Get_rank``,
``Get_size`` and the pickle-based lower-case collectives ``allgather``, ``alltoall``, ``bcast``, ``barrier`` -- with the
method names ``scatterer.TorchComm`` has, so ``NativeComm(bootstrap=...)`` does not care which one it was given.  No
``torch.distributed`` process group is created or needed in an ``mpirun`` world, and mpi4py is never imported here: the
object is used through its methods only (tests drive it with a file-backed stand-in across real processes).
"""

from __future__ import annotations

import numpy as np

_REQUIRED = ("Get_rank", "Get_size", "allgather", "alltoall", "bcast")


def is_mpi_comm(obj) -> bool:
    """Duck test for an mpi4py-style communicator (``MPI.COMM_WORLD``, ``MPI.Intracomm``, a stand-in with the same methods)."""
    return all(callable(getattr(obj, name, None)) for name in _REQUIRED)


def world_if_available():
    """``mpi4py.MPI.COMM_WORLD`` if mpi4py is importable (the reference's default communicator, cuda/utils.py:62), else None.
    Importing mpi4py initialises MPI: done only when a caller asks for a default communicator and no torch group is up."""
    try:
        from mpi4py import MPI  # noqa: PLC0415
    except Exception:  # noqa: BLE001  (absent in this image; ImportError or a broken MPI install)
        return None
    return MPI.COMM_WORLD


class MpiBootstrap:
    """Set-up collectives of ``NativeComm`` / ``compute_scatterer_data`` over an mpi4py-style communicator."""

    backend = "mpi"

    def __init__(self, comm):
        if not is_mpi_comm(comm):
            raise TypeError(f"not an MPI communicator (needs {', '.join(_REQUIRED)}): {type(comm).__name__}")
        self.mpi = comm
        self.rank = int(comm.Get_rank())
        self.size = int(comm.Get_size())

    def all_ok(self, ok: bool) -> bool:
        """True iff ``ok`` on every rank (a set-up step fails on all ranks together or on none)."""
        return all(bool(v) for v in self.mpi.allgather(bool(ok)))

    def allgather_bytes(self, payload: bytes):
        return [bytes(b) for b in self.mpi.allgather(bytes(payload))]

    def bcast_bytes(self, payload: bytes, root: int = 0) -> bytes:
        return bytes(self.mpi.bcast(bytes(payload) if self.rank == root else None, root))

    def barrier(self):
        fn = getattr(self.mpi, "barrier", None) or getattr(self.mpi, "Barrier", None)
        if fn is not None:
            fn()
        else:
            self.mpi.allgather(0)

    def alltoallv_int64(self, send_np, send_counts, recv_counts):
        """The index exchange of ``compute_scatterer_data`` (cuda/utils.py:54-71 does it with Isend / Irecv): segment r of
        ``send_np`` goes to rank r; returns the received segments concatenated in rank order."""
        send = np.ascontiguousarray(send_np, dtype=np.int64)
        off = np.concatenate(([0], np.cumsum(np.asarray(send_counts, dtype=np.int64))))
        parts = self.mpi.alltoall([send[off[r]: off[r + 1]] for r in range(self.size)])
        for r, (p, c) in enumerate(zip(parts, recv_counts)):
            if len(p) != int(c):
                raise ValueError(f"index exchange: rank {r} sent {len(p)} indices, {int(c)} expected")
        return np.concatenate([np.asarray(p, dtype=np.int64) for p in parts]) if parts else np.zeros(0, dtype=np.int64)
