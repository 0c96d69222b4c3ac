"""
Ghost-dof halo exchange: ``scatter_reverse`` / ``scatter_forward`` with the
reference's call surface and semantics

    scatter_reverse(comm, owners_data, ghosts_data, N, float_type) -> scatter(buffer)
    scatter_forward(comm, owners_data, ghosts_data, N, float_type) -> scatter(buffer)
        cuda/scatterer.py:104-188, 191-277   (owners_data/ghosts_data: 3-element lists)
        numba-cpu/scatterer.py:78-141, 144-207 (4-element flat lists with offsets)

  reverse: my ghost values -> their owners, ADDED there          (ghost partial sums)
  forward: my owned values -> the ranks that ghost them, COPIED  (ghost refresh)
both in place on ``buffer`` (length nlocal + nghost, ghosts at ``[N:]``).

MI355X-first differences from the reference:
  * one fused pack and one fused unpack launch for ALL neighbours (index lists
    concatenated) instead of one tiny kernel per neighbour;
  * transport = one neighbour all-to-all-v as grouped RCCL send/recv over xGMI
    instead of per-neighbour MPI Isend/Irecv on device pointers; no
    host-blocking device synchronisation anywhere: ordering is by stream and
    events (pack -> exchange -> unpack), so the host runs ahead;
  * split-phase ``begin()`` / ``end()`` so interior-cell operator application
    overlaps the exchange (``HaloApply`` below).

``comm`` selects the transport:
  * an ``mpi4py.MPI.Comm`` -- what the reference's drivers pass (cuda/demo_linear_box.py:41, 206-207) -- is accepted as it
    is: it becomes the bootstrap of a ``NativeComm`` (``as_comm``; ``mpi_bootstrap.py``), no ``torch.distributed`` needed;
  * ``NativeComm`` (default of the drivers): the exchange lives in libfusgpu.so
    (csrc/halo_comm.hpp, C ABI ``fus_halo_*``): pack, ``ncclGroupStart ...
    ncclSend/ncclRecv ... ncclGroupEnd`` and unpack are issued from C++ on a
    library-owned high-priority stream; torch.distributed only broadcasts the
    RCCL unique id at start-up;
  * ``TorchComm``: ``torch.distributed.all_to_all_single`` (backend "nccl" == RCCL,
    or "gloo" for CPU tests) with the pack / unpack kernels launched from Python;
    ``kernels`` then selects their implementation: the HIP kernels of libfusgpu.so
    by default (GPU tensors; no CPU fallback) -- tests inject their own for
    gloo-on-CPU runs.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .utils import to_flat


class TorchComm:
    """Thin communicator over a ``torch.distributed`` process group
    (backend "nccl" == RCCL on ROCm; "gloo" for CPU tests)."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def alltoallv(self, send, send_counts, recv, recv_counts, async_op=False):
        """Neighbour all-to-all-v: ``send`` / ``recv`` are flat tensors whose
        consecutive segments (``*_counts[r]`` elements, zero for non-neighbours)
        go to / come from rank r."""
        return dist.all_to_all_single(
            recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=self.group, async_op=async_op
        )

    def alltoallv_int64(self, send_np, send_counts, recv_counts):
        """Set-up path (index exchange of compute_scatterer_data)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        send = torch.from_numpy(np.ascontiguousarray(send_np, dtype=np.int64)).to(dev)
        recv = torch.empty(int(np.sum(recv_counts)), dtype=torch.int64, device=dev)
        self.alltoallv(send, [int(c) for c in send_counts], recv, [int(c) for c in recv_counts])
        return recv.cpu().numpy()

    def barrier(self):
        dist.barrier(group=self.group)

    def all_ok(self, ok: bool) -> bool:
        """True iff ``ok`` on EVERY rank (one all-reduce): lets a set-up step fail on all ranks together instead of
        leaving the healthy ones inside the next collective."""
        dev = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item() == 1.0)

    def bcast_bytes(self, payload: bytes, root: int = 0) -> bytes:
        """``payload`` of group rank ``root`` on every rank (same length everywhere: the 128-byte RCCL unique id)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone().to(dev)
        dist.broadcast(t, src=dist.get_global_rank(self.group, root) if self.group is not None else root, group=self.group)
        return bytes(t.cpu().numpy().tobytes())

    def allgather_bytes(self, payload: bytes):
        """Every rank's ``payload`` (host bytes of any length), as a list indexed by rank."""
        dev = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n) for _ in range(self.size)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(t.item()) for t in sizes]
        mx = max(max(sizes), 1)
        buf = torch.zeros(mx, dtype=torch.uint8)
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if payload else buf[:0]
        out = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(self.size)]
        dist.all_gather(out, buf.to(dev), group=self.group)
        return [bytes(t.cpu().numpy()[:sz].tobytes()) for t, sz in zip(out, sizes)]


class NativeComm:
    """Communicator owned by libfusgpu.so (RCCL over xGMI; csrc/halo_comm.hpp).

    ``NativeComm()``: one rank per process / GPU.  The 128-byte RCCL unique id is created on rank 0
    and broadcast through the default ``torch.distributed`` group (any backend: it is 128 bytes of
    host data), which also carries the one-off integer index exchange of
    ``compute_scatterer_data``; in a 1-rank world no process group is needed.
    ``NativeComm(local=(world_id, nranks, rank))``: all ranks in THIS process (tests on a one-GPU
    box), transport = stream-ordered device copies; every rank's ``begin`` of an exchange must be
    called before any rank's ``end``.

    ``transport="peer"`` (csrc/halo_ipc.hpp): no RCCL.  Each scatter closure owns a receive arena in uncached
    device memory whose HIP IPC handle goes once to its neighbours (all-gathered through ``torch.distributed``, any
    backend); an exchange is a send kernel that stores straight into the neighbours' arenas and a receive kernel that
    waits for a sequence flag -- small kernels that run NEXT TO a chip-filling operator launch, which RCCL's
    264-register kernel does not.  With ``local=...`` the ranks of one process use the same protocol (their arenas
    are plain pointers to each other); the closures connect at their first exchange, when every rank has been built.
    ``hosted=[ranks]`` with ``local=...`` and ``transport="peer"``: THIS process drives only those ranks of the world and
    the other ranks live in other processes of the ``torch.distributed`` group (one process driving several GPUs, or --
    tests -- the 8-rank 2x2x2 partition on 4 processes where the pool allows no more): the processes all-gather the
    arena handles of their ranks once per closure; neighbours of the same process are reached through plain pointers,
    the others through HIP IPC mappings, by the same kernels.  Every process must build its closures in the same order."""

    _peer_local = {}  # (world_id, halo index) -> {rank: blob}: in-process PEER worlds
    _peer_gathered = {}  # (world_id, halo index) -> {rank: blob} of ALL processes (hybrid worlds: gathered once per process)

    def __init__(self, group=None, local=None, transport="rccl", hosted=None, bootstrap=None):
        """``bootstrap``: the object that carries the one-off set-up collectives (rank / size, the votes, the all-gather of
        the arena handles or the broadcast of the RCCL id, the index exchange) instead of the default ``torch.distributed``
        group -- ``mpi_bootstrap.MpiBootstrap(MPI.COMM_WORLD)`` in an ``mpirun`` world (``as_comm`` builds it from a raw
        MPI communicator)."""
        import ctypes as C

        if transport not in ("rccl", "peer"):
            raise ValueError(f"transport must be 'rccl' or 'peer', got {transport!r}")
        lib = _lib.load()
        self._lib = lib
        self.handle = C.c_void_p()
        self._torch = None
        self.transport = transport
        self._stream = None
        self._world_id = None
        self._nhalos = 0
        self._hosted = None
        if local is not None:
            world_id, self.size, self.rank = (int(v) for v in local)
            self._world_id = world_id
            if hosted is not None:
                if transport != "peer":
                    raise ValueError("hosted= needs transport='peer'")
                self._hosted = sorted(int(r) for r in hosted)
                if self.rank not in self._hosted:
                    raise ValueError(f"rank {self.rank} is not among the hosted ranks {self._hosted}")
                if len(self._hosted) < self.size:
                    if not (dist.is_available() and dist.is_initialized()):
                        raise _lib.FusGpuError("NativeComm(hosted=...): the other ranks' arena handles travel over torch.distributed")
                    self._torch = TorchComm(group)
            if transport == "peer":
                self.backend = "peer-local"
                _lib.check(lib.fus_comm_create_peer(self.size, self.rank, C.byref(self.handle)), "fus_comm_create_peer")
            else:
                self.backend = "local"
                _lib.check(lib.fus_comm_create_local(world_id, self.size, self.rank, C.byref(self.handle)), "fus_comm_create_local")
            return
        if bootstrap is not None:
            self._torch = bootstrap  # same methods as TorchComm (all_ok, allgather_bytes, bcast_bytes, barrier, alltoallv_int64)
            self.rank, self.size = int(bootstrap.rank), int(bootstrap.size)
        elif dist.is_available() and dist.is_initialized():
            self._torch = TorchComm(group)
            self.rank, self.size = self._torch.rank, self._torch.size
        else:
            self.rank, self.size = 0, 1
        if transport == "peer":
            self.backend = "peer"
            _lib.check(lib.fus_comm_create_peer(self.size, self.rank, C.byref(self.handle)), "fus_comm_create_peer")
            return
        self.backend = "rccl"
        # Bootstrap that fails on ALL ranks or on none: (1) every rank probes librccl (dlopen + ncclGetUniqueId; only
        # rank 0's id is used) and the ranks agree on the outcome BEFORE anything is broadcast; (2) rank 0's id is
        # broadcast; (3) ncclCommInitRank, then the ranks agree again before any of them proceeds.
        buf = C.create_string_buffer(128)
        rc = lib.fus_comm_unique_id(buf)
        err = None if rc == 0 else f"fus_comm_unique_id: {lib.fus_error_string(rc).decode()}: {(lib.fus_comm_last_error(None) or b'?').decode()}"
        if self.size > 1 and not self._torch.all_ok(err is None):
            raise _lib.FusGpuError(f"RCCL is not usable on every rank (this rank: {err or 'ok'}): no native communicator")
        if err is not None:
            raise _lib.FusGpuError(err)
        raw = bytes(buf.raw)
        if self.size > 1:
            raw = self._torch.bcast_bytes(raw, 0)
        rc = lib.fus_comm_create(raw, self.size, self.rank, C.byref(self.handle))
        err = None if rc == 0 else f"fus_comm_create: {lib.fus_error_string(rc).decode()}: {(lib.fus_comm_last_error(None) or b'?').decode()}"
        if self.size > 1 and not self._torch.all_ok(err is None):
            if err is None:
                self.close()
            raise _lib.FusGpuError(f"ncclCommInitRank did not succeed on every rank (this rank: {err or 'ok'})")
        if err is not None:
            raise _lib.FusGpuError(err)

    def alltoallv_int64(self, send_np, send_counts, recv_counts):
        """Set-up path (index exchange of compute_scatterer_data): through torch.distributed."""
        if self._torch is None:
            raise _lib.FusGpuError("NativeComm: the index exchange of a multi-rank world needs a bootstrap (torch.distributed or an MPI communicator)")
        return self._torch.alltoallv_int64(send_np, send_counts, recv_counts)

    def barrier(self):
        if self._torch is not None:
            self._torch.barrier()

    def allgather_floats(self, values):
        """Every rank's ``values`` (a few host floats) as an array [size, len(values)] -- set-up agreements of the drivers
        (a min / max over the ranks) through whatever carries this communicator's bootstrap.  One-rank world or all ranks in
        this process without a bootstrap: the local values alone."""
        import struct

        v = [float(x) for x in values]
        if self._torch is None or self.size == 1:
            return np.asarray([v])
        blobs = self._torch.allgather_bytes(struct.pack(f"<{len(v)}d", *v))
        return np.asarray([struct.unpack(f"<{len(v)}d", b) for b in blobs])

    def close(self):
        if self.handle:
            if self._lib.fus_comm_destroy(self.handle) == 0:  # refused while halo objects are alive
                self.handle = None
        if self._world_id is not None and self.transport == "peer":
            for key in [k for k in NativeComm._peer_local if k[0] == self._world_id]:
                NativeComm._peer_local[key].pop(self.rank, None)
                if not NativeComm._peer_local[key]:
                    del NativeComm._peer_local[key]
                    NativeComm._peer_gathered.pop(key, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        """The library-owned high-priority stream the exchanges run on, as a torch stream (``fus_comm_stream``)."""
        if self._stream is None:
            ptr = self._lib.fus_comm_stream(self.handle)
            self._stream = torch.cuda.ExternalStream(int(ptr)) if ptr else None
        return self._stream

    def fork(self, lazy=False, attach=False):
        """Order the communicator's stream after the caller's current stream, without an event (``fus_comm_fork_ex``).
        ``lazy`` (PEER transport): no wait kernel -- the first send kernel of the exchange the caller posts NEXT on the
        communicator's stream waits for the fork flag itself.  ``attach``: no signal kernel -- the next PLANNED operator
        launch on the caller's stream publishes the flag when it starts (``fork_flush()`` if none follows)."""
        _lib.check(self._lib.fus_comm_fork_ex(self.handle, _lib.stream_ptr(), (1 if lazy else 0) | (2 if attach else 0)), "fus_comm_fork", self.handle)

    def fork_flush(self):
        """Publish an attached fork signal that no planned launch has carried (``fus_comm_fork_flush``)."""
        _lib.check(self._lib.fus_comm_fork_flush(self.handle), "fus_comm_fork_flush", self.handle)

    def arm_join(self):
        """PEER transport: the last receive kernel of the exchange posted next publishes the join flag, so that ``join()``
        launches only its wait kernel (``fus_comm_arm_join``).  A no-op for the other transports."""
        _lib.check(self._lib.fus_comm_arm_join(self.handle), "fus_comm_arm_join", self.handle)

    def join(self):
        """Order the caller's current stream after the communicator's stream, without an event (``fus_comm_join``)."""
        _lib.check(self._lib.fus_comm_join(self.handle, _lib.stream_ptr()), "fus_comm_join", self.handle)

    def health(self):
        """Failed device-side waits (time-outs + poisoned flags) of every live halo object of this communicator and of its
        fork / join kernels: 0 = every exchange so far delivered (``fus_comm_health``; synchronises the exchange streams)."""
        import ctypes as C

        n = C.c_int64(0)
        _lib.check(self._lib.fus_comm_health(self.handle, C.byref(n)), "fus_comm_health", self.handle)
        return int(n.value)

    def health_detail(self):
        """``{"timeouts", "poisoned", "sync_timeouts"}``: what ``health()`` adds up."""
        import ctypes as C

        out = (C.c_int64 * 3)()
        _lib.check(self._lib.fus_comm_health_detail(self.handle, out), "fus_comm_health_detail", self.handle)
        return {"timeouts": int(out[0]), "poisoned": int(out[1]), "sync_timeouts": int(out[2])}

    def sync_timeouts(self):
        import ctypes as C

        n = C.c_int64(0)
        _lib.check(self._lib.fus_comm_sync_timeouts(self.handle, C.byref(n)), "fus_comm_sync_timeouts", self.handle)
        return int(n.value)

    # ---- PEER transport: hand a halo object's arena handle to its neighbours
    def _peer_connect(self, halo_handle, index, lazy_ok=True):
        """Connect halo object number ``index`` of this rank with the other ranks' object number ``index``.
        Returns False if (in-process world) some rank has not built its object yet: retried at the first exchange."""
        import ctypes as C

        lib = self._lib
        n = int(lib.fus_halo_ipc_blob_bytes(halo_handle))
        if n <= 0:
            raise _lib.FusGpuError("fus_halo_ipc_blob_bytes failed")
        buf = C.create_string_buffer(n)
        _lib.check(lib.fus_halo_ipc_export(halo_handle, buf), "fus_halo_ipc_export", self.handle)
        mine = bytes(buf.raw)
        if self._world_id is not None:
            reg = NativeComm._peer_local.setdefault((self._world_id, index), {})
            if self.rank in reg and reg[self.rank] != mine:  # a new world re-uses the id: drop the old world's handles
                reg.clear()
            reg[self.rank] = mine
            hosted = self._hosted if self._hosted is not None else list(range(self.size))
            if any(r not in reg for r in hosted):
                if lazy_ok:
                    return False
                raise _lib.FusGpuError(f"PEER halo {index}: only ranks {sorted(reg)} of {hosted} have built their closure")
            if len(hosted) < self.size:
                # hybrid world: the rank of this process that completes the set all-gathers the processes' blobs (a
                # collective: every process builds its closures in the same order); the others find them cached
                key = (self._world_id, index)
                got = NativeComm._peer_gathered.get(key)
                if got is None or any(got.get(r) != reg[r] for r in hosted):
                    import struct

                    payload = b"".join(struct.pack("<q", len(reg[r])) + reg[r] for r in hosted)
                    got = {}
                    for chunk in self._torch.allgather_bytes(payload):
                        off = 0
                        while off < len(chunk):
                            (n_,) = struct.unpack_from("<q", chunk, off)
                            b = chunk[off + 8: off + 8 + n_]
                            got[struct.unpack_from("<Iiii", b)[2]] = b  # IpcBlobHeader: magic, version, rank, ...
                            off += 8 + n_
                    NativeComm._peer_gathered[key] = got
                missing = [r for r in range(self.size) if r not in got]
                if missing:
                    raise _lib.FusGpuError(f"PEER halo {index}: no process hosts rank(s) {missing}")
                blobs = [got[r] for r in range(self.size)]
            else:
                blobs = [reg[r] for r in range(self.size)]
        elif self._torch is not None:
            blobs = self._torch.allgather_bytes(mine)
        else:
            blobs = [mine]
        keep = [C.create_string_buffer(b, len(b)) for b in blobs]
        arr = (C.c_void_p * len(keep))(*[C.cast(k, C.c_void_p) for k in keep])
        rc = lib.fus_halo_ipc_connect(halo_handle, len(keep), arr)
        if self._world_id is None and self._torch is not None and not self._torch.all_ok(rc == 0):
            # a rank that cannot map a neighbour's arena must not leave the others waiting for its messages
            detail = (lib.fus_comm_last_error(self.handle) or b"").decode() if rc != 0 else "ok here"
            raise _lib.FusGpuError(f"PEER halo {index}: mapping the neighbours' arenas did not succeed on every rank ({detail})")
        _lib.check(rc, "fus_halo_ipc_connect", self.handle)
        return True


def default_comm(group=None):
    """The communicator a driver should use at N > 1 (one process per GPU): the PEER transport of libfusgpu.so unless
    ``FUS_HALO=native`` (grouped RCCL send / recv) or ``FUS_HALO=torch`` (``all_to_all_single``).  Bootstrap: the
    ``torch.distributed`` group if one is initialised; otherwise ``MPI.COMM_WORLD`` when mpi4py is importable (a driver
    started with ``mpirun``, as the reference's are: cuda/demo_linear_box.py:41); otherwise a one-rank world.  Creation fails
    on all ranks or on none."""
    import os

    kind = os.environ.get("FUS_HALO", "peer")
    if dist.is_available() and dist.is_initialized():
        if kind == "torch":
            return TorchComm(group)
        return NativeComm(group, transport="peer" if kind == "peer" else "rccl")
    from . import mpi_bootstrap

    world = mpi_bootstrap.world_if_available()
    if world is not None:
        return as_comm(world)
    return NativeComm(transport="peer" if kind != "native" else "rccl")


_MPI_COMMS = {}  # id(MPI communicator) -> (the communicator itself, its NativeComm): one library communicator per MPI communicator


def as_comm(comm):
    """The package communicator behind whatever a driver passes as ``comm``:

      * ``NativeComm`` / ``TorchComm`` (or a stand-in with their ``rank`` / ``size`` / ``alltoallv`` members): itself;
      * an ``mpi4py.MPI.Comm`` -- what the reference's drivers pass (cuda/demo_linear_box.py:41, 206-207;
        cuda/scatterer.py:104-110): a ``NativeComm`` bootstrapped over it (``mpi_bootstrap.MpiBootstrap``), transport PEER
        unless ``FUS_HALO=native`` (RCCL); built once per MPI communicator (collectively: every rank must pass it at the same
        point, as every rank of the reference's driver reaches its ``scatter_reverse(comm, ...)`` line);
      * ``None``: ``None``.

    Anything else raises ``TypeError`` naming the accepted kinds."""
    import os

    from . import mpi_bootstrap

    if comm is None or isinstance(comm, (NativeComm, TorchComm)):
        return comm
    if mpi_bootstrap.is_mpi_comm(comm):
        hit = _MPI_COMMS.get(id(comm))
        if hit is None or hit[0] is not comm or not hit[1].handle:
            kind = os.environ.get("FUS_HALO", "peer")
            hit = (comm, NativeComm(transport="rccl" if kind == "native" else "peer", bootstrap=mpi_bootstrap.MpiBootstrap(comm)))
            _MPI_COMMS[id(comm)] = hit
        return hit[1]
    if isinstance(getattr(comm, "rank", None), int) and isinstance(getattr(comm, "size", None), int) and \
            (comm.size == 1 or callable(getattr(comm, "alltoallv", None))):
        return comm  # a TorchComm-shaped object (tests, bench.py's staged rehearsal communicator)
    raise TypeError(f"comm: expected a NativeComm, a TorchComm, or an MPI communicator (mpi4py.MPI.Comm: Get_rank / Get_size / allgather / "
                    f"alltoall / bcast), got {type(comm).__name__}")


class _NativeScatter:
    """scatter_forward / scatter_reverse closure over ``fus_halo_*`` (the exchange is issued from C++)."""

    def __init__(self, comm: NativeComm, owners_data, ghosts_data, N, float_type, reverse: bool, halo=None):
        import ctypes as C

        self.comm, self.N, self.reverse = comm, int(N), reverse
        self.dtype = _lib.torch_dtype(float_type)
        lib = _lib.load()
        self._lib = lib
        if halo is not None:  # share the plan + buffers of another closure (same vector never in flight twice)
            self._owner, self.handle = halo, halo.handle
        else:
            o_idx, o_size, _, o_ranks = to_flat(owners_data)
            g_idx, g_size, _, g_ranks = to_flat(ghosts_data)
            self.handle = C.c_void_p()
            self._owner = None

            def arr(a, dt):
                a = np.ascontiguousarray(a, dtype=dt)
                return a, a.ctypes.data_as(C.c_void_p)

            keep = [arr(o_ranks, np.int32), arr(o_size, np.int64), arr(o_idx, np.int64),
                    arr(g_ranks, np.int32), arr(g_size, np.int64), arr(g_idx, np.int64)]
            rc = lib.fus_halo_create(comm.handle, 8 if self.dtype == torch.float64 else 4, self.N, int(len(o_idx)),
                                     len(o_ranks), keep[0][1], keep[1][1], keep[2][1],
                                     len(g_ranks), keep[3][1], keep[4][1], keep[5][1], C.byref(self.handle))
            # PEER, one process per rank: building a closure continues with a collective (the all-gather of the arena
            # handles).  A rank whose object could not be created (out of memory, a bad plan) must not leave the others
            # inside that all-gather: agree on the outcome first, so that the failure is raised on every rank here.
            if getattr(comm, "transport", "rccl") == "peer" and comm._world_id is None and comm._torch is not None and comm.size > 1:
                if not comm._torch.all_ok(rc == 0) and rc == 0:
                    lib.fus_halo_destroy(self.handle)
                    self.handle = None
                    raise _lib.FusGpuError("fus_halo_create failed on another rank: no halo object is built on any rank")
            _lib.check(rc, "fus_halo_create", comm.handle)
            self._index = comm._nhalos
            comm._nhalos += 1
        self.nghost = None
        self.direct = bool(lib.fus_halo_is_direct(self.handle) == 1)
        self.active = True
        sfx = "reverse" if reverse else "forward"
        self._begin, self._end = getattr(lib, f"fus_halo_{sfx}_begin"), getattr(lib, f"fus_halo_{sfx}_end")
        self._whole = getattr(lib, f"fus_halo_{sfx}")
        # PEER transport: map the neighbours' arenas (collective all-gather of the handles; in-process worlds connect
        # at the first exchange, when all ranks exist)
        self._connected = True
        if halo is None and getattr(comm, "transport", "rccl") == "peer":
            self._connected = comm._peer_connect(self.handle, self._index)

    def _ensure_connected(self):
        owner = self._owner if self._owner is not None else self
        if not owner._connected:
            owner._connected = self.comm._peer_connect(owner.handle, owner._index, lazy_ok=False)

    def status(self):
        """PEER transport: ``{"timeouts": device-side waits that gave up (0 = healthy), ...}`` (synchronises)."""
        import ctypes as C

        if getattr(self.comm, "transport", "rccl") != "peer":
            return {"failures": 0, "timeouts": 0, "poisoned": 0, "dead": False}
        out = (C.c_int64 * 8)()
        _lib.check(self._lib.fus_halo_ipc_status(self.handle, out), "fus_halo_ipc_status", self.comm.handle)
        return {"failures": int(out[0]), "timeouts": int(out[4]), "poisoned": int(out[5]), "dead": bool(out[6]),
                "forward_posted": int(out[1]), "reverse_posted": int(out[2]),
                "arena_memory": ("fine-grained", "uncached", "ordinary")[int(out[3])],
                "fenced": bool(out[7])}  # FUS_IPC_FENCED=1 at creation: system-scope release / acquire around the flags

    def begin(self, buffer):
        _lib.require_device_tensor(buffer, self.dtype, "buffer")
        if not self._connected or (self._owner is not None and not self._owner._connected):
            self._ensure_connected()
        _lib.check(self._begin(self.handle, buffer.data_ptr(), _lib.stream_ptr()), "fus_halo_begin", self.comm.handle)
        return None

    def end(self, buffer, work=None):
        _lib.check(self._end(self.handle, buffer.data_ptr(), _lib.stream_ptr()), "fus_halo_end", self.comm.handle)

    def __call__(self, buffer):
        """``scatter(buffer)`` of the reference's closures: the whole exchange, in stream order on the caller's stream
        (``fus_halo_forward`` / ``fus_halo_reverse``: the PEER transport then runs its two kernels on that stream itself, no
        event edge to the communicator's stream and back)."""
        _lib.require_device_tensor(buffer, self.dtype, "buffer")
        if not self._connected or (self._owner is not None and not self._owner._connected):
            self._ensure_connected()
        _lib.check(self._whole(self.handle, buffer.data_ptr(), _lib.stream_ptr()), "fus_halo_exchange", self.comm.handle)

    def close(self):
        # a halo must be destroyed before its communicator; if the communicator is already gone
        # (interpreter shutdown order) the handle is dropped without touching it
        if self._owner is None and self.handle:
            if getattr(self.comm, "handle", None):
                self._lib.fus_halo_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipHaloKernels:
    """pack/unpack through the C ABI (csrc/halo.hpp)."""

    def __init__(self, dtype: torch.dtype):
        lib = _lib.load()
        suf = _lib.suffix(dtype)
        self.dtype = dtype
        self._pack_fwd = getattr(lib, f"fus_pack_fwd_{suf}")
        self._unpack_fwd = getattr(lib, f"fus_unpack_fwd_{suf}")
        self._pack_rev = getattr(lib, f"fus_pack_rev_{suf}")
        self._unpack_rev = getattr(lib, f"fus_unpack_rev_{suf}")

    def index_tensor(self, idx_np):
        dev = torch.device("cuda", torch.cuda.current_device())
        return torch.from_numpy(np.ascontiguousarray(idx_np, dtype=np.int64)).to(dev)

    def buffer(self, n):
        return torch.empty(int(n), dtype=self.dtype, device=torch.device("cuda", torch.cuda.current_device()))

    def _chk(self, t, name):
        _lib.require_device_tensor(t, self.dtype, name)

    def pack_fwd(self, in_, out, index):
        self._chk(in_, "buffer")
        _lib.check(self._pack_fwd(in_.data_ptr(), out.data_ptr(), index.data_ptr(), index.numel(), _lib.stream_ptr()), "fus_pack_fwd")

    def unpack_fwd(self, in_, out, index, N):
        self._chk(out, "buffer")
        _lib.check(self._unpack_fwd(in_.data_ptr(), out.data_ptr(), index.data_ptr(), index.numel(), int(N), _lib.stream_ptr()), "fus_unpack_fwd")

    def pack_rev(self, in_, out, index, N):
        self._chk(in_, "buffer")
        _lib.check(self._pack_rev(in_.data_ptr(), out.data_ptr(), index.data_ptr(), index.numel(), int(N), _lib.stream_ptr()), "fus_pack_rev")

    def unpack_rev(self, in_, out, index):
        self._chk(out, "buffer")
        _lib.check(self._unpack_rev(in_.data_ptr(), out.data_ptr(), index.data_ptr(), index.numel(), _lib.stream_ptr()), "fus_unpack_rev")


class _Scatter:
    """Callable returned by scatter_forward / scatter_reverse."""

    def __init__(self, comm, owners_data, ghosts_data, N, float_type, reverse: bool, kernels=None):
        self.comm = comm
        self.N = int(N)
        self.reverse = reverse
        tdt = _lib.torch_dtype(float_type)
        self.k = kernels if kernels is not None else HipHaloKernels(tdt)
        o_idx, o_size, _, o_ranks = to_flat(owners_data)
        g_idx, g_size, _, g_ranks = to_flat(ghosts_data)
        size = comm.size
        o_counts = [0] * size
        g_counts = [0] * size
        for r, c in zip(o_ranks, o_size):
            o_counts[int(r)] = int(c)
        for r, c in zip(g_ranks, g_size):
            g_counts[int(r)] = int(c)
        # the flat index lists are grouped by ascending neighbour rank, which is the segment
        # order all_to_all_single uses
        assert list(o_ranks) == sorted(o_ranks) and list(g_ranks) == sorted(g_ranks)
        self.o_idx = self.k.index_tensor(o_idx)
        self.g_idx = self.k.index_tensor(g_idx)
        if reverse:  # ghosts -> owners
            self.send_counts, self.recv_counts = o_counts, g_counts
        else:  # owners -> ghosts
            self.send_counts, self.recv_counts = g_counts, o_counts
        # send/recv buffers are owned by the closure and allocated once (cuda/scatterer.py:133-138)
        self.send = self.k.buffer(sum(self.send_counts))
        self.recv = self.k.buffer(sum(self.recv_counts))
        # ghosts numbered owner by owner (BoxMesh does; a dolfinx mesh may not): the ghost block of
        # the vector IS the message buffer of the owners' side -- receive into it (forward) / send
        # from it (reverse) directly, no unpack_fwd / pack_rev launch
        self.nghost = int(len(o_idx))
        self.direct = self.nghost > 0 and bool(np.array_equal(np.asarray(o_idx), np.arange(self.nghost)))
        self.active = (sum(self.send_counts) + sum(self.recv_counts)) > 0 or comm.size > 1

    def begin(self, buffer):
        """Pack and post the exchange; returns a handle for ``end``."""
        send, recv = self.send, self.recv
        if self.reverse:
            if self.direct:
                send = buffer[self.N : self.N + self.nghost]
            else:
                self.k.pack_rev(buffer, self.send, self.o_idx, self.N)
        else:
            self.k.pack_fwd(buffer, self.send, self.g_idx)
            if self.direct:
                recv = buffer[self.N : self.N + self.nghost]
        if self.comm.size == 1:
            return None
        return self.comm.alltoallv(send, self.send_counts, recv, self.recv_counts, async_op=True)

    def end(self, buffer, work):
        """Complete the exchange (stream-ordered for RCCL) and unpack."""
        if work is not None:
            work.wait()
        if self.reverse:
            self.k.unpack_rev(self.recv, buffer, self.g_idx)
        elif not self.direct:
            self.k.unpack_fwd(self.recv, buffer, self.o_idx, self.N)

    def __call__(self, buffer):
        self.end(buffer, self.begin(buffer))


def begin_all(pairs, arm_join=None):
    """Post the exchanges of several ``(scatter closure, vector)`` pairs; returns ``[(closure, vector, handle)]``
    for ``closure.end(vector, handle)``.  Native closures of one communicator and one direction go out as ONE
    RCCL group (``fus_halo_*_begin_group``): the RK4 stage's two forward scatters cost one launch, not two.

    ``arm_join`` (a ``NativeComm``): these exchanges are the LAST work of an apply on the communicator's stream; the receive
    kernel of the last one publishes the join flag (``fus_comm_arm_join``).  The library hands the join to the last halo of
    ONE begin call, so it is armed right before the call that posts the last exchange -- the group, or, when the pairs go out
    one by one (more than 8, mixed closures), the last ``begin`` (armed before the first of several calls, the
    first receive kernel would publish the flag while the later exchanges are still queued)."""
    import ctypes as C

    pairs = list(pairs)
    native = [sc for sc, _ in pairs if isinstance(sc, _NativeScatter)]
    if len(pairs) > 1 and len(native) == len(pairs) and len({id(sc.comm) for sc in native}) == 1 \
            and len({sc.reverse for sc in native}) == 1 and len(pairs) <= 8:
        for sc, vec in pairs:
            _lib.require_device_tensor(vec, sc.dtype, "buffer")
            sc._ensure_connected()
        halos = (C.c_void_p * len(pairs))(*[sc.handle for sc, _ in pairs])
        bufs = (C.c_void_p * len(pairs))(*[vec.data_ptr() for _, vec in pairs])
        fn = getattr(_lib.load(), "fus_halo_reverse_begin_group" if native[0].reverse else "fus_halo_forward_begin_group")
        if arm_join is not None and native[0].comm is arm_join:
            arm_join.arm_join()
        _lib.check(fn(halos, bufs, len(pairs), _lib.stream_ptr()), "fus_halo_begin_group", native[0].comm.handle)
        return [(sc, vec, None) for sc, vec in pairs]
    out = []
    for i, (sc, vec) in enumerate(pairs):
        if arm_join is not None and i == len(pairs) - 1 and isinstance(sc, _NativeScatter) and sc.comm is arm_join:
            sc._ensure_connected()  # nothing that can fail between arming and the begin that takes the join
            arm_join.arm_join()
        out.append((sc, vec, sc.begin(vec)))
    return out


def scatter_reverse(comm, owners_data, ghosts_data, N, float_type, kernels=None):
    """cuda/scatterer.py:104-188 / numba-cpu/scatterer.py:78-141.  ``comm``: see ``as_comm`` (an ``MPI.Comm`` is accepted)."""
    comm = as_comm(comm)
    if isinstance(comm, NativeComm):
        return _NativeScatter(comm, owners_data, ghosts_data, N, float_type, True)
    return _Scatter(comm, owners_data, ghosts_data, N, float_type, True, kernels)


def scatter_forward(comm, owners_data, ghosts_data, N, float_type, kernels=None):
    """cuda/scatterer.py:191-277 / numba-cpu/scatterer.py:144-207.  ``comm``: see ``as_comm`` (an ``MPI.Comm`` is accepted)."""
    comm = as_comm(comm)
    if isinstance(comm, NativeComm):
        return _NativeScatter(comm, owners_data, ghosts_data, N, float_type, False)
    return _Scatter(comm, owners_data, ghosts_data, N, float_type, False, kernels)


class HaloApply:
    """Distributed operator apply  y += K x  on a partitioned mesh, with the
    halo exchange overlapped with interior-cell work (one process per GPU):

        fwd.begin(x) | apply(lead slice) apply(interior half 1) | fwd.end(x)
        apply(boundary cells)                               # the only cells touching ghost dofs
        rev.begin(y) | apply(lead slice) apply(interior half 2) | rev.end(y)

    This is the reference's per-stage sequence scatter_fwd -> stiffness ->
    scatter_rev (cuda/demo_linear_box.py:537-553) with its host syncs removed
    and the cells split so that both exchanges hide behind interior work.
    ``mesh.dofmap`` must list the ghost-touching cells first
    (``mesh.num_boundary_cells``), as ``BoxMesh`` does.

    Lead slices (``lead_cells``; profiles/r02z_overlap_*.log): the operator kernels hold every vector
    register of every CU (4 workgroups x 128 VGPRs per SIMD lane), and RCCL's send/recv kernel needs 264 per
    wave, so a kernel posted next to a chip-filling launch is not scheduled until that launch drains -- the
    exchange would run AFTER the interior cells, not under them.  Each overlapped region therefore starts
    with a launch too small to fill the chip (about 0.6 of the resident-workgroup slots): the exchange kernels
    become resident next to it and stay resident under the large launch that follows.
    """

    def __init__(self, mesh, op, comm, float_type, overlap=True, kernels=None, apply_fn=None, plan=None, lead_cells="auto",
                 schedule="auto"):
        from .utils import compute_scatterer_data_flat

        self.mesh = mesh
        # the sub-launches of one apply run next to each other and next to the reverse exchange's receive kernel, all adding
        # into the same y: an operator with an atomic-free default (mass_operator) hands over its atomic twin
        self.op = getattr(op, "atomic", op)
        # ... unless the operator can be applied ROW BY ROW (``mass_operator``: y[d] depends on x[d] alone): the apply is then
        # split by dof instead of by cell -- set A: owned dofs the reverse exchange does not add into (one launch next to the
        # exchanges, all their entries, whatever cells those are in); set B: ghost dofs (their x arrives with the forward
        # exchange) and owned dofs that are ghosted elsewhere (the reverse receive adds into them), launched between the two
        # exchanges on the communicator's stream.  No launch and no receive kernel ever adds into the same y[d] concurrently,
        # so the atomic-free kernel stays valid at N > 1 (csrc/mass_gather.hpp).
        self._op_rows = op if hasattr(op, "apply_rows") and hasattr(op, "rows_available") else None
        self._row_sets = {}  # (dofmap pointer, vector length) -> uint8 marks or None (not available: cell split + atomics)
        comm = as_comm(comm)
        self.comm = comm
        # plan = (owners_data, ghosts_data) already computed (e.g. by the reference-style
        # compute_scatterer_data of a driver); default: exchange the indices over ``comm`` now
        od, gd = plan if plan is not None else compute_scatterer_data_flat(mesh.index_map, comm)
        self.owners_data, self.ghosts_data = od, gd
        self.fwd = scatter_forward(comm, od, gd, mesh.nlocal, float_type, kernels)
        self.rev = scatter_reverse(comm, od, gd, mesh.nlocal, float_type, kernels)
        self.overlap = overlap
        nb, nc = mesh.num_boundary_cells, mesh.ncells
        mid = nb + (nc - nb) // 2
        has_neighbours = (len(self.neighbour_ranks()) > 0)
        import os as _os

        # "concurrent" (csrc/halo_ipc.hpp transports): ONE launch over all interior cells on the caller's stream; the
        # boundary cells and both exchanges on a high-priority side stream next to it (profiles/r03a_overlap_local.log:
        # a same-shape kernel on a second stream takes the workgroup slots the interior launch frees, so the apply is
        # not cut into three launches).  "split": interior half | boundary | interior half on one stream, the form
        # RCCL's kernel needs (with lead slices).
        if schedule == "auto":
            schedule = _os.environ.get("FUS_HALO_SCHEDULE", "auto")
        if schedule == "auto":
            small_kernels = isinstance(comm, NativeComm) and comm.backend in ("peer", "peer-local", "local")
            schedule = "concurrent" if (small_kernels and apply_fn is None) else "split"
        if schedule not in ("concurrent", "split"):
            raise ValueError(f"schedule must be 'auto', 'concurrent' or 'split', got {schedule!r}")
        self.schedule_kind = schedule if overlap else "sequential"
        if schedule == "concurrent":
            lead_cells = 0  # lead slices exist for RCCL's kernel only
        if lead_cells == "auto":
            lead = self._auto_lead_cells(mesh)
            if (nc - nb) < 8 * lead:  # mesh too small to slice
                lead = 0
        else:
            lead = max(0, min(int(lead_cells or 0), mid - nb, nc - mid))
        if not overlap or not has_neighbours:  # nothing to hide
            lead = 0
        self.lead_cells = lead
        self.ranges = {"boundary": (0, nb), "lead1": (nb, nb + lead), "interior1": (nb + lead, mid),
                       "lead2": (mid, mid + lead), "interior2": (mid + lead, nc), "interior": (nb, nc)}
        self._events = None
        self._views_cache = {}
        self._apply_fn = apply_fn  # tests: CPU stand-in for the operator
        import os

        # FUS_HALO_SIDE_STREAM=1: run pack -> exchange -> unpack on a high-priority side stream.
        # Off by default: in a 1-rank world it costs +11 us wall and +57 us host issue time per
        # apply (event traffic) for ~20 us of pack/unpack launches it could take off the main
        # stream on a real partition (profiles/r01f_host_overhead.log); to be re-measured on 8 GPUs.
        self.side_stream = os.environ.get("FUS_HALO_SIDE_STREAM", "0") == "1"
        self._lib_sync = os.environ.get("FUS_HALO_EVENT_SYNC", "0") != "1"  # 1: fork / join the side stream with events
        self._fold_sync = os.environ.get("FUS_HALO_FOLD_SYNC", "1") != "0"  # 0: fork / join as kernels of their own (A/B runs)
        self._attach_sync = os.environ.get("FUS_HALO_ATTACH_SYNC", "1") != "0"  # 0: the fork's signal as a kernel of its own
        self._warm = set()  # interior cell ranges (by the pointers of their per-cell arrays) that have been applied once
        self._hs = None

    @staticmethod
    def concurrent_safe(op):
        """The operator to use for anything that adds into the apply's output vector WHILE the apply runs -- the cell
        operator's sub-launches (done by ``HaloApply`` itself) and whatever a caller's ``boundary_terms`` launches: an operator
        with an atomic-free default (``mass_operator``: plain load + store per dof) hands over its float-atomic twin, every
        other operator is returned as it is."""
        return getattr(op, "atomic", op)

    def row_split(self, dofmap, ndofs):
        """The uint8 marks of the row split for ``dofmap`` on vectors of ``ndofs`` entries (0: set A, 1: set B), or ``None`` when
        the operator is not row-wise or its atomic-free kernel is not available for this dofmap (decided once per dofmap)."""
        if self._op_rows is None or self._apply_fn is not None or not isinstance(dofmap, torch.Tensor) or not dofmap.is_cuda:
            return None
        key = (dofmap.data_ptr(), int(ndofs), dofmap._version)
        if key not in self._row_sets:
            marks = torch.zeros(int(ndofs), dtype=torch.uint8, device=dofmap.device)
            marks[self.mesh.nlocal:] = 1
            ghosted = torch.from_numpy(np.ascontiguousarray(to_flat(self.ghosts_data)[0], dtype=np.int64)).to(dofmap.device)
            if ghosted.numel():
                marks[ghosted] = 1
            self._row_sets[key] = marks if self._op_rows.rows_available(dofmap, int(ndofs), marks) else None
        return self._row_sets[key]

    def neighbour_ranks(self):
        """Neighbour ranks of this rank (either direction)."""
        od, gd = self.owners_data, self.ghosts_data
        return [int(r) for r in list(np.asarray(od[-1]).reshape(-1)) + list(np.asarray(gd[-1]).reshape(-1))]

    @staticmethod
    def _auto_lead_cells(mesh):
        """Cells of a launch that fills about 0.6 of the chip's resident-workgroup slots (4 workgroups per
        CU up to P = 5, 3 above: tools/resource_usage.py)."""
        try:
            nd = int(mesh.dofmap.shape[1])
            P = int(round(nd ** (1.0 / 3.0))) - 1
            ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
            epb = int(_lib.load().fus_plan_entities_per_batch(nd))
        except Exception:
            return 0
        if epb < 1 or (P + 1) ** 3 != nd:
            return 0
        return int(0.6 * ncu * (4 if P <= 5 else 3)) * epb

    def _halo_stream(self):
        if self._hs is None:
            lib_stream = self.comm.stream() if isinstance(self.comm, NativeComm) and self.comm.backend != "rccl" else None
            # high priority: small kernels between big ones
            self._hs = lib_stream if lib_stream is not None else torch.cuda.Stream(priority=-1)
        return self._hs

    @staticmethod
    def _plans_ready(views):
        """The batch plans of the dofmaps among ``views`` (int32 [cells, dofs per cell]) are cached: the launch does no set-up."""
        from . import operators as ops

        dms = [t for t in views if t.dtype == torch.int32 and t.dim() == 2 and t.shape[1] > 8]
        return len(dms) > 0 and all(ops._PLANS.has(t) for t in dms)

    def _views(self, name, percell):
        key = (name,) + tuple(t.data_ptr() for t in percell)
        v = self._views_cache.get(key)
        if v is None:
            a, b = self.ranges[name]
            v = tuple(t[a:b] for t in percell)  # views are kept so the operators' plan cache hits
            self._views_cache[key] = v
        return v

    def run(self, cell_fn, percell, forward, reverse, boundary_terms=None):
        """Generic overlapped stage.  ``cell_fn(*views)`` applies the cell operator(s) to one
        contiguous sub-range of cells, given views of the per-cell tensors ``percell`` (constants,
        G, detJ, dofmap, ...).  ``forward`` / ``reverse``: lists of ``(scatter closure, vector)``.
        ``boundary_terms()`` adds boundary-facet contributions; it runs after every forward scatter
        has landed and before the reverse scatters are posted (facet dofs can be ghosts).  In the concurrent schedule it
        runs NEXT TO the interior launch: whatever it launches must add with atomics (``concurrent_safe(op)``)."""
        for _ in self.schedule(cell_fn, percell, forward, reverse, boundary_terms):
            pass

    def schedule(self, cell_fn, percell, forward, reverse, boundary_terms=None, by_rows=False):
        """``by_rows``: ``cell_fn(which)`` applies the operator to row set ``which`` (0: A, 1: B; see ``row_split``) over ALL cells;
        ``percell`` is unused.

        The stage of ``run`` as a generator that yields each time this rank has POSTED a set of
        exchanges (``"forward"`` / ``"reverse"``) and is about to do work that does not depend on
        them.  One rank per process never needs the yields (``run`` just exhausts them); a host
        that drives several ranks from one thread (``NativeComm(local=...)``) advances all ranks'
        generators in lock step, so every rank has posted before any rank completes."""
        def part(name):
            if by_rows:  # set A where the schedule has its (first) interior launch, set B in place of the boundary cells
                if name in ("interior", "interior1"):
                    cell_fn(0)
                elif name == "boundary":
                    cell_fn(1)
                return
            a, b = self.ranges[name]
            if b > a:
                cell_fn(*self._views(name, percell))

        if not self.overlap:
            fw = begin_all(forward)
            yield "forward"
            for sc, vec, wk in fw:
                sc.end(vec, wk)
            for name in ("boundary", "lead1", "interior1", "lead2", "interior2"):
                part(name)
            if boundary_terms is not None:
                boundary_terms()
            rv = begin_all(reverse)
            yield "reverse"
            for sc, vec, wk in rv:
                sc.end(vec, wk)
            return
        if by_rows and boundary_terms is not None:
            raise ValueError("a row-split apply takes no boundary_terms (they would add into rows of set A next to its launch)")
        if self.schedule_kind == "concurrent" and (by_rows or (len(percell) > 0 and percell[0].is_cuda)):
            # main stream: ONE launch over all interior cells.  Side stream (the communicator's own high-priority stream
            # where the library has one: send, receive and the boundary kernels then follow each other in stream order,
            # no event edge between them): forward exchange -> boundary cells -> reverse exchange.
            main, side = torch.cuda.current_stream(), self._halo_stream()
            # fork / join: the library's event-free pair where the side stream is the communicator's (2.4 us on the
            # caller's stream instead of 7 + 3.5 for event record / wait next to chip-filling launches); events otherwise
            lib_sync = side is getattr(self.comm, "_stream", None) and self._lib_sync
            if self._events is None:
                self._events = (torch.cuda.Event(), torch.cuda.Event())
            ev_start, ev_side = self._events
            # PEER: fork and join are folded into the first send / last receive kernel of the chain (two kernels fewer)
            fold = lib_sync and getattr(self.comm, "transport", None) == "peer" and self._fold_sync
            # ... and the fork's signal rides on the interior launch itself (planned kernels publish the fork flag when their
            # first workgroup starts).  The flag then depends on the HOST reaching that launch, so nothing that can
            # synchronise the device may sit between the fork and the launch: the launch follows the fork at once (the
            # forward exchange is posted after it; it still runs under the interior kernel), and only once this cell
            # range has been applied before -- its batch plan exists, no first-use set-up inside the launch call.
            a_, b_ = self.ranges["interior"]
            warm_key = ("interior",) + (() if by_rows else tuple(t.data_ptr() for t in percell))
            attach = (lib_sync and self._attach_sync and self._apply_fn is None and not by_rows and b_ > a_ and warm_key in self._warm
                      and self._plans_ready(self._views("interior", percell)))
            if lib_sync:
                self.comm.fork(lazy=fold and len(forward) > 0, attach=attach)
            else:
                ev_start.record(main)
                side.wait_event(ev_start)
            part("interior")
            if attach:
                self.comm.fork_flush()  # a launch that could not carry the signal (plan-free kernel): a signal kernel after all
            self._warm.add(warm_key)
            with torch.cuda.stream(side):
                fw = begin_all(forward)
            yield "forward"
            with torch.cuda.stream(side):
                for sc, vec, wk in fw:
                    sc.end(vec, wk)
                part("boundary")
                if boundary_terms is not None:
                    boundary_terms()
                rv = begin_all(reverse, arm_join=self.comm if (fold and len(reverse) > 0) else None)
            yield "reverse"
            with torch.cuda.stream(side):
                for sc, vec, wk in rv:
                    sc.end(vec, wk)
                if not lib_sync:
                    ev_side.record(side)
            if lib_sync:
                self.comm.join()
            else:
                main.wait_event(ev_side)
            return
        on_gpu = self.side_stream and len(forward) > 0 and forward[0][1].is_cuda and not isinstance(self.comm, NativeComm)
        if not on_gpu:
            # NativeComm: begin() hands pack -> exchange -> unpack to the library's own high-priority
            # stream and returns; end() only makes this stream wait for it.  TorchComm: pack is
            # enqueued here, the collective on RCCL's stream, the unpack in end().
            fw = begin_all(forward)
            yield "forward"
            part("lead1")
            part("interior1")
            for sc, vec, wk in fw:
                sc.end(vec, wk)
            part("boundary")
            if boundary_terms is not None:
                boundary_terms()
            rv = begin_all(reverse)
            yield "reverse"
            part("lead2")
            part("interior2")
            for sc, vec, wk in rv:
                sc.end(vec, wk)
            return
        # TorchComm on the GPU with FUS_HALO_SIDE_STREAM=1: the whole exchange chain (pack ->
        # all-to-all-v -> unpack) runs on a high-priority side stream; the two streams meet only
        # where data demands it:
        #   side waits for main   : vectors ready to pack (start; after the boundary cells)
        #   main waits for side   : ghosts refreshed (before the boundary cells); sums landed (end)
        # Hazards: interior cells never touch ghost entries (unpack_fwd writes, pack_rev reads them);
        # unpack_rev and the interior kernel both ADD into owned entries of y with atomics.
        main = torch.cuda.current_stream()
        hs = self._halo_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        hs.wait_event(ev)
        with torch.cuda.stream(hs):
            fw = [(sc, vec, sc.begin(vec)) for sc, vec in forward]
            for sc, vec, wk in fw:
                sc.end(vec, wk)
            ev_fwd = torch.cuda.Event()
            ev_fwd.record(hs)
        part("lead1")
        part("interior1")
        main.wait_event(ev_fwd)
        part("boundary")
        if boundary_terms is not None:
            boundary_terms()
        ev_b = torch.cuda.Event()
        ev_b.record(main)
        hs.wait_event(ev_b)
        with torch.cuda.stream(hs):
            rv = [(sc, vec, sc.begin(vec)) for sc, vec in reverse]
            for sc, vec, wk in rv:
                sc.end(vec, wk)
            ev_rev = torch.cuda.Event()
            ev_rev.record(hs)
        part("lead2")
        part("interior2")
        main.wait_event(ev_rev)

    def apply(self, x, cell_constants, y, G, dofmap, extra_forward=(), boundary_terms=None):
        """y += K x on the partitioned mesh (``extra_forward``: further ``(scatter_forward closure,
        vector)`` pairs to refresh alongside x, e.g. v_n of the RK stage)."""
        for _ in self.apply_schedule(x, cell_constants, y, G, dofmap, extra_forward, boundary_terms):
            pass

    def apply_schedule(self, x, cell_constants, y, G, dofmap, extra_forward=(), boundary_terms=None):
        """``apply`` as a generator (see ``schedule``)."""
        marks = self.row_split(dofmap, min(x.numel(), y.numel())) if boundary_terms is None else None
        if marks is not None:
            op = self._op_rows
            return self.schedule(lambda which: op.apply_rows(x, cell_constants, y, G, dofmap, marks, which), (),
                                 [(self.fwd, x)] + list(extra_forward), [(self.rev, y)], None, by_rows=True)
        fn = self._apply_fn if self._apply_fn is not None else self.op
        return self.schedule(lambda c_, G_, d_: fn(x, c_, y, G_, d_), (cell_constants, G, dofmap),
                             [(self.fwd, x)] + list(extra_forward), [(self.rev, y)], boundary_terms)

    def prepare(self, x, cell_constants, G, dofmap):
        """Set-up, not an apply: build the batch plans of the three cell sub-ranges and bring the
        communicator up (RCCL creates its channels on first use) with two no-effect exchanges --
        a forward scatter of x (ghosts receive their owners' values) and a reverse scatter of a
        zero vector."""
        if self.row_split(dofmap, x.numel()) is None:  # (a row-split apply has built its three transposed plans by now)
            for name in self._launch_names():
                a, b = self.ranges[name]
                if b > a and self.op is not None and hasattr(self.op, "prepare"):
                    self.op.prepare(self._views(name, (cell_constants, G, dofmap))[2])
        # the device-side waits of the PEER transport are bounded: line the ranks up on the host before the first
        # exchange, so that a rank whose set-up took longer is not mistaken for a dead one
        if hasattr(self.comm, "barrier") and getattr(self.comm, "_world_id", None) is None:
            self.comm.barrier()
        self.fwd(x)
        self.rev(x.new_zeros(x.shape))

    def apply_local_only(self, x, cell_constants, y, G, dofmap):
        """The three kernel launches without any exchange (bench: kernel time at N > 1)."""
        marks = self.row_split(dofmap, min(x.numel(), y.numel()))
        if marks is not None:
            for which in (1, 0):
                self._op_rows.apply_rows(x, cell_constants, y, G, dofmap, marks, which)
            return
        fn = self._apply_fn if self._apply_fn is not None else self.op
        for name in self._launch_names():
            a, b = self.ranges[name]
            if b > a:
                c_, G_, d_ = self._views(name, (cell_constants, G, dofmap))
                fn(x, c_, y, G_, d_)

    def apply_no_exchange(self, x, cell_constants, y, G, dofmap):
        """The launches of ``apply`` in its own schedule (streams, events, sub-ranges) with NO exchange: what cutting
        the apply into sub-launches costs by itself (bench.py: ``halo_split_cost_ms``).  Not an apply."""
        marks = self.row_split(dofmap, min(x.numel(), y.numel()))
        if marks is not None:
            op = self._op_rows
            for _ in self.schedule(lambda which: op.apply_rows(x, cell_constants, y, G, dofmap, marks, which), (), [], [], None, by_rows=True):
                pass
            return
        fn = self._apply_fn if self._apply_fn is not None else self.op
        for _ in self.schedule(lambda c_, G_, d_: fn(x, c_, y, G_, d_), (cell_constants, G, dofmap), [], []):
            pass

    def _launch_names(self):
        """The cell sub-ranges this schedule launches, in issue order."""
        if self.schedule_kind == "concurrent":
            return ("boundary", "interior")
        return ("lead1", "interior1", "boundary", "lead2", "interior2")

    def health(self):
        """Failed device-side waits (time-outs, poisoned flags of failed neighbours) of EVERY closure of this communicator
        and of its fork / join kernels (0 = every exchange so far delivered); synchronises the exchange streams."""
        if isinstance(self.comm, NativeComm) and self.comm.handle:
            return self.comm.health()
        return 0

    def check_health(self, what="halo exchange"):
        """Raise if any exchange of this communicator failed on this rank (a device-side wait gave up, or a neighbour's
        failure reached this rank through a poisoned flag).  The reference would block in MPI Waitall
        (cuda/scatterer.py:175); here every wait is bounded, so a failure must be made loud instead."""
        n = self.health()
        if n:
            d = self.comm.health_detail() if hasattr(self.comm, "health_detail") else {}
            raise _lib.FusGpuError(
                f"{what}: {n} device-side wait(s) of the halo exchange failed on rank {self.comm.rank} ({d.get('timeouts', '?')} time-out(s) after "
                f"FUS_IPC_SPIN_SECONDS, {d.get('poisoned', '?')} flag(s) poisoned by a neighbour's failed exchange, {d.get('sync_timeouts', '?')} "
                "fork / join time-out(s)): ghost data is stale, the result is INVALID")
