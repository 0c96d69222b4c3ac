"""
Halo index plumbing (host, integer, run once).

``compute_scatterer_data(index_map, comm)`` keeps the call surface and the
return format of the reference cuda/utils.py:8-78:

    owners_data = [owners_idx (list of int64 arrays), owners_size, unique_owners]
    ghosts_data = [ghosts_idx (list of int64 arrays), ghosts_size, unique_ghosts]

  * ``owners_*``: my ghost dofs grouped by the rank that owns them;
    ``owners_idx[i]`` are positions inside my ghost block (0-based, add
    ``nlocal`` for the vector index) of the ghosts owned by ``unique_owners[i]``
    (cuda/utils.py:23-37).
  * ``ghosts_*``: my owned dofs that are ghosts on other ranks;
    ``ghosts_idx[i]`` are my local indices that ``unique_ghosts[i]`` ghosts, in
    the order that rank packs them (cuda/utils.py:40-73: the ghosting rank sends
    the global indices, the owner subtracts ``local_range[0]``).

The numba-cpu drivers use the same data flattened with offsets
(numba-cpu/test_operators.py:196-225): ``[idx_flat, size, offsets, ranks]``;
``to_flat`` / ``to_lists`` convert between the two.

The reference does the index exchange with ``MPI.COMM_WORLD.Isend/Irecv``
(cuda/utils.py:54-71) and finds the ghosting ranks with an O(ranks x nlocal)
Python loop (:43-47); here the exchange goes through the package's comm object
(torch.distributed) and the loop is a ``bincount``.
"""

from __future__ import annotations

import numpy as np


def _owners_side(index_map, stable=True):
    owners = np.asarray(index_map.owners)
    unique_owners, owners_size = np.unique(owners, return_counts=True)
    # stable: ghosts keep their relative order inside one owner's group, so ghosts numbered owner by
    # owner give the identity list (direct mode of the exchange).  stable=False is the reference's
    # ``np.argsort(owners)`` (cuda/utils.py:28, numpy's default introsort): same groups, possibly
    # another order inside a group -- any order is a valid plan as both sides of an exchange share it
    order = (np.argsort(owners, kind="stable") if stable else np.argsort(owners)).astype(np.int64)
    offsets = np.concatenate(([0], np.cumsum(owners_size))).astype(np.int64)
    return unique_owners.astype(np.int32), owners_size.astype(np.int64), offsets, order


def _ghosting_ranks(index_map):
    dest = index_map.index_to_dest_ranks()
    arr = np.asarray(dest.array)
    if arr.size == 0:
        return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int64)
    unique_ghosts, ghosts_size = np.unique(arr, return_counts=True)
    return unique_ghosts.astype(np.int32), ghosts_size.astype(np.int64)


def compute_scatterer_data_all(index_maps, stable=True):
    """All ranks at once, in one process (tests / single-process simulation).

    Returns ``(owners_data_all, ghosts_data_all)`` in the flat 4-element format
    ``[idx_flat, size, offsets, ranks]`` per rank.
    """
    R = len(index_maps)
    owners_all, ghosts_all = [], []
    sent = {}
    for r, im in enumerate(index_maps):
        uo, osz, ooff, order = _owners_side(im, stable)
        owners_all.append([order, osz, ooff, uo])
        gl = np.asarray(im.ghosts)[order]
        for i, o in enumerate(uo):
            sent[(r, int(o))] = gl[ooff[i] : ooff[i + 1]]
    for r, im in enumerate(index_maps):
        ug, gsz = _ghosting_ranks(im)
        goff = np.concatenate(([0], np.cumsum(gsz))).astype(np.int64)
        idx = np.empty(int(goff[-1]), dtype=np.int64)
        for i, g in enumerate(ug):
            recv = sent[(int(g), r)]
            assert recv.size == gsz[i], "halo plan mismatch between ghosting rank and owner"
            idx[goff[i] : goff[i + 1]] = recv - im.local_range[0]
        ghosts_all.append([idx, gsz, goff, ug])
    return owners_all, ghosts_all


def compute_scatterer_data_flat(index_map, comm=None, stable=True):
    """One rank's halo plan, flat format; the index exchange uses ``comm``
    (``.rank``, ``.size``, ``.alltoallv_int64(send, send_counts, recv_counts)``: a package communicator, or an
    ``mpi4py.MPI.Comm`` as the reference uses -- wrapped here)."""
    uo, osz, ooff, order = _owners_side(index_map, stable)
    ug, gsz = _ghosting_ranks(index_map)
    goff = np.concatenate(([0], np.cumsum(gsz))).astype(np.int64)
    from . import mpi_bootstrap

    if comm is None:
        # the reference hard-codes MPI.COMM_WORLD (cuda/utils.py:62,68): the default torch.distributed group if one is up,
        # else MPI.COMM_WORLD itself when mpi4py is importable (a driver started with mpirun)
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            if dist.get_world_size() > 1:
                from .scatterer import TorchComm

                comm = TorchComm()
        else:
            comm = mpi_bootstrap.world_if_available()
    if comm is not None and mpi_bootstrap.is_mpi_comm(comm):  # a raw MPI communicator: only its collectives are needed here
        comm = mpi_bootstrap.MpiBootstrap(comm)
    size = 1 if comm is None else comm.size
    if size == 1:
        if uo.size or ug.size:
            raise ValueError("index_map has ghosts but no communicator was given")
        return [order, osz, ooff, uo], [np.zeros(0, dtype=np.int64), gsz, goff, ug]
    send_counts = np.zeros(size, dtype=np.int64)
    recv_counts = np.zeros(size, dtype=np.int64)
    send_counts[uo] = osz
    recv_counts[ug] = gsz
    send = np.asarray(index_map.ghosts, dtype=np.int64)[order]  # grouped by owner rank, ascending
    recv = comm.alltoallv_int64(send, send_counts, recv_counts)  # grouped by source rank, ascending
    idx = recv - index_map.local_range[0]
    if idx.size and (idx.min() < 0 or idx.max() >= index_map.size_local):
        raise ValueError("received a ghost index outside this rank's owned range")
    return [order, osz, ooff, uo], [idx, gsz, goff, ug]


def to_lists(data_flat):
    """flat ``[idx, size, offsets, ranks]`` -> cuda-style ``[list_of_idx, size, ranks]``."""
    idx, size, off, ranks = data_flat
    return [[np.ascontiguousarray(idx[off[i] : off[i + 1]]) for i in range(len(ranks))], size, ranks]


def to_flat(data):
    """Accept either format, return the flat one."""
    if len(data) == 4:
        idx, size, off, ranks = data
        return [np.asarray(idx, dtype=np.int64), np.asarray(size, dtype=np.int64), np.asarray(off, dtype=np.int64), np.asarray(ranks, dtype=np.int32)]
    idx_list, size, ranks = data
    size = np.asarray(size, dtype=np.int64)
    off = np.concatenate(([0], np.cumsum(size))).astype(np.int64)
    conv = []
    for a in idx_list:
        if hasattr(a, "detach"):  # device tensor handed over by a cuda-style driver
            a = a.detach().cpu().numpy()
        conv.append(np.asarray(a, dtype=np.int64))
    idx = np.concatenate(conv) if conv else np.zeros(0, dtype=np.int64)
    return [idx, size, off, np.asarray(ranks, dtype=np.int32)]


def compute_scatterer_data(index_map, comm=None, stable=True):
    """cuda/utils.py:8-78 call surface (3-element list format)."""
    od, gd = compute_scatterer_data_flat(index_map, comm, stable)
    return to_lists(od), to_lists(gd)


def boundary_first_cell_order(dofmap, nlocal):
    """Permutation that lists the cells touching a ghost dof (index >= nlocal) first, keeping the
    relative order inside both groups, and the number of such cells.  ``HaloApply`` needs
    ``dofmap[perm]``, ``G[perm]``, ``detJ[perm]`` and ``cell_constants[perm]`` in this order so that
    boundary / interior cell sets are contiguous sub-ranges; ``BoxMesh`` already provides it, a
    dolfinx-built mesh applies this once at set-up."""
    dm = np.asarray(dofmap)
    touches = (dm >= nlocal).any(axis=1)
    perm = np.concatenate((np.nonzero(touches)[0], np.nonzero(~touches)[0]))
    return perm, int(touches.sum())


def config4_self_plan(n1, permuted=False, seed=0):
    """Halo plan of ONE rank of BASELINE config 4 (2x2x2 blocks of 54^3 P = 4 cells) that is its own neighbour: the three low
    faces of its n1^3 lexicographic block (one contiguous plane, one plane of runs of n1, one plane of stride n1), three
    edges and the corner -- 3 x 47 089 + 3 x 217 + 1 elements, 1.14 MB per direction.  Returns (owners_data, ghosts_data, N)."""
    ng = 3 * n1 * n1 + 3 * n1 + 1
    N = n1**3 - ng
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(n1), np.arange(n1), indexing="ij")
    lex = lambda i, j, k: ((i * n1 + j) * n1 + k).reshape(-1)  # noqa: E731
    z0 = np.zeros_like(ii)
    ar, zr = np.arange(n1), np.zeros(n1, dtype=np.int64)
    g_idx = np.concatenate([lex(z0, ii, jj), lex(ii, z0, jj), lex(ii, jj, z0), lex(zr, zr, ar), lex(zr, ar, zr), lex(ar, zr, zr),
                            np.array([0])]).astype(np.int64) % N
    o_idx = rng.permutation(ng).astype(np.int64) if permuted else np.arange(ng, dtype=np.int64)
    od = [o_idx, np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)]
    gd = [g_idx, np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)]
    return od, gd, N
