"""
ORACLE (test infrastructure -- NOT product code).

numpy restatement of the reference's CPU algorithm for the hot path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package never does.

Each function cites the reference lines it follows (paths relative to the
reference checkout):

  mass_apply          numba-cpu/operators.py:19-68
  stiffness_apply     numba-cpu/operators.py:71-227  (+ contract/transpose,
                      numba-cpu/sum_factorisation.py:17-95)
  axpy/copy/fill/pointwise_divide   numba-cpu/operators.py:230-300
  square              cuda/operators.py:261-274
  pack/unpack_rev/unpack_fwd        numba-cpu/scatterer.py:18-75
  scatter_reverse/scatter_forward (N simulated ranks in one process)
                      numba-cpu/scatterer.py:78-207

Pinning: checked against golden vectors produced by importing the reference
itself (tests/golden/generate_golden.py -> tests/golden/*.npz), see
tests/test_oracle_golden.py.

The per-cell loop nest of the reference is batched over cells here; the
arithmetic per cell is the same sequence of contractions:
  fw_a = (D along axis a) x_cell                     operators.py:174-190
  fw   = c * Gsym(q) . fw                            operators.py:91-119,193
  y_cell = sum_a (D^T along axis a) fw_a             operators.py:205-221
  y[dofmap] += y_cell                                operators.py:224-225
"""

from __future__ import annotations

import numpy as np


def _scatter_add(y, idx, vals):
    # np.add.at semantic (repeated indices accumulate), but fast.
    y += np.bincount(idx.reshape(-1), weights=vals.reshape(-1).astype(np.float64), minlength=y.size).astype(y.dtype)


def mass_apply(x, entity_constants, y, entity_detJ, entity_dofmap):
    """y[dofmap[e, i]] += x[dofmap[e, i]] * detJ[e, i] * const[e]."""
    if entity_dofmap.shape[0] == 0:
        return
    x_ = x[entity_dofmap] * (entity_detJ * entity_constants[:, None])
    _scatter_add(y, entity_dofmap, x_)


def stiffness_cell_values(P, dphi, x, cell_constants, G, dofmap):
    """Per-cell output block ``[ncell, n^3]`` before the scatter-add."""
    n = P + 1
    D = np.asarray(dphi, dtype=x.dtype).reshape(n, n)  # D[q, i]
    nc = dofmap.shape[0]
    xe = x[dofmap].reshape(nc, n, n, n)  # [c, i, j, k]
    fw0 = np.einsum("qi,cijk->cqjk", D, xe)
    fw1 = np.einsum("qj,cijk->ciqk", D, xe)
    fw2 = np.einsum("qk,cijk->cijq", D, xe)
    Gc = G.reshape(nc, n, n, n, 6)
    c = cell_constants.reshape(nc, 1, 1, 1)
    t0 = c * (Gc[..., 0] * fw0 + Gc[..., 1] * fw1 + Gc[..., 2] * fw2)
    t1 = c * (Gc[..., 1] * fw0 + Gc[..., 3] * fw1 + Gc[..., 4] * fw2)
    t2 = c * (Gc[..., 2] * fw0 + Gc[..., 4] * fw1 + Gc[..., 5] * fw2)
    ye = (
        np.einsum("qi,cqjk->cijk", D, t0)
        + np.einsum("qj,ciqk->cijk", D, t1)
        + np.einsum("qk,cijq->cijk", D, t2)
    )
    return ye.reshape(nc, n * n * n)


def stiffness_apply(P, dphi, x, cell_constants, y, G, dofmap, chunk=16384):
    """y += K(c) x, matrix-free (numba-cpu/operators.py:121-225)."""
    nc = dofmap.shape[0]
    for c0 in range(0, nc, chunk):
        c1 = min(c0 + chunk, nc)
        ye = stiffness_cell_values(P, dphi, x, cell_constants[c0:c1], G[c0:c1], dofmap[c0:c1])
        _scatter_add(y, dofmap[c0:c1], ye)


# ---- streaming vector ops ---------------------------------------------------
def axpy(alpha, x, y, n=None):
    n = y.size if n is None else n
    y[:n] = alpha * x[:n] + y[:n]


def copy(a, b):
    b[:] = a


def fill(alpha, x):
    x[:] = alpha


def pointwise_divide(a, b, c):
    c[:] = a / b


def square(a, b):
    b[:] = a * a


# ---- halo pack / unpack -----------------------------------------------------
def pack(in_, out_, index):
    out_[: index.size] = in_[index]


def unpack_rev(in_, out_, index):
    np.add.at(out_, index, in_[: index.size])


def unpack_fwd(in_, out_, index):
    out_[index] = in_[: index.size]


def scatter_reverse_all(buffers, owners_data_all, ghosts_data_all, nlocal_all):
    """Reverse scatter (ghost partial sums -> owner, add) for all ranks of a
    simulated communicator; ``buffers[r]`` is rank r's ``[nlocal+nghost]``
    vector, modified in place.  numba-cpu/scatterer.py:78-141 with the MPI
    messages replaced by direct hand-over."""
    R = len(buffers)
    sent = {}
    for r in range(R):
        o_idx, o_size, o_off, o_ranks = owners_data_all[r]
        N = nlocal_all[r]
        send = np.empty(int(np.sum(o_size)), dtype=buffers[r].dtype)
        pack(buffers[r][N:], send, np.asarray(o_idx))
        for i, dest in enumerate(o_ranks):
            sent[(r, int(dest))] = send[o_off[i] : o_off[i + 1]].copy()
    for r in range(R):
        g_idx, g_size, g_off, g_ranks = ghosts_data_all[r]
        for i, src in enumerate(g_ranks):
            unpack_rev(sent[(int(src), r)], buffers[r], np.asarray(g_idx[g_off[i] : g_off[i + 1]]))


def scatter_forward_all(buffers, owners_data_all, ghosts_data_all, nlocal_all):
    """Forward scatter (owner -> ghost copies, overwrite); numba-cpu/scatterer.py:144-207."""
    R = len(buffers)
    sent = {}
    for r in range(R):
        g_idx, g_size, g_off, g_ranks = ghosts_data_all[r]
        send = np.empty(int(np.sum(g_size)), dtype=buffers[r].dtype)
        pack(buffers[r], send, np.asarray(g_idx))
        for i, dest in enumerate(g_ranks):
            sent[(r, int(dest))] = send[g_off[i] : g_off[i + 1]].copy()
    for r in range(R):
        o_idx, o_size, o_off, o_ranks = owners_data_all[r]
        N = nlocal_all[r]
        for i, src in enumerate(o_ranks):
            unpack_fwd(sent[(int(src), r)], buffers[r][N:], np.asarray(o_idx[o_off[i] : o_off[i + 1]]))
