"""
Set-up-time cell order for batch plans whose kernels are bound by the scatter side (float-atomic requests), not by a per-cell
data stream: TWO-ROW STRIPS.

A batch of the plan (csrc/plan.hpp) is ``CPB`` consecutive cells of the plan's cell order; what a batch flushes is one global
atomic per DISTINCT dof it touches.  Ten P = 4 cells in a row touch 1 029 distinct dofs; the same ten cells as a 2 x 5 tile
touch 945 (-8 %), and the kernels that read no per-cell array worth mentioning -- in-kernel geometry (stiffness_geom.hpp,
westervelt_geom.hpp), the affine path -- run 5-8 % faster on such batches (profiles/r02u_experiment_batch_shape.log); the
general-G kernel does not (its 6 kB G slabs want consecutive cells) and keeps the row order.

Instead of forming tiles (whose size would have to divide every row length), the order INTERLEAVES two adjacent rows of cells
column by column:  a0 b0 a1 b1 a2 b2 ...  -- any ``CPB`` consecutive cells of that sequence are a connected two-row piece
(2 x 5 for CPB = 10; 4 + 3 for CPB = 7), whatever the row length, and a batch that spans the end of one strip and the start of
the next is the only irregular one (1 in ~11 at config 3).

Everything is derived from the dofmap alone (tensor-product local order ``l = i n^2 + j n + k``, numba-cpu/operators.py:71-227),
so it works for any mesh of hexahedra whose cell order has runs of face-adjacent cells (a structured box; a mesh ordered along
its extrusion / sweep direction):
  1. face adjacency from the 6 face-interior dofs of every cell (a dof strictly inside a face belongs to exactly the two cells
     that share the face; needs P >= 2);
  2. rows = maximal runs of consecutive cells (in the given order) that are face neighbours through OPPOSITE faces;
  3. rows are paired greedily: the partner of a row is an unpaired row of the same length whose k-th cell is a face neighbour
     of the row's k-th cell for every k;
  4. paired rows are emitted interleaved, unpaired rows as they are.
Host numpy, once per dofmap (config 3: 0.2 s).  The result is only a candidate: the plan cache builds the plan with it and keeps
it if its batches touch fewer distinct dofs (operators._PlanCache).
"""

from __future__ import annotations

import numpy as np


def face_interior_local_dofs(n: int):
    """Local ids of one dof strictly inside each of the 6 faces, in face order (i = 0, i = P, j = 0, j = P, k = 0, k = P);
    faces 2a and 2a + 1 are opposite."""
    P, m = n - 1, 1
    loc = lambda i, j, k: (i * n + j) * n + k  # noqa: E731
    return np.array([loc(0, m, m), loc(P, m, m), loc(m, 0, m), loc(m, P, m), loc(m, m, 0), loc(m, m, P)], dtype=np.int64)


def face_neighbours(face_dofs: np.ndarray) -> np.ndarray:
    """``face_dofs`` int[ncell, 6] (the dof inside each face) -> nbr int64[ncell, 6]: the cell across each face, -1 at the
    boundary (or where a dof is shared by more than two cells: not a conforming hexahedral mesh there)."""
    nc = face_dofs.shape[0]
    key = face_dofs.reshape(-1).astype(np.int64)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    same_next = np.zeros(ks.size, dtype=bool)
    same_next[:-1] = ks[1:] == ks[:-1]
    same_prev = np.zeros(ks.size, dtype=bool)
    same_prev[1:] = same_next[:-1]
    # exactly two entries with this dof: (first: same_next & ~same_prev & ~next's same_next)
    nn = np.zeros(ks.size, dtype=bool)
    nn[:-1] = same_next[1:]
    first = same_next & ~same_prev & ~nn
    a = order[first]
    b = order[np.nonzero(first)[0] + 1]
    nbr = np.full(nc * 6, -1, dtype=np.int64)
    nbr[a] = b // 6
    nbr[b] = a // 6
    return nbr.reshape(nc, 6)


def two_row_strip_order(dofmap_faces: np.ndarray, seq=None, min_row=2):
    """Candidate cell order (int64[ncell]) that interleaves adjacent rows, or ``None`` when the cell order has no rows to pair.

    ``dofmap_faces``: int[ncell, 6], the dofmap columns ``face_interior_local_dofs(n)``;  ``seq``: the order the rows are looked
    for in (default: natural; the plan cache passes its locality order when it made one)."""
    nc = dofmap_faces.shape[0]
    if nc < 4:
        return None
    nbr = face_neighbours(np.asarray(dofmap_faces))
    seq = np.arange(nc, dtype=np.int64) if seq is None else np.asarray(seq, dtype=np.int64)
    cur, nxt = seq[:-1], seq[1:]
    hit = nbr[cur] == nxt[:, None]  # [nc - 1, 6]: face of cur through which nxt is reached
    has_out = hit.any(axis=1)
    out_face = np.where(has_out, hit.argmax(axis=1), -1)
    back = nbr[nxt] == cur[:, None]
    in_face = np.where(back.any(axis=1), back.argmax(axis=1), -1)  # face of nxt through which cur is reached
    # the link position p -> p + 1 continues a row if the cells are neighbours and, when p itself was entered through a link,
    # leaves p through the face opposite to the one it was entered by
    link = has_out.copy()
    entered = np.full(nc, -1, dtype=np.int64)  # face of seq[p] facing seq[p - 1], if they are neighbours
    entered[1:] = in_face
    straight = (entered[:-1] < 0) | (out_face == (entered[:-1] ^ 1))
    link &= straight
    # a turn breaks the row BEFORE the turning cell's successor; the turning cell then starts no straight continuation either
    starts = np.concatenate(([0], np.nonzero(~link)[0] + 1))
    lens = np.diff(np.concatenate((starts, [nc])))
    nrows = starts.size
    if nrows > nc // max(2, min_row) or lens.max() < min_row:
        return None
    row_of = np.repeat(np.arange(nrows), lens)  # by position in seq
    pos_of_cell = np.empty(nc, dtype=np.int64)
    pos_of_cell[seq] = np.arange(nc)
    # axis of a row: the face pair its links go through (rows of one cell have none)
    first_out = np.full(nrows, -1, dtype=np.int64)
    multi = lens > 1
    first_out[multi] = out_face[starts[multi]]
    paired = np.full(nrows, -1, dtype=np.int64)
    for r in range(nrows):
        if paired[r] >= 0 or lens[r] < min_row:
            continue
        c0 = seq[starts[r]]
        axis = first_out[r] >> 1
        best = -1
        for f in range(6):
            if (f >> 1) == axis:
                continue
            q = nbr[c0, f]
            if q < 0:
                continue
            pq = pos_of_cell[q]
            rq = row_of[pq]
            if rq == r or paired[rq] >= 0 or lens[rq] != lens[r] or starts[rq] != pq:
                continue
            if best < 0 or starts[rq] < starts[best]:
                # every column must be a pair of face neighbours
                a = seq[starts[r]: starts[r] + lens[r]]
                b = seq[starts[rq]: starts[rq] + lens[rq]]
                if bool((nbr[a] == b[:, None]).any(axis=1).all()):
                    best = rq
        if best >= 0:
            paired[r], paired[best] = best, r
    if not (paired >= 0).any():
        return None
    out = np.empty(nc, dtype=np.int64)
    w = 0
    done = np.zeros(nrows, dtype=bool)
    for r in range(nrows):
        if done[r]:
            continue
        a = seq[starts[r]: starts[r] + lens[r]]
        if paired[r] >= 0:
            q = paired[r]
            b = seq[starts[q]: starts[q] + lens[q]]
            out[w: w + 2 * a.size: 2] = a
            out[w + 1: w + 2 * a.size: 2] = b
            w += 2 * a.size
            done[q] = True
        else:
            out[w: w + a.size] = a
            w += a.size
        done[r] = True
    assert w == nc
    return out
