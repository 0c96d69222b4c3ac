"""
Point evaluation of a degree-P GLL field on a (trilinear) hexahedral mesh: the output side of the reference's drivers.

    cuda/utils.py:117-154            compute_eval_params(mesh, points, float_type) -> (points_on_proc, cells)
    cuda/demo_linear_box.py:128-141  100 x 100 points on the z = 0 plane
    cuda/demo_linear_box.py:587-605  u_n_.eval(x_eval, cell_eval) -> rows "x, y, value"
    cuda/demo_nonlinear_bowl.py:662-680  the same, once per step over the last period

The reference leans on dolfinx for both halves (bounding-box tree + collision test for the cells, ``Function.eval`` for
the values); dolfinx exists nowhere in this pipeline, so both are done here on the plain arrays a driver holds --
``x_dofs`` (8 vertices per cell, vertex ``v = vx + 2 vy + 4 vz``), ``x_g``, ``dofmap`` (tensor-product local order) -- for
ANY mesh of trilinear hexahedra (structured or not): cells are found through a uniform hash grid over their bounding
boxes, the reference coordinates by Newton iteration on the trilinear map, the value by the tensor product of the 1-D
Lagrange bases on the GLL nodes.  Host-side numpy (output is off the hot path, as in the reference).
"""

from __future__ import annotations

import numpy as np

from .gll import gll_points_weights


def _shape(ref):
    """Trilinear shape functions [npts, 8] and their gradients [npts, 8, 3] at reference points in [0, 1]^3."""
    n = ref.shape[0]
    N = np.empty((n, 8))
    dN = np.empty((n, 8, 3))
    for v in range(8):
        b = (v & 1, (v >> 1) & 1, (v >> 2) & 1)
        f = [ref[:, a] if b[a] else 1.0 - ref[:, a] for a in range(3)]
        d = [1.0 if b[a] else -1.0 for a in range(3)]
        N[:, v] = f[0] * f[1] * f[2]
        dN[:, v, 0] = d[0] * f[1] * f[2]
        dN[:, v, 1] = f[0] * d[1] * f[2]
        dN[:, v, 2] = f[0] * f[1] * d[2]
    return N, dN


def _invert(cell_xyz, pts, iters=30, tol=1e-13):
    """Reference coordinates of ``pts[k]`` in the trilinear cell with vertices ``cell_xyz[k]`` ([K, 8, 3]): Newton from
    the centre.  Returns (xi [K, 3], converged [K])."""
    xi = np.full((pts.shape[0], 3), 0.5)
    scale = np.maximum(np.ptp(cell_xyz, axis=1).max(axis=1), 1e-300)
    ok = np.zeros(pts.shape[0], dtype=bool)
    for _ in range(iters):
        N, dN = _shape(xi)
        r = np.einsum("kv,kvd->kd", N, cell_xyz) - pts
        ok = np.linalg.norm(r, axis=1) <= tol * scale
        if ok.all():
            break
        J = np.einsum("kva,kvd->kda", dN, cell_xyz)  # d x_d / d xi_a
        det = np.linalg.det(J)
        good = np.abs(det) > 1e-300
        step = np.zeros_like(xi)
        step[good] = np.linalg.solve(J[good], r[good][..., None])[..., 0]
        xi = np.clip(xi - step, -1.0, 2.0)  # a point outside the cell wanders off; it is rejected below
    return xi, ok


class CellLocator:
    """Uniform hash grid over the cells' bounding boxes (the role of dolfinx's ``bb_tree`` + ``compute_collisions_points``
    + ``compute_colliding_cells``, cuda/utils.py:141-144)."""

    def __init__(self, x_dofs, x_g, padding=1e-12):
        self.x_dofs = np.asarray(x_dofs)
        self.x_g = np.asarray(x_g, dtype=np.float64)
        v = self.x_g[self.x_dofs]  # [nc, 8, 3]
        self.lo, self.hi = v.min(axis=1) - padding, v.max(axis=1) + padding
        nc = self.x_dofs.shape[0]
        self.origin = self.lo.min(axis=0)
        extent = np.maximum(self.hi.max(axis=0) - self.origin, 1e-300)
        mean = np.maximum((self.hi - self.lo).mean(axis=0), 1e-300)
        self.nb = np.maximum(1, np.minimum((extent / mean).astype(np.int64), 256))
        self.h = extent / self.nb
        b0 = np.clip(((self.lo - self.origin) / self.h).astype(np.int64), 0, self.nb - 1)
        b1 = np.clip(((self.hi - self.origin) / self.h).astype(np.int64), 0, self.nb - 1)
        # (bin, cell) pairs: every bin a cell's box overlaps
        span = b1 - b0 + 1
        bins, cells = [], []
        for dx in range(int(span[:, 0].max())):
            for dy in range(int(span[:, 1].max())):
                for dz in range(int(span[:, 2].max())):
                    m = (span[:, 0] > dx) & (span[:, 1] > dy) & (span[:, 2] > dz)
                    if not m.any():
                        continue
                    b = b0[m] + np.array([dx, dy, dz])
                    bins.append((b[:, 0] * self.nb[1] + b[:, 1]) * self.nb[2] + b[:, 2])
                    cells.append(np.nonzero(m)[0])
        bins = np.concatenate(bins) if bins else np.zeros(0, np.int64)
        cells = np.concatenate(cells) if cells else np.zeros(0, np.int64)
        order = np.argsort(bins, kind="stable")
        self._cells = cells[order]
        nbins = int(np.prod(self.nb))
        self._start = np.searchsorted(bins[order], np.arange(nbins + 1))
        self.ncells = nc

    def locate(self, points, tol=1e-10):
        """``points`` [npts, 3] -> (cell [npts] or -1, xi [npts, 3]); a point on a shared face goes to the lowest cell
        that contains it (dolfinx takes the first collision, cuda/utils.py:150)."""
        pts = np.asarray(points, dtype=np.float64)
        npts = pts.shape[0]
        cell = np.full(npts, -1, dtype=np.int64)
        xi_out = np.zeros((npts, 3))
        inside_box = np.all((pts >= self.origin - 1e-12) & (pts <= self.origin + self.h * self.nb + 1e-12), axis=1)
        b = np.clip(((pts - self.origin) / self.h).astype(np.int64), 0, self.nb - 1)
        flat = (b[:, 0] * self.nb[1] + b[:, 1]) * self.nb[2] + b[:, 2]
        cnt = self._start[flat + 1] - self._start[flat]
        for k in range(int(cnt.max()) if npts else 0):  # k-th candidate of every point still unresolved
            todo = np.nonzero((cell < 0) & inside_box & (cnt > k))[0]
            if todo.size == 0:
                break
            cand = self._cells[self._start[flat[todo]] + k]
            p = pts[todo]
            inbb = np.all((p >= self.lo[cand]) & (p <= self.hi[cand]), axis=1)
            todo, cand, p = todo[inbb], cand[inbb], p[inbb]
            if todo.size == 0:
                continue
            xi, ok = _invert(self.x_g[self.x_dofs[cand]], p)
            hit = ok & np.all((xi >= -tol) & (xi <= 1.0 + tol), axis=1)
            cell[todo[hit]] = cand[hit]
            xi_out[todo[hit]] = np.clip(xi[hit], 0.0, 1.0)
        return cell, xi_out


def compute_eval_params(mesh, points, float_type=np.float64, locator=None):
    """Counterpart of cuda/utils.py:117-154: ``points`` is 3 x n (rows x, y, z) as there; returns
    ``(points_on_proc [m, 3], cells [m])`` for the points that lie in a cell of THIS rank's mesh.  ``mesh`` needs
    ``x_dofs`` and ``x_g``."""
    pts = np.asarray(points, dtype=np.float64).T
    loc = locator if locator is not None else CellLocator(mesh.x_dofs, mesh.x_g)
    cell, _ = loc.locate(pts)
    keep = cell >= 0
    return pts[keep].astype(float_type), cell[keep].tolist()


def _lagrange_1d(nodes, x):
    """Values [len(x), n] of the n Lagrange polynomials on ``nodes`` at ``x``."""
    n = nodes.size
    out = np.ones((x.size, n))
    for i in range(n):
        for j in range(n):
            if j != i:
                out[:, i] *= (x - nodes[j]) / (nodes[i] - nodes[j])
    return out


def eval_function(mesh, u, points_on_proc, cells, locator=None):
    """``Function.eval(points, cells)`` (cuda/demo_linear_box.py:587-590) for a field given by its local dof vector
    ``u`` (numpy, owned + ghosts) on ``mesh`` (``P``, ``dofmap`` in tensor-product local order, ``x_dofs``, ``x_g``):
    values [m] at ``points_on_proc`` [m, 3], each inside the cell ``cells[k]``."""
    pts = np.asarray(points_on_proc, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    if pts.shape[0] == 0:
        return np.zeros(0)
    P = int(mesh.P)
    n = P + 1
    nodes, _ = gll_points_weights(P)
    xi, ok = _invert(np.asarray(mesh.x_g, dtype=np.float64)[np.asarray(mesh.x_dofs)[cells]], pts)
    if not ok.all():
        raise ValueError("a point could not be mapped into the cell it was said to lie in")
    xi = np.clip(xi, 0.0, 1.0)
    Lx, Ly, Lz = (_lagrange_1d(nodes, xi[:, a]) for a in range(3))
    uc = np.asarray(u, dtype=np.float64)[np.asarray(mesh.dofmap)[cells]].reshape(-1, n, n, n)  # l = i n^2 + j n + k
    return np.einsum("mijk,mi,mj,mk->m", uc, Lx, Ly, Lz)
