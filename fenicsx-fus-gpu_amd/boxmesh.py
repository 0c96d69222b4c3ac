"""
Synthetic structured hexahedral box meshes (host, numpy) -- the stand-in for
the dolfinx plumbing the reference drivers use and this container lacks:

  create_box(comm, ..., (N, N, N), CellType.hexahedron, GhostMode.none)
                                            numba-cpu/time_operators.py:51-59
  basix.create_tp_element(...) / functionspace(mesh, element) / V.dofmap.list
                                            numba-cpu/time_operators.py:71-80
  V.dofmap.index_map (size_local, num_ghosts, ghosts, owners, local_range,
  index_to_dest_ranks)                      cuda/utils.py:8-78

Layout: box [0, L]^3, ``Nx x Ny x Nz`` cells, degree ``P``,
``(P Nx + 1)(P Ny + 1)(P Nz + 1)`` dofs on the tensor grid of GLL nodes, local
dof ``l = i n^2 + j n + k`` (x slowest, tensor-product order, ascending GLL
nodes), P1 geometry with basix vertex order ``v = vx + 2 vy + 4 vz``.

Partitioning follows dolfinx ``GhostMode.none`` semantics: a non-overlapping
cell partition (structured ``px x py x pz`` blocks); each rank numbers its owned
dofs ``[0, nlocal)`` then its ghosts ``[nlocal, nlocal + nghost)``; a dof shared
by several ranks is owned by the lowest rank touching it, which for block
partitions means a rank owns the *upper* faces of its block and ghosts the
*lower* ones.  Global indices are process-blockwise (``local_range[0] + local``)
exactly like dolfinx, so ``cuda/utils.py:compute_scatterer_data`` semantics
carry over unchanged.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .gll import gll_points_weights


class _AdjacencyList:
    """Minimal stand-in for ``dolfinx.graph.AdjacencyList`` (``.array``,
    ``.offsets``, ``.links(i)``) as consumed at cuda/utils.py:40-47."""

    def __init__(self, offsets: np.ndarray, array: np.ndarray):
        self.offsets = offsets
        self.array = array

    def links(self, i: int) -> np.ndarray:
        return self.array[self.offsets[i] : self.offsets[i + 1]]


@dataclass
class IndexMap:
    """The subset of ``dolfinx.common.IndexMap`` the reference touches."""

    size_local: int
    ghosts: np.ndarray  # int64 global indices of the ghost dofs
    owners: np.ndarray  # int32 owning rank of each ghost
    local_range: tuple
    size_global: int
    rank: int = 0
    comm_size: int = 1
    _dest_offsets: np.ndarray = field(default=None, repr=False)
    _dest_array: np.ndarray = field(default=None, repr=False)

    @property
    def num_ghosts(self) -> int:
        return int(self.ghosts.size)

    def index_to_dest_ranks(self) -> _AdjacencyList:
        return _AdjacencyList(self._dest_offsets, self._dest_array)


def _split(ncell: int, parts: int, idx: int):
    """Cell range [c0, c1) of block ``idx`` out of ``parts`` along one axis."""
    base, rem = divmod(ncell, parts)
    c0 = idx * base + min(idx, rem)
    return c0, c0 + base + (1 if idx < rem else 0)


def rank_to_coords(rank: int, grid):
    px, py, pz = grid
    return rank // (py * pz), (rank // pz) % py, rank % pz


def coords_to_rank(rc, grid) -> int:
    px, py, pz = grid
    return (rc[0] * py + rc[1]) * pz + rc[2]


def default_grid(world_size: int):
    """2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2; otherwise near-cubic."""
    g = [1, 1, 1]
    w, a = world_size, 0
    f = 2
    factors = []
    while w > 1:
        while w % f == 0:
            factors.append(f)
            w //= f
        f += 1
    for f in sorted(factors, reverse=True):
        a = int(np.argmin(g))
        g[a] *= f
    g.sort(reverse=True)
    return tuple(g)


class BoxMesh:
    """One rank's part of a structured hex box (the whole box when grid=(1,1,1)).

    Attributes
    ----------
    dofmap : int32 [ncells, n^3]   local dof indices, tensor-product local order
    x_dofs : int32 [ncells, 8]     local vertex indices (basix P1 hex order)
    x_g    : float [nverts, 3]     vertex coordinates
    index_map : IndexMap
    num_boundary_cells : cells [0, num_boundary_cells) touch at least one ghost
        dof (they need the forward halo before, and feed the reverse halo
        after, an operator apply); the rest are interior.

    ``ghost_order``: "owner" (default) numbers the ghosts owner by owner, which lets the halo
    exchange use the ghost block of a vector as its message buffer; "lex" (lexicographic in the local
    grid, owners interleaved) or an integer seed (random) reproduce what a general dolfinx
    ``IndexMap`` looks like and exercise the unpack_fwd / pack_rev kernels.
    """

    def __init__(
        self,
        P: int,
        ncells,
        grid=(1, 1, 1),
        rank: int = 0,
        length=1.0,
        perturb: float = 0.0,
        seed: int = 0,
        warp=None,
        dtype=np.float64,
        ghost_order="owner",
    ):
        if np.isscalar(ncells):
            ncells = (int(ncells),) * 3
        if np.isscalar(length):
            length = (float(length),) * 3
        self.P = int(P)
        self.n = self.P + 1
        self.global_cells = tuple(int(c) for c in ncells)
        self.grid = tuple(int(g) for g in grid)
        self.rank = int(rank)
        self.comm_size = int(np.prod(self.grid))
        self.length = tuple(float(x) for x in length)
        self.dtype = np.dtype(dtype)
        P_, n = self.P, self.n
        rc = rank_to_coords(self.rank, self.grid)
        self.rank_coords = rc

        # ---- per-axis ranges -------------------------------------------------
        self.cell_range = [_split(self.global_cells[a], self.grid[a], rc[a]) for a in range(3)]
        cr = self.cell_range
        self.local_cells = tuple(c1 - c0 for c0, c1 in cr)
        # dof-grid coordinates (global) covered by this block, inclusive
        full_lo = [P_ * cr[a][0] for a in range(3)]
        full_hi = [P_ * cr[a][1] for a in range(3)]
        has_lower = [cr[a][0] > 0 for a in range(3)]
        own_lo = [full_lo[a] + (1 if has_lower[a] else 0) for a in range(3)]
        fdim = [full_hi[a] - full_lo[a] + 1 for a in range(3)]
        odim = [full_hi[a] - own_lo[a] + 1 for a in range(3)]
        self._full_lo, self._full_dim, self._own_lo, self._own_dim = full_lo, fdim, own_lo, odim
        self.global_dof_dims = tuple(P_ * self.global_cells[a] + 1 for a in range(3))
        nlocal = int(np.prod(odim))

        # ---- local numbering: owned lexicographic, then ghosts ----------------
        ax = [np.arange(fdim[a]) for a in range(3)]
        is_ghost_ax = [(ax[a] == 0) & has_lower[a] for a in range(3)]
        ghost_mask = is_ghost_ax[0][:, None, None] | is_ghost_ax[1][None, :, None] | is_ghost_ax[2][None, None, :]
        lid = np.empty(fdim, dtype=np.int32)
        shift = [1 if has_lower[a] else 0 for a in range(3)]
        oi = [ax[a] - shift[a] for a in range(3)]
        owned_idx = (oi[0][:, None, None] * odim[1] + oi[1][None, :, None]) * odim[2] + oi[2][None, None, :]
        lid[...] = owned_idx
        gpos = np.nonzero(ghost_mask.reshape(-1))[0]  # lexicographic order in the full local grid
        nghost = gpos.size
        if nghost:
            # ghosts are numbered owner by owner (ascending rank), lexicographically inside an owner:
            # the ghost block of a vector is then exactly the concatenation of the per-owner
            # messages, so the halo exchange can receive into / send from it without an
            # unpack / pack pass (scatterer._Scatter detects the identity index list)
            gi_, gj_, gk_ = np.unravel_index(gpos, fdim)
            orc = [rc[a] - ((np.array((gi_, gj_, gk_)[a]) == 0) & has_lower[a]).astype(np.int64) for a in range(3)]
            gowner = (orc[0] * self.grid[1] + orc[1]) * self.grid[2] + orc[2]
            if ghost_order == "owner":
                gpos = gpos[np.argsort(gowner, kind="stable")]
            elif ghost_order == "lex":
                pass  # lexicographic in the local grid: owners interleave, like a dolfinx IndexMap in general
            else:  # int seed: arbitrary ghost numbering
                gpos = gpos[np.random.default_rng(int(ghost_order)).permutation(gpos.size)]
        lid.reshape(-1)[gpos] = nlocal + np.arange(nghost, dtype=np.int32)
        self._lid = lid
        self.nlocal, self.nghost = nlocal, int(nghost)

        # ---- index map ---------------------------------------------------------
        all_nlocal = np.empty(self.comm_size, dtype=np.int64)
        own_lo_all, own_dim_all = [], []
        for r in range(self.comm_size):
            rcr = rank_to_coords(r, self.grid)
            lo, dim = [], []
            for a in range(3):
                c0, c1 = _split(self.global_cells[a], self.grid[a], rcr[a])
                l = P_ * c0 + (1 if c0 > 0 else 0)
                lo.append(l)
                dim.append(P_ * c1 - l + 1)
            own_lo_all.append(lo)
            own_dim_all.append(dim)
            all_nlocal[r] = int(np.prod(dim))
        offsets = np.concatenate(([0], np.cumsum(all_nlocal)))
        self._rank_offsets = offsets
        gi, gj, gk = np.unravel_index(gpos, fdim)
        gflag = [is_ghost_ax[0][gi], is_ghost_ax[1][gj], is_ghost_ax[2][gk]]
        owner_rc = [rc[a] - gflag[a].astype(np.int64) for a in range(3)]
        owners = ((owner_rc[0] * self.grid[1] + owner_rc[1]) * self.grid[2] + owner_rc[2]).astype(np.int32)
        gglob = [gi + full_lo[0], gj + full_lo[1], gk + full_lo[2]]  # global grid coordinates
        ghosts = np.empty(nghost, dtype=np.int64)
        for r in np.unique(owners):
            m = owners == r
            lo, dim = own_lo_all[r], own_dim_all[r]
            loc = ((gglob[0][m] - lo[0]) * dim[1] + (gglob[1][m] - lo[1])) * dim[2] + (gglob[2][m] - lo[2])
            ghosts[m] = offsets[r] + loc
        # destination ranks of owned dofs (who ghosts them): upper shared planes
        has_upper = [rc[a] < self.grid[a] - 1 for a in range(3)]
        oax = [np.arange(odim[a]) for a in range(3)]
        up = [(oax[a] == odim[a] - 1) & has_upper[a] for a in range(3)]
        U = [
            np.broadcast_to(up[0][:, None, None], odim).reshape(-1),
            np.broadcast_to(up[1][None, :, None], odim).reshape(-1),
            np.broadcast_to(up[2][None, None, :], odim).reshape(-1),
        ]
        counts = ((1 + U[0].astype(np.int64)) * (1 + U[1]) * (1 + U[2])) - 1
        d_off = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        d_arr = np.empty(int(d_off[-1]), dtype=np.int32)
        fill = np.zeros(nlocal, dtype=np.int64)
        for sx in (0, 1):
            for sy in (0, 1):
                for sz in (0, 1):
                    if sx + sy + sz == 0:
                        continue
                    m = np.ones(nlocal, dtype=bool)
                    if sx:
                        m &= U[0]
                    if sy:
                        m &= U[1]
                    if sz:
                        m &= U[2]
                    idx = np.nonzero(m)[0]
                    if idx.size == 0:
                        continue
                    r = coords_to_rank((rc[0] + sx, rc[1] + sy, rc[2] + sz), self.grid)
                    d_arr[d_off[idx] + fill[idx]] = r
                    fill[idx] += 1
        self.index_map = IndexMap(
            size_local=nlocal,
            ghosts=ghosts,
            owners=owners,
            local_range=(int(offsets[self.rank]), int(offsets[self.rank + 1])),
            size_global=int(offsets[-1]),
            rank=self.rank,
            comm_size=self.comm_size,
            _dest_offsets=d_off,
            _dest_array=d_arr,
        )

        # ---- cells (boundary cells first) and dofmap ---------------------------
        lc = self.local_cells
        cx, cy, cz = np.meshgrid(np.arange(lc[0]), np.arange(lc[1]), np.arange(lc[2]), indexing="ij")
        cx, cy, cz = cx.reshape(-1), cy.reshape(-1), cz.reshape(-1)
        touches_ghost = (
            ((cx == 0) & has_lower[0]) | ((cy == 0) & has_lower[1]) | ((cz == 0) & has_lower[2])
        )
        order = np.concatenate((np.nonzero(touches_ghost)[0], np.nonzero(~touches_ghost)[0]))
        self.num_boundary_cells = int(touches_ghost.sum())
        cx, cy, cz = cx[order], cy[order], cz[order]
        self._cell_ijk = np.stack([cx, cy, cz], axis=1)
        self.ncells = int(cx.size)
        li = np.arange(n)
        I, J, K = np.meshgrid(li, li, li, indexing="ij")
        I, J, K = I.reshape(-1), J.reshape(-1), K.reshape(-1)
        flat = ((cx[:, None] * P_ + I[None, :]) * fdim[1] + (cy[:, None] * P_ + J[None, :])) * fdim[2] + (
            cz[:, None] * P_ + K[None, :]
        )
        self.dofmap = np.ascontiguousarray(lid.reshape(-1)[flat].astype(np.int32))

        # ---- geometry (P1): local vertex grid ---------------------------------
        vdim = [lc[a] + 1 for a in range(3)]
        gv = [np.arange(vdim[a]) + cr[a][0] for a in range(3)]  # global vertex coords
        h = [self.length[a] / self.global_cells[a] for a in range(3)]
        self.h = tuple(h)
        VX, VY, VZ = np.meshgrid(gv[0], gv[1], gv[2], indexing="ij")
        xg = np.stack([VX.reshape(-1) * h[0], VY.reshape(-1) * h[1], VZ.reshape(-1) * h[2]], axis=1).astype(np.float64)
        if perturb:
            gvd = [self.global_cells[a] + 1 for a in range(3)]
            rng = np.random.default_rng(seed)
            disp = rng.uniform(-1.0, 1.0, size=(gvd[0] * gvd[1] * gvd[2], 3))
            gid = (VX.reshape(-1) * gvd[1] + VY.reshape(-1)) * gvd[2] + VZ.reshape(-1)
            xg += perturb * np.asarray(h)[None, :] * disp[gid]
        if warp is not None:
            xg = np.asarray(warp(xg), dtype=np.float64)
        self.x_g = np.ascontiguousarray(xg.astype(self.dtype))
        xd = np.empty((self.ncells, 8), dtype=np.int32)
        for v in range(8):
            bx, by, bz = v & 1, (v >> 1) & 1, (v >> 2) & 1
            xd[:, v] = ((cx + bx) * vdim[1] + (cy + by)) * vdim[2] + (cz + bz)
        self.x_dofs = xd

    # ------------------------------------------------------------------------
    @property
    def ndofs(self) -> int:
        return self.nlocal + self.nghost

    @property
    def ndofs_global(self) -> int:
        return int(np.prod(self.global_dof_dims))

    def dof_coordinates(self) -> np.ndarray:
        """Physical coordinates ``[ndofs, 3]`` of the local dofs (image of the
        reference GLL nodes under the trilinear geometry map)."""
        pts, _ = gll_points_weights(self.P)
        n = self.n
        X, Y, Z = np.meshgrid(pts, pts, pts, indexing="ij")
        ref = np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], axis=1)
        shp = np.empty((n**3, 8))
        for v in range(8):
            b = (v & 1, (v >> 1) & 1, (v >> 2) & 1)
            f = [ref[:, a] if b[a] else 1.0 - ref[:, a] for a in range(3)]
            shp[:, v] = f[0] * f[1] * f[2]
        out = np.empty((self.ndofs, 3))
        xg = self.x_g.astype(np.float64)
        chunk = 8192
        for c0 in range(0, self.ncells, chunk):
            c1 = min(c0 + chunk, self.ncells)
            co = np.einsum("lv,cvd->cld", shp, xg[self.x_dofs[c0:c1]])
            out[self.dofmap[c0:c1].reshape(-1)] = co.reshape(-1, 3)
        return out

    def global_lexicographic_ids(self) -> np.ndarray:
        """Rank-independent id of each local dof: its lexicographic index in
        the global ``(P Nx + 1)(P Ny + 1)(P Nz + 1)`` dof grid.  Used by tests
        to compare partitioned results with a single-rank run."""
        fdim, lo, gd = self._full_dim, self._full_lo, self.global_dof_dims
        gi, gj, gk = np.meshgrid(
            np.arange(fdim[0]) + lo[0], np.arange(fdim[1]) + lo[1], np.arange(fdim[2]) + lo[2], indexing="ij"
        )
        lex = (gi * gd[1] + gj) * gd[2] + gk
        out = np.empty(self.ndofs, dtype=np.int64)
        out[self._lid.reshape(-1)] = lex.reshape(-1)
        return out

    # ------------------------------------------------------------------------
    def local_facet_dofs(self) -> np.ndarray:
        """``[6, n^2]`` local dofs in the closure of each local facet (stand-in
        for ``basix_element.entity_closure_dofs[2]``,
        numba-cpu/test_operators.py:127-129), ordered like ``facet_points``."""
        from .dolfinx_adaptor import local_facet_dofs

        return local_facet_dofs(self.P)

    def boundary_facets(self, faces=None) -> np.ndarray:
        """``boundary_data[i] = (cell, local_facet)`` for this rank's cells on
        the given faces of the *global* box (stand-in for
        ``locate_entities_boundary`` + ``facet_integration_domain``,
        numba-cpu/test_operators.py:120-126).  ``faces``: iterable of local
        facet ids 0..5 = (z=0, y=0, x=0, x=1, y=1, z=1); default all six."""
        from .precompute import HEX_FACET_AXIS_SIDE

        if faces is None:
            faces = range(6)
        out = []
        ijk = self._cell_ijk
        for f in faces:
            axis, side = HEX_FACET_AXIS_SIDE[f]
            c0, c1 = self.cell_range[axis]
            if side == 0 and c0 != 0:
                continue
            if side == 1 and c1 != self.global_cells[axis]:
                continue
            target = 0 if side == 0 else self.local_cells[axis] - 1
            cells = np.nonzero(ijk[:, axis] == target)[0]
            out.append(np.stack([cells, np.full(cells.size, f)], axis=1))
        if not out:
            return np.zeros((0, 2), dtype=np.int32)
        return np.concatenate(out, axis=0).astype(np.int32)

    def facet_dofmap(self, boundary_data: np.ndarray) -> np.ndarray:
        """``bfacet_dofmap[i, :] = dofmap[cell][local_facet_dof[local_facet]]``
        (numba-cpu/test_operators.py:161-167)."""
        lfd = self.local_facet_dofs()
        bd = np.asarray(boundary_data)
        if bd.shape[0] == 0:
            return np.zeros((0, self.n * self.n), dtype=np.int32)
        return np.ascontiguousarray(self.dofmap[bd[:, 0][:, None], lfd[bd[:, 1]]].astype(np.int32))
