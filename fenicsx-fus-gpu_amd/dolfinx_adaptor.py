"""
Adaptor from what a dolfinx driver holds to what the operators and ``HaloApply`` take.

dolfinx / basix exist nowhere in this pipeline, so nothing here imports them: the functions take the
plain arrays the reference's drivers pull out of dolfinx objects, and anything that quacks like
``dolfinx.common.IndexMap`` (``size_local``, ``num_ghosts``, ``ghosts``, ``owners``, ``local_range``,
``index_to_dest_ranks()``).  What the reference's drivers do, and the counterpart here:

  cuda/demo_linear_box.py:167-176   V on the standard basix element; ``perm = argsort(tp_element.dof_ordering)``,
                                    ``dofmap = V.dofmap.list[:, perm]``            -> ``tensor_product_dofmap``
  cuda/demo_linear_box.py:178-207   geometry dofmap / coordinates ``(x_dofs, x_g)``, per-cell material arrays
  cuda/utils.py:8-78                ``compute_scatterer_data(V.dofmap.index_map)``   -> ``utils.compute_scatterer_data``
  cuda/demo_linear_box.py:537-553   scatter_fwd -> operators -> scatter_rev per stage -> ``HaloApply`` (needs the cells
                                    that touch ghost dofs stored first)            -> ``partition_for_overlap``

  cuda/demo_nonlinear_bowl.py:98-105,255-345 / cuda/utils.py:81-114   XDMF mesh with cell / facet tags; tagged boundary facets as
                                    ``(cell, local facet)`` pairs (``facet_integration_domain``)      -> ``ArrayMesh``: the mesh the solver
                                    classes step, built from those plain arrays (no structured box needed)

Tested with mock inputs shaped like dolfinx's (ghosts in arbitrary order, cells in arbitrary order,
basix-like local dof order): tests/test_dolfinx_adaptor.py, tests/test_halo_gpu.py.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .utils import boundary_first_cell_order


def tensor_product_dofmap(dofmap_list, dof_ordering):
    """``V.dofmap.list`` of a space built on the STANDARD basix element -> the tensor-product local order
    ``l = i n^2 + j n + k`` the kernels use (cuda/demo_linear_box.py:167-176: ``perm = argsort(np.array(
    tp_element.dof_ordering))``; ``dofmap = dofmap[:, perm]``).  Spaces built directly on the tp element
    (numba-cpu/test_operators.py:76-81) are already in that order: pass ``dof_ordering=None``."""
    dm = np.asarray(dofmap_list)
    if dof_ordering is None:
        return np.ascontiguousarray(dm.astype(np.int32))
    order = np.asarray(dof_ordering)
    if order.size != dm.shape[1] or not np.array_equal(np.sort(order), np.arange(dm.shape[1])):
        raise ValueError("dof_ordering must be a permutation of the local dofs")
    perm = np.argsort(order)
    return np.ascontiguousarray(dm[:, perm].astype(np.int32))


@dataclass
class RankMesh:
    """What ``HaloApply`` / the solvers need to know about one rank's part of a mesh."""

    dofmap: np.ndarray  # int32 [ncells, n^3], tensor-product local order, ghost-touching cells first
    index_map: object
    nlocal: int
    nghost: int
    ncells: int
    num_boundary_cells: int
    cell_permutation: np.ndarray  # new cell c is the driver's cell cell_permutation[c]

    @property
    def ndofs(self):
        return self.nlocal + self.nghost


def partition_for_overlap(dofmap, index_map, per_cell=()):
    """Order one rank's cells so that the cells touching a ghost dof come first (``HaloApply`` overlaps
    the halo exchange with the rest) and apply that order to every per-cell array.

    ``dofmap``: int32 [ncells, n^3] in tensor-product local order (``tensor_product_dofmap``), local dof
    indices with ghosts at ``[size_local, size_local + num_ghosts)`` as dolfinx numbers them;
    ``per_cell``: arrays whose first axis is the cell (G, detJ, cell constants, x_dofs, ...).
    Returns ``(RankMesh, [per-cell arrays in the new order])``.  Boundary facets: map a facet's cell c
    through ``np.argsort(rank_mesh.cell_permutation)[c]``."""
    dm = np.asarray(dofmap)
    nlocal = int(index_map.size_local)
    nghost = int(index_map.num_ghosts)
    if dm.size and (dm.min() < 0 or dm.max() >= nlocal + nghost):
        raise ValueError("dofmap entries outside [0, size_local + num_ghosts)")
    perm, nb = boundary_first_cell_order(dm, nlocal)
    rm = RankMesh(dofmap=np.ascontiguousarray(dm[perm].astype(np.int32)), index_map=index_map, nlocal=nlocal,
                  nghost=nghost, ncells=int(dm.shape[0]), num_boundary_cells=int(nb), cell_permutation=perm)
    return rm, [np.ascontiguousarray(np.asarray(a)[perm]) for a in per_cell]


def local_facet_dofs(P):
    """``[6, n^2]`` tensor-product local dofs in the closure of each local facet of a hexahedron (the reference takes them
    from ``basix_element.entity_closure_dofs[2]`` and permutes them like the dofmap, numba-cpu/test_operators.py:127-129):
    local facet f = (axis, side) of ``precompute.HEX_FACET_AXIS_SIDE``, dofs ordered like the facet's quadrature points."""
    from .precompute import HEX_FACET_AXIS_SIDE

    n = int(P) + 1
    out = np.empty((6, n * n), dtype=np.int32)
    a_, b_ = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a_, b_ = a_.reshape(-1), b_.reshape(-1)
    for f, (axis, side) in enumerate(HEX_FACET_AXIS_SIDE):
        idx = [None, None, None]
        free = [a for a in range(3) if a != axis]
        idx[free[0]], idx[free[1]] = a_, b_
        idx[axis] = np.full(n * n, side * (n - 1))
        out[f] = idx[0] * n * n + idx[1] * n + idx[2]
    return out


class ArrayMesh:
    """One rank's part of a general (unstructured) trilinear hexahedral mesh, handed over as the plain arrays the
    reference's production drivers hold (cuda/demo_nonlinear_bowl.py:98-105,255-345): everything ``LinearSpectral3D`` /
    ``WesterveltSpectral3D`` / ``HaloApply`` / ``time_step_parameters`` read from a mesh, with no structured box behind it.

      P            polynomial degree of the space
      dofmap       int [ncells, n^3]  local dof indices in tensor-product local order (``tensor_product_dofmap``), owned dofs
                   ``[0, size_local)``, ghosts after them, as dolfinx numbers them
      x_dofs       int [ncells, 8]    geometry dofmap (P1 hexahedron, vertex v = vx + 2 vy + 4 vz)
      x_g          float [nverts, 3]  vertex coordinates
      index_map    anything with ``dolfinx.common.IndexMap``'s members (None: one rank, every dof owned)
      facet_tags   {tag: int [nfacets, 2]}  tagged boundary facets as ``(cell, local facet)`` pairs -- what
                   ``facet_integration_domain`` returns (cuda/utils.py:81-114), cells in the ORDER OF ``dofmap`` AS PASSED
      source_tag / absorbing_tag   the tags the solvers take their two facet sets from (the reference's meshes: 1 and 2)

    With an index map the cells are re-ordered so that those touching a ghost dof come first (``HaloApply`` overlaps the
    exchange with the rest); every per-cell array -- and the cell column of the facet tags -- follows, and
    ``cell_permutation`` says how (new cell c is the caller's cell ``cell_permutation[c]``; per-cell material arrays go
    through ``permute_cells``)."""

    def __init__(self, P, dofmap, x_dofs, x_g, index_map=None, facet_tags=None, source_tag=1, absorbing_tag=2, ndofs_global=None):
        self.P, self.n = int(P), int(P) + 1
        dm = np.asarray(dofmap)
        if dm.ndim != 2 or dm.shape[1] != self.n**3:
            raise ValueError(f"dofmap must be [ncells, {self.n ** 3}] for P = {self.P}")
        xd = np.asarray(x_dofs)
        if xd.shape != (dm.shape[0], 8):
            raise ValueError("x_dofs must be [ncells, 8] (trilinear hexahedra)")
        self.index_map = index_map
        if index_map is not None:
            self.nlocal, self.nghost = int(index_map.size_local), int(index_map.num_ghosts)
        else:
            self.nlocal, self.nghost = int(dm.max()) + 1 if dm.size else 0, 0
        if dm.size and (dm.min() < 0 or dm.max() >= self.nlocal + self.nghost):
            raise ValueError("dofmap entries outside [0, size_local + num_ghosts)")
        perm, nb = boundary_first_cell_order(dm, self.nlocal)
        self.cell_permutation = perm
        self.num_boundary_cells = int(nb)
        self.ncells = int(dm.shape[0])
        self.dofmap = np.ascontiguousarray(dm[perm].astype(np.int32))
        self.x_dofs = np.ascontiguousarray(xd[perm].astype(np.int32))
        self.x_g = np.ascontiguousarray(np.asarray(x_g))
        if self.x_g.ndim != 2 or self.x_g.shape[1] != 3 or (xd.size and xd.max() >= self.x_g.shape[0]):
            raise ValueError("x_g must be [nverts, 3] and hold every vertex x_dofs names")
        inv = np.empty(self.ncells, dtype=np.int64)
        inv[perm] = np.arange(self.ncells)
        self.facet_tags = {}
        for tag, bd in (facet_tags or {}).items():
            bd = np.asarray(bd).reshape(-1, 2)
            if bd.size and (bd[:, 0].min() < 0 or bd[:, 0].max() >= self.ncells or bd[:, 1].min() < 0 or bd[:, 1].max() > 5):
                raise ValueError(f"facet tag {tag}: (cell, local facet) pairs out of range")
            self.facet_tags[tag] = np.stack([inv[bd[:, 0]], bd[:, 1]], axis=1).astype(np.int32) if bd.size else np.zeros((0, 2), np.int32)
        self.source_tag, self.absorbing_tag = source_tag, absorbing_tag
        self.ndofs_global = int(ndofs_global) if ndofs_global is not None else (int(index_map.size_global) if index_map is not None and hasattr(index_map, "size_global") else self.nlocal)

    @property
    def ndofs(self):
        return self.nlocal + self.nghost

    def permute_cells(self, *per_cell):
        """Per-cell arrays of the caller (material constants, ...) in this mesh's cell order."""
        out = [np.ascontiguousarray(np.asarray(a)[self.cell_permutation]) for a in per_cell]
        return out[0] if len(out) == 1 else out

    def boundary_facets(self, tags):
        """The ``(cell, local facet)`` pairs of the given tags, concatenated (the solvers ask for ``[source_tag]`` and
        ``[absorbing_tag]``): the reference's ``facet_integration_domain(ft.indices[ft.values == tag], mesh)``."""
        got = [self.facet_tags[t] for t in tags if t in self.facet_tags]
        return np.concatenate(got, axis=0).astype(np.int32) if got else np.zeros((0, 2), dtype=np.int32)

    def local_facet_dofs(self):
        return local_facet_dofs(self.P)

    def facet_dofmap(self, boundary_data):
        """``bfacet_dofmap[i, :] = dofmap[cell][local_facet_dof[local_facet]]`` (numba-cpu/test_operators.py:161-167)."""
        bd = np.asarray(boundary_data)
        if bd.shape[0] == 0:
            return np.zeros((0, self.n * self.n), dtype=np.int32)
        lfd = local_facet_dofs(self.P)
        return np.ascontiguousarray(self.dofmap[bd[:, 0][:, None], lfd[bd[:, 1]]].astype(np.int32))
