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
