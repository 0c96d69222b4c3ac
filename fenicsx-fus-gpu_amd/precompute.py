"""
Host-side geometric precompute (run once, CPU): produces the ``detJ``, ``G`` and
boundary-facet ``detJ_f`` arrays that the operators stream.

Same call surface and array conventions as the reference
numba-cpu/precompute.py (identical to cuda/precompute.py):

  compute_scaled_jacobian_determinant(detJ, (x_dofs, x_g), num_cell, dphi, weights)   :76-112
  compute_scaled_geometrical_factor(G, (x_dofs, x_g), num_cell, dphi, weights)        :115-163
  compute_boundary_facets_scaled_jacobian_determinant(detJ_f, (x_dofs, x_g),
                                      boundary_data, dphi_f, weights)                 :17-73

Conventions that callers rely on:
  * ``J_[a, d] = sum_v dphi[a, q, v] * coord_dofs[v, d]`` = d x_d / d X_a
    (the transpose of the usual Jacobian);
  * ``G[c, q, :] = w_q |det J| * (inv(J_).T @ inv(J_))`` upper triangle
    ``(00, 01, 02, 11, 12, 22)``, indexed by *reference* directions;
  * ``detJ[c, q] = |det J_| w_q``;
  * facet: ``|| J_f[:, 0] x J_f[:, 1] || w_q`` with ``J_f = J_.T @ R_facet``.

The reference loops cell by cell under numba; this is a batched numpy
formulation of the same formulas (closed-form 3x3 adjugate instead of
``np.linalg.inv``), checked against the imported reference on the golden
vectors (tests/test_precompute.py).

The geometry tables the reference takes from basix (P1 hexahedron gradients,
``gelement.tabulate(1, pts)[1:, :, :, 0]``, numba-cpu/test_operators.py:103-107)
are generated here by ``tabulate_hex_p1_gradients``.
"""

from __future__ import annotations

import numpy as np

# Reference facet Jacobians of the hexahedron, facet order
# (z=0, y=0, x=0, x=1, y=1, z=1) -- numba-cpu/precompute.py:49-59.
HEX_REFERENCE_FACET_JACOBIAN = np.array(
    [
        [[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]],
        [[1.0, 0.0], [0.0, 0.0], [0.0, 1.0]],
        [[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]],
        [[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]],
        [[1.0, 0.0], [0.0, 0.0], [0.0, 1.0]],
        [[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]],
    ]
)

# (fixed axis, fixed value) of each local facet, same order as above.
HEX_FACET_AXIS_SIDE = ((2, 0), (1, 0), (0, 0), (0, 1), (1, 1), (2, 1))


def tabulate_hex_p1_gradients(points: np.ndarray, dtype=np.float64) -> np.ndarray:
    """Gradients of the 8 trilinear hexahedron shape functions.

    ``points``: ``[nq, 3]`` reference coordinates. Returns ``dphi[3, nq, 8]``
    with vertex ``v = vx + 2 vy + 4 vz`` (x fastest, the basix P1 hexahedron
    vertex order used by ``mesh.geometry.dofmap``).
    """
    X = np.asarray(points, dtype=np.float64)
    nq = X.shape[0]
    out = np.zeros((3, nq, 8), dtype=np.float64)
    for v in range(8):
        b = (v & 1, (v >> 1) & 1, (v >> 2) & 1)
        f = [X[:, a] if b[a] else 1.0 - X[:, a] for a in range(3)]
        df = [np.full(nq, 1.0 if b[a] else -1.0) for a in range(3)]
        out[0, :, v] = df[0] * f[1] * f[2]
        out[1, :, v] = f[0] * df[1] * f[2]
        out[2, :, v] = f[0] * f[1] * df[2]
    return out.astype(dtype)


def facet_points(pts_1d: np.ndarray) -> np.ndarray:
    """Facet quadrature points on the reference hexahedron, ``[6, n^2, 3]``.

    Follows numba-cpu/test_operators.py:141-151: facet-local point ``(p0, p1)``
    is embedded on the two free axes in increasing axis order, ``p0`` slowest.
    """
    p = np.asarray(pts_1d, dtype=np.float64)
    n = p.size
    P0, P1 = np.meshgrid(p, p, indexing="ij")
    p0, p1 = P0.reshape(-1), P1.reshape(-1)
    out = np.zeros((6, n * n, 3))
    for f, (axis, side) in enumerate(HEX_FACET_AXIS_SIDE):
        free = [a for a in range(3) if a != axis]
        out[f, :, free[0]] = p0
        out[f, :, free[1]] = p1
        out[f, :, axis] = float(side)
    return out


def tabulate_facet_gradients(pts_1d: np.ndarray, dtype=np.float64) -> np.ndarray:
    """``dphi_f[6, 3, n^2, 8]`` (numba-cpu/test_operators.py:153-158)."""
    fp = facet_points(pts_1d)
    return np.stack([tabulate_hex_p1_gradients(fp[f], dtype) for f in range(6)], axis=0)


def _jacobians(x_dofs, x_g, cells, dphi):
    """``J_[c, q, a, d]`` for the given cells (float64)."""
    coords = np.asarray(x_g, dtype=np.float64)[np.asarray(x_dofs)[cells]]  # [nc, 8, 3]
    return np.einsum("aqv,cvd->cqad", np.asarray(dphi, dtype=np.float64), coords, optimize=True)


def _det3(J):
    return (
        J[..., 0, 0] * (J[..., 1, 1] * J[..., 2, 2] - J[..., 1, 2] * J[..., 2, 1])
        - J[..., 0, 1] * (J[..., 1, 0] * J[..., 2, 2] - J[..., 1, 2] * J[..., 2, 0])
        + J[..., 0, 2] * (J[..., 1, 0] * J[..., 2, 1] - J[..., 1, 1] * J[..., 2, 0])
    )


def _inv3(J, det):
    inv = np.empty_like(J)
    inv[..., 0, 0] = J[..., 1, 1] * J[..., 2, 2] - J[..., 1, 2] * J[..., 2, 1]
    inv[..., 0, 1] = J[..., 0, 2] * J[..., 2, 1] - J[..., 0, 1] * J[..., 2, 2]
    inv[..., 0, 2] = J[..., 0, 1] * J[..., 1, 2] - J[..., 0, 2] * J[..., 1, 1]
    inv[..., 1, 0] = J[..., 1, 2] * J[..., 2, 0] - J[..., 1, 0] * J[..., 2, 2]
    inv[..., 1, 1] = J[..., 0, 0] * J[..., 2, 2] - J[..., 0, 2] * J[..., 2, 0]
    inv[..., 1, 2] = J[..., 0, 2] * J[..., 1, 0] - J[..., 0, 0] * J[..., 1, 2]
    inv[..., 2, 0] = J[..., 1, 0] * J[..., 2, 1] - J[..., 1, 1] * J[..., 2, 0]
    inv[..., 2, 1] = J[..., 0, 1] * J[..., 2, 0] - J[..., 0, 0] * J[..., 2, 1]
    inv[..., 2, 2] = J[..., 0, 0] * J[..., 1, 1] - J[..., 0, 1] * J[..., 1, 0]
    inv /= det[..., None, None]
    return inv


_CHUNK = 4096  # cells per batch: bounds the float64 temporaries


def compute_scaled_jacobian_determinant(detJ, mesh, num_cell, dphi, weights):
    """``detJ[c, q] = |det J_| w_q`` (numba-cpu/precompute.py:76-112)."""
    x_dofs, x_g = mesh
    w = np.asarray(weights, dtype=np.float64)
    for c0 in range(0, num_cell, _CHUNK):
        cells = np.arange(c0, min(c0 + _CHUNK, num_cell))
        J = _jacobians(x_dofs, x_g, cells, dphi)
        detJ[cells, :] = (np.abs(_det3(J)) * w[None, :]).astype(detJ.dtype)


def compute_scaled_geometrical_factor(G, mesh, num_cell, dphi, weights):
    """``G[c, q, 0..5]`` (numba-cpu/precompute.py:115-163)."""
    x_dofs, x_g = mesh
    w = np.asarray(weights, dtype=np.float64)
    for c0 in range(0, num_cell, _CHUNK):
        cells = np.arange(c0, min(c0 + _CHUNK, num_cell))
        J = _jacobians(x_dofs, x_g, cells, dphi)
        det = _det3(J)
        Ji = _inv3(J, det)  # inv(J_)[d, a]
        # G_ = inv(J_).T @ inv(J_):  G_[a, b] = sum_d Ji[d, a] Ji[d, b]
        sdet = np.abs(det) * w[None, :]
        k = 0
        for a in range(3):
            for b in range(a, 3):
                G[cells, :, k] = (sdet * np.sum(Ji[..., :, a] * Ji[..., :, b], axis=-1)).astype(G.dtype)
                k += 1


def compute_boundary_facets_scaled_jacobian_determinant(detJ_f, mesh, boundary_data, dphi_f, weights):
    """``detJ_f[i, q]`` for ``boundary_data[i] = (cell, local_facet)``
    (numba-cpu/precompute.py:17-73)."""
    x_dofs, x_g = mesh
    w = np.asarray(weights, dtype=np.float64)
    bd = np.asarray(boundary_data)
    if bd.shape[0] == 0:
        return
    for f in range(6):
        sel = np.nonzero(bd[:, 1] == f)[0]
        if sel.size == 0:
            continue
        J = _jacobians(x_dofs, x_g, bd[sel, 0], dphi_f[f])  # [m, q, a, d]
        # J_facet = J_cell.T @ R  ->  J_facet[d, t] = sum_a J[a, d] R[a, t]
        R = HEX_REFERENCE_FACET_JACOBIAN[f]
        Jf = np.einsum("mqad,at->mqdt", J, R)
        cr = np.cross(Jf[..., 0], Jf[..., 1])
        detJ_f[sel, :] = (np.linalg.norm(cr, axis=-1) * w[None, :]).astype(detJ_f.dtype)


# ---------------------------------------------------------------------------------------------
# Device versions (csrc/geometry.hpp): same argument order, all arrays device arrays.
def _dev_args(mesh, dphi, weights, out):
    import torch

    from . import _lib

    x_dofs, x_g = mesh
    dt = out.dtype
    for name, t, d in (("x_g", x_g, dt), ("dphi", dphi, dt), ("weights", weights, dt), ("x_dofs", x_dofs, torch.int32)):
        _lib.require_device_tensor(t, d, name)
    _lib.require_device_tensor(out, dt, "out")
    return _lib.load(), _lib.suffix(dt), x_dofs, x_g


def compute_scaled_geometrical_factor_device(G, mesh, num_cell, dphi, weights, detJ=None):
    """Device twin of ``compute_scaled_geometrical_factor`` (optionally also fills ``detJ``)."""
    from . import _lib

    lib, suf, x_dofs, x_g = _dev_args(mesh, dphi, weights, G)
    nq = weights.numel()
    _lib.check(
        getattr(lib, f"fus_geometry_factors_{suf}")(
            x_g.data_ptr(), x_dofs.data_ptr(), dphi.data_ptr(), weights.data_ptr(), nq, int(num_cell), G.data_ptr(),
            detJ.data_ptr() if detJ is not None else None, _lib.stream_ptr()),
        "fus_geometry_factors")


def compute_scaled_jacobian_determinant_device(detJ, mesh, num_cell, dphi, weights):
    from . import _lib

    lib, suf, x_dofs, x_g = _dev_args(mesh, dphi, weights, detJ)
    _lib.check(
        getattr(lib, f"fus_geometry_factors_{suf}")(
            x_g.data_ptr(), x_dofs.data_ptr(), dphi.data_ptr(), weights.data_ptr(), weights.numel(), int(num_cell),
            None, detJ.data_ptr(), _lib.stream_ptr()),
        "fus_geometry_factors")


def compute_boundary_facets_scaled_jacobian_determinant_device(detJ_f, mesh, boundary_data, dphi_f, weights):
    import torch

    from . import _lib

    lib, suf, x_dofs, x_g = _dev_args(mesh, dphi_f, weights, detJ_f)
    _lib.require_device_tensor(boundary_data, torch.int32, "boundary_data")
    _lib.check(
        getattr(lib, f"fus_facet_jacobian_{suf}")(
            x_g.data_ptr(), x_dofs.data_ptr(), boundary_data.data_ptr(), dphi_f.data_ptr(), weights.data_ptr(),
            weights.numel(), int(boundary_data.shape[0]), detJ_f.data_ptr(), _lib.stream_ptr()),
        "fus_facet_jacobian")
