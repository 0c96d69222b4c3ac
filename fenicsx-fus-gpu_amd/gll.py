"""
Gauss-Lobatto-Legendre (GLL) tables for degree-P spectral elements.

The reference obtains these from basix (un-vendored):
  - 1-D quadrature points/weights: ``basix.quadrature.make_quadrature(interval,
    Q[P], QuadratureType.gll)``  (numba-cpu/time_operators.py:205-207)
  - 1-D derivative table ``dphi_1D = element_1D.tabulate(1, pts_1D)[1, :, :, 0]``
    (numba-cpu/time_operators.py:209-213), flat row-major ``[q, i]``.
basix is not available here, so the tables are generated from their
mathematical definition and pinned by known-answer tests (tests/test_gll.py):
nodes are the roots of (1 - xi^2) P'_P(xi) mapped to [0, 1], weights integrate
polynomials of degree <= 2P-1 exactly, D differentiates degree-<=P polynomials
exactly.

Local ordering used throughout this package: nodes ascending in [0, 1]
(basix lists the two end points first; the kernels are agnostic as long as
``dofmap``, ``G``/``detJ`` quadrature order and ``dphi`` agree -- SURVEY 8c).
"""

from __future__ import annotations

import numpy as np
from numpy.polynomial import legendre as _leg

# Quadrature-degree map duplicated in every reference file
# (numba-cpu/time_operators.py:35-45); each entry yields P+1 GLL points.
QUADRATURE_DEGREE = {2: 3, 3: 4, 4: 6, 5: 8, 6: 10, 7: 12, 8: 14, 9: 16, 10: 18}


def gll_points_weights(P: int, dtype=np.float64):
    """GLL nodes (ascending) and weights on the reference interval [0, 1].

    Returns ``(pts[P+1], wts[P+1])`` with ``sum(wts) == 1``.
    """
    if P < 1:
        raise ValueError("P must be >= 1")
    n = P + 1
    LP = _leg.Legendre.basis(P)
    dLP = LP.deriv()
    if P == 1:
        x = np.array([-1.0, 1.0])
    else:
        xi = np.sort(np.real(dLP.roots()))
        # Newton polish on P'_P (roots of a Legendre-series companion matrix
        # are good to ~1e-14; polishing brings them to round-off).
        d2 = dLP.deriv()
        for _ in range(3):
            xi = xi - dLP(xi) / d2(xi)
        x = np.concatenate(([-1.0], xi, [1.0]))
    w = 2.0 / (P * n * LP(x) ** 2)
    pts = 0.5 * (x + 1.0)
    wts = 0.5 * w
    # symmetrise against round-off
    pts = 0.5 * (pts + (1.0 - pts[::-1]))
    wts = 0.5 * (wts + wts[::-1])
    return pts.astype(dtype), wts.astype(dtype)


def lagrange_derivative_matrix(nodes: np.ndarray) -> np.ndarray:
    """``D[q, i] = l_i'(nodes[q])`` for the Lagrange basis on ``nodes``.

    Barycentric form; rows sum to zero by construction (negative-sum trick).
    """
    x = np.asarray(nodes, dtype=np.float64)
    n = x.size
    diff = x[:, None] - x[None, :]
    np.fill_diagonal(diff, 1.0)
    bw = 1.0 / np.prod(diff, axis=1)  # barycentric weights
    D = (bw[None, :] / bw[:, None]) / diff
    np.fill_diagonal(D, 0.0)
    np.fill_diagonal(D, -np.sum(D, axis=1))
    return D


def tabulate_1d(P: int, dtype=np.float64):
    """1-D tables consumed by the operators.

    Returns ``(pts, wts, dphi)`` where ``dphi`` is the derivative table
    ``[q, i]`` of shape ``(P+1, P+1)``. ``dphi.flatten()`` is what
    ``stiffness_operator(P, dphi, float_type)`` of numba-cpu/operators.py:71
    expects; the 2-D array is what the cuda-style operator expects
    (cuda/operators.py:73-192, last argument).
    """
    pts, wts = gll_points_weights(P, np.float64)
    D = lagrange_derivative_matrix(pts)
    return pts.astype(dtype), wts.astype(dtype), np.ascontiguousarray(D.astype(dtype))


def tensor_weights_3d(wts: np.ndarray) -> np.ndarray:
    """Hex quadrature weights in tensor order ``q = qx*n*n + qy*n + qz``."""
    w = np.asarray(wts)
    return (w[:, None, None] * w[None, :, None] * w[None, None, :]).reshape(-1)


def tensor_weights_2d(wts: np.ndarray) -> np.ndarray:
    w = np.asarray(wts)
    return (w[:, None] * w[None, :]).reshape(-1)


def tensor_points_3d(pts: np.ndarray) -> np.ndarray:
    """Hex quadrature points ``[n^3, 3]`` in tensor order (x slowest)."""
    p = np.asarray(pts)
    X, Y, Z = np.meshgrid(p, p, p, indexing="ij")
    return np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], axis=1)
