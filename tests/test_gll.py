"""Known-answer tests pinning the GLL tables the reference takes from basix
('Known-answer tests that pin the path without dolfinx')."""

import numpy as np
import pytest
from numpy.polynomial import legendre as leg

from conftest import pkg


@pytest.mark.parametrize("P", range(1, 11))
def test_gll_nodes_weights(P):
    gll = pkg("gll")
    pts, wts = gll.gll_points_weights(P)
    assert pts.size == P + 1 and pts[0] == 0.0 and pts[-1] == 1.0
    assert np.all(np.diff(pts) > 0)
    xi = 2 * pts - 1
    # interior nodes are the roots of P'_P
    dLP = leg.Legendre.basis(P).deriv()
    if P > 1:
        assert np.max(np.abs(dLP(xi[1:-1]))) < 1e-11 * max(1, P**2)
    assert abs(wts.sum() - 1.0) < 1e-14
    assert np.allclose(wts, wts[::-1], atol=1e-15) and np.allclose(pts, 1 - pts[::-1], atol=1e-15)
    # exact for monomials up to degree 2P-1 on [0, 1]
    for k in range(2 * P):
        assert abs(np.dot(wts, pts**k) - 1.0 / (k + 1)) < 1e-13


@pytest.mark.parametrize("P", range(1, 11))
def test_derivative_matrix(P):
    gll = pkg("gll")
    pts, wts, D = gll.tabulate_1d(P)
    assert D.shape == (P + 1, P + 1)
    assert np.max(np.abs(D @ np.ones(P + 1))) < 1e-12
    for k in range(1, P + 1):
        assert np.max(np.abs(D @ pts**k - k * pts ** (k - 1))) < 1e-10
    # known P=2 table on [0,1]: nodes 0, 1/2, 1
    if P == 2:
        assert np.allclose(D, [[-3, 4, -1], [-1, 0, 1], [1, -4, 3]], atol=1e-14)


def test_quadrature_degree_map_matches_reference():
    # numba-cpu/time_operators.py:35-45
    gll = pkg("gll")
    assert gll.QUADRATURE_DEGREE == {2: 3, 3: 4, 4: 6, 5: 8, 6: 10, 7: 12, 8: 14, 9: 16, 10: 18}
