"""demo_linear_box caller (BASELINE config 3 shape, small size): the GPU RK4 solver -- both the
reference launch sequence and the fused stage -- against the oracle-side solver."""

import numpy as np
import pytest

from conftest import pkg, rel_l2
import rk4_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [False, True], ids=["reference-sequence", "fused"])
@pytest.mark.parametrize("source_time", ["tn", "t"])
def test_linear_box_pressure_field(oracle_c, fused, source_time):
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    P, N, L = 4, 6, 0.012
    mesh = boxmesh.BoxMesh(P, N, length=L)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    assert abs(h - np.sqrt(3) * L / N) < 1e-12
    dt, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    nsteps = 12
    solver = ls.LinearSpectral3D(mesh, np.float64, fused=fused, source_time=source_time)
    solver.init()
    t, steps = solver.rk4(0.0, tf, dt, max_steps=nsteps)
    assert steps == nsteps and abs(t - nsteps * dt) < 1e-15
    u_ref, v_ref = rk4_oracle.solve(mesh, nsteps, dt, source_time=source_time, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(solver.u_sol(), u_ref) < 1e-11
    assert rel_l2(solver.v_sol(), v_ref) < 1e-11


def test_fused_matches_reference_sequence_perturbed_mesh():
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    mesh = boxmesh.BoxMesh(3, (5, 4, 4), length=0.01, perturb=0.1, seed=2)
    h = ls.time_step_parameters(mesh, 3, 1500.0, 0.5e6, 0.01)
    dt, tf, _ = ls.snap_time_step(h, 3, 1500.0, 0.5e6, 0.01)
    res = []
    for fused in (False, True):
        s = ls.LinearSpectral3D(mesh, np.float64, fused=fused)
        s.init()
        s.rk4(0.0, tf, dt, max_steps=20)
        res.append((s.u_sol(), s.v_sol()))
    assert rel_l2(res[1][0], res[0][0]) < 1e-12 and rel_l2(res[1][1], res[0][1]) < 1e-12


def _bowl_warp(xg):
    """Smooth non-affine map (SURVEY 8d geometry iii): the x = 0 face bulges like a bowl."""
    L = 0.006
    y, z = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
    out = xg.copy()
    out[:, 0] = xg[:, 0] + 0.15 * L * (y * y + z * z) * (1.0 - xg[:, 0] / L)
    return out


@pytest.mark.parametrize("fused", [False, True], ids=["reference-sequence", "fused"])
@pytest.mark.parametrize("P,cells", [(6, (3, 2, 2)), (4, (4, 3, 3)), (2, (7, 5, 6))], ids=["P6", "P4", "P2"])
def test_westervelt_bowl_pressure_field(oracle_c, P, cells, fused):
    """BASELINE config 5 shape (Westervelt, curved trilinear cells, P = 6) at test size."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, nls = pkg("boxmesh"), pkg("linear_solver"), pkg("nonlinear_solver")
    L = 0.006
    mesh = boxmesh.BoxMesh(P, cells, length=L, warp=_bowl_warp)
    h = ls.time_step_parameters(mesh, P, 1480.0, 1.1e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1480.0, 1.1e6, L)
    nsteps = 10
    s = nls.WesterveltSpectral3D(mesh, np.float64, fused=fused)
    s.init()
    s.rk4(0.0, tf, dt, max_steps=nsteps)
    u_ref, v_ref = rk4_oracle.solve_westervelt(mesh, nsteps, dt, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(s.u_sol(), u_ref) < 1e-11
    assert rel_l2(s.v_sol(), v_ref) < 1e-11
