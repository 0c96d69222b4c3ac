"""The time loop against REFERENCE-HELD data: tests/golden/rk4_*.npz were produced by driving the
imported reference's own operators and scatter closures (numba-cpu/operators.py, scatterer.py) through the stage sequence
of its RK4 loop (cuda/demo_linear_box.py:487-566) -- tests/golden/generate_golden.py --only rk4.

  * CPU: the oracle-side loop (oracle/rk4_oracle.py, what every solver test compares with) reproduces them, for the
    source evaluated at the stage time (numba-cpu / C++ drivers) and at the step time (the CUDA demos' quirk);
  * GPU: the solver itself -- reference launch sequence and fused stages, one rank and two in-process ranks over both
    in-process transports -- reproduces them.
This pins the operator-composition half of the loop (which vector feeds which operator, accumulate-into, t vs tn,
the order of the updates); the source formula and the material constants are restated in the generator."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, pkg, rel_l2
from oracle import rk4_oracle


def _case(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    boxmesh = pkg("boxmesh")
    P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
    kw = dict(length=tuple(float(v) for v in d["lengths"]), perturb=float(d["perturb"]), seed=int(d["seed"]))
    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, **kw) for r in range(R)]
    serial = boxmesh.BoxMesh(P, shape, **kw)
    return d, meshes, serial


@pytest.mark.parametrize("source_time", ["tn", "t"])
def test_oracle_loop_reproduces_reference_driven_loop(source_time):
    d, meshes, serial = _case("rk4_P2_2x2x2_pert_1rank")
    u, v = rk4_oracle.solve(serial, int(d["nsteps"]), float(d["dt"]), c0=float(d["c0"]), rho0=float(d["rho0"]), f0=float(d["f0"]),
                            p0=float(d["p0"]), source_time=source_time)
    assert np.max(np.abs(d[f"ref_u_{source_time}_0"])) > 1e3
    assert rel_l2(u, d[f"ref_u_{source_time}_0"]) < 1e-12 and rel_l2(v, d[f"ref_v_{source_time}_0"]) < 1e-12
    # the two source conventions really differ (the test would not notice a t / tn slip otherwise)
    assert rel_l2(d["ref_u_t_0"], d["ref_u_tn_0"]) > 1e-3


def test_oracle_loop_reproduces_reference_driven_two_rank_loop():
    """The reference's scatter closures inside the loop (forward of u_n, v_n; reverse of b and of the lumped mass):
    the serial oracle loop on the whole box equals every rank's owned AND ghost entries."""
    d, meshes, serial = _case("rk4_P2_4x2x2_pert_2ranks")
    u, v = rk4_oracle.solve(serial, int(d["nsteps"]), float(d["dt"]), c0=float(d["c0"]), rho0=float(d["rho0"]), f0=float(d["f0"]),
                            p0=float(d["p0"]), source_time="tn")
    for r, m in enumerate(meshes):
        lex = m.global_lexicographic_ids()
        assert rel_l2(u[lex], d[f"ref_u_tn_{r}"]) < 1e-12 and rel_l2(v[lex], d[f"ref_v_tn_{r}"]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True], ids=["reference-sequence", "fused"])
@pytest.mark.parametrize("source_time", ["tn", "t"])
def test_gpu_solver_reproduces_reference_driven_loop(fused, source_time):
    import torch

    torch.cuda.set_device(0)
    ls = pkg("linear_solver")
    d, meshes, serial = _case("rk4_P2_2x2x2_pert_1rank")
    s = ls.LinearSpectral3D(serial, np.float64, speed_of_sound=float(d["c0"]), density=float(d["rho0"]), source_frequency=float(d["f0"]),
                            source_amplitude=float(d["p0"]), fused=fused, source_time=source_time)
    assert not s.affine
    s.init()
    nsteps, dt = int(d["nsteps"]), float(d["dt"])
    _, steps = s.rk4(0.0, 1.0, dt, max_steps=nsteps)
    assert steps == nsteps
    assert rel_l2(s.u_sol(), d[f"ref_u_{source_time}_0"]) < 1e-11 and rel_l2(s.v_sol(), d[f"ref_v_{source_time}_0"]) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["local", "peer"])
@pytest.mark.parametrize("fused", [False, True], ids=["reference-sequence", "fused"])
def test_gpu_partitioned_solver_reproduces_reference_driven_loop(fused, transport):
    """Two in-process ranks, the library's exchange inside every stage, against what the reference's own scatter
    closures produced inside the same loop."""
    import torch

    from test_solver_gpu import _lockstep

    torch.cuda.set_device(0)
    ls, scat, utils = pkg("linear_solver"), pkg("scatterer"), pkg("utils")
    d, meshes, serial = _case("rk4_P2_4x2x2_pert_2ranks")
    R = len(meshes)
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = 7300 + 2 * int(fused) + (transport == "peer")
    comms = [scat.NativeComm(local=(wid, R, r), transport="peer" if transport == "peer" else "rccl") for r in range(R)]
    solvers = [ls.LinearSpectral3D(meshes[r], np.float64, speed_of_sound=float(d["c0"]), density=float(d["rho0"]),
                                   source_frequency=float(d["f0"]), source_amplitude=float(d["p0"]), comm=comms[r], fused=fused,
                                   halo_plan=(od[r], gd[r]), defer_setup_exchange=True) for r in range(R)]
    _lockstep([s._setup for s in solvers])
    for s in solvers:
        s.init()
    nsteps, dt = int(d["nsteps"]), float(d["dt"])
    res = _lockstep([s.rk4_schedule(0.0, 1.0, dt, max_steps=nsteps) for s in solvers])
    torch.cuda.synchronize()
    assert all(r[1] == nsteps for r in res)
    for r, (m, s) in enumerate(zip(meshes, solvers)):
        assert rel_l2(s.u_sol(), d[f"ref_u_tn_{r}"][: m.nlocal]) < 1e-11
        assert rel_l2(s.v_sol(), d[f"ref_v_tn_{r}"][: m.nlocal]) < 1e-11


# ------------------------------------------------------------------------------------------- Westervelt loop
def _case_nl(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    boxmesh = pkg("boxmesh")
    P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
    lengths, amp = tuple(float(v) for v in d["lengths"]), float(d["bowl_amplitude"])
    L = lengths[0]

    def bowl(xg):
        out = xg.copy()
        yy, zz = xg[:, 1] / lengths[1] - 0.5, xg[:, 2] / lengths[2] - 0.5
        out[:, 0] = xg[:, 0] + amp * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
        return out

    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, length=lengths, warp=bowl) for r in range(R)]
    serial = boxmesh.BoxMesh(P, shape, length=lengths, warp=bowl)
    kw = dict(c0=float(d["c0"]), rho0=float(d["rho0"]), f0=float(d["f0"]), p0=float(d["p0"]), beta=float(d["beta"]), att_dB=float(d["att_dB"]))
    return d, meshes, serial, kw


@pytest.mark.parametrize("name", ["rk4nl_P2_2x2x2_bowl_1rank", "rk4nl_P2_4x2x2_bowl_2ranks", "rk4nl_P3_4x2x2_bowl_2ranks"])
def test_oracle_westervelt_loop_reproduces_reference_driven_loop(name):
    """tests/golden/rk4nl_*.npz: the reference's own operators and scatter closures driven through the stage sequence of
    cuda/demo_nonlinear_bowl.py:458-475,533-650 (``generate_golden.py --only rk4nl``); ``rk4_oracle.solve_westervelt`` --
    what the Westervelt solver tests compare with -- reproduces them on every rank's owned and ghost entries."""
    d, meshes, serial, kw = _case_nl(name)
    u, v = rk4_oracle.solve_westervelt(serial, int(d["nsteps"]), float(d["dt"]), **kw)
    assert np.max(np.abs(u)) > 1e3
    for r, m in enumerate(meshes):
        lex = m.global_lexicographic_ids()
        assert rel_l2(u[lex], d[f"ref_u_tn_{r}"]) < 1e-12 and rel_l2(v[lex], d[f"ref_v_tn_{r}"]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["reference-sequence", "fused", "fused-two-gather", "fused-in-kernel-geometry"])
def test_gpu_westervelt_solver_reproduces_reference_driven_loop(variant):
    import torch

    torch.cuda.set_device(0)
    nls = pkg("nonlinear_solver")
    d, meshes, serial, kw = _case_nl("rk4nl_P2_2x2x2_bowl_1rank")
    s = nls.WesterveltSpectral3D(serial, np.float64, speed_of_sound=kw["c0"], density=kw["rho0"], source_frequency=kw["f0"],
                                 source_amplitude=kw["p0"], nonlinear_coefficient=kw["beta"], attenuation_coefficient_dB=kw["att_dB"],
                                 fused=variant != "reference-sequence", in_kernel_geometry=variant.endswith("geometry"),
                                 uniform_ratio=False if variant == "fused-two-gather" else True)
    s.init()
    _, steps = s.rk4(0.0, 1.0, float(d["dt"]), max_steps=int(d["nsteps"]))
    assert steps == int(d["nsteps"])
    assert rel_l2(s.u_sol(), d["ref_u_tn_0"]) < 1e-11 and rel_l2(s.v_sol(), d["ref_v_tn_0"]) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("transport,name", [("local", "rk4nl_P2_4x2x2_bowl_2ranks"), ("peer", "rk4nl_P2_4x2x2_bowl_2ranks"),
                                            ("peer", "rk4nl_P3_4x2x2_bowl_2ranks")])
def test_gpu_partitioned_westervelt_solver_reproduces_reference_driven_loop(transport, name):
    """P = 3 (round 6): the fused stage's DEFAULT cell kernel from degree 3 on forms G in the kernel (westervelt_cell_geom_kernel) --
    here against reference-held data on two ranks."""
    import torch

    from test_solver_gpu import _lockstep

    torch.cuda.set_device(0)
    nls, scat, utils = pkg("nonlinear_solver"), pkg("scatterer"), pkg("utils")
    d, meshes, serial, kw = _case_nl(name)
    R = len(meshes)
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = 7400 + (transport == "peer") + 2 * (int(d["P"]) == 3)
    comms = [scat.NativeComm(local=(wid, R, r), transport="peer" if transport == "peer" else "rccl") for r in range(R)]
    solvers = [nls.WesterveltSpectral3D(meshes[r], np.float64, speed_of_sound=kw["c0"], density=kw["rho0"], source_frequency=kw["f0"],
                                        source_amplitude=kw["p0"], nonlinear_coefficient=kw["beta"], attenuation_coefficient_dB=kw["att_dB"],
                                        comm=comms[r], fused=True, halo_plan=(od[r], gd[r]), defer_setup_exchange=True) for r in range(R)]
    _lockstep([s._setup for s in solvers])
    for s in solvers:
        s.init()
    res = _lockstep([s.rk4_schedule(0.0, 1.0, float(d["dt"]), max_steps=int(d["nsteps"])) for s in solvers])
    torch.cuda.synchronize()
    assert all(r[1] == int(d["nsteps"]) for r in res)
    for r, (m, s) in enumerate(zip(meshes, solvers)):
        assert rel_l2(s.u_sol(), d[f"ref_u_tn_{r}"][: m.nlocal]) < 1e-11
        assert rel_l2(s.v_sol(), d[f"ref_v_tn_{r}"][: m.nlocal]) < 1e-11
