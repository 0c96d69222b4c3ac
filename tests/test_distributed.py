"""N > 1 path: world_size 2 and 4 under gloo on CPU.  The partitioned
apply  (forward halo -> stiffness on local cells -> reverse halo)  must equal the
single-rank apply on the same global mesh, dof for dof -- the distributed
known-answer test of SURVEY 8c."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, build_problem, pkg, rel_l2, ref_field
from halo_cpu import global_cell_constants


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(mode, tmp_path, P, cells, grid, overlap):
    world = int(np.prod(grid))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), mode, str(tmp_path), str(P),
               *map(str, cells), *map(str, grid), str(int(overlap))]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]


def serial_reference(P, cells, oracle_c):
    pb = build_problem(P, cells, perturb=0.16, seed=3)
    mesh = pb["mesh"]
    cc = global_cell_constants(mesh)
    y = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], cc, y, pb["G"], mesh.dofmap)
    return pb["x"], y  # single rank: local index == lexicographic id


def check(res, x_ser, y_ser):
    seen = np.zeros(y_ser.size, dtype=int)
    for d in res:
        lex = d["lex_owned"]
        seen[lex] += 1
        assert rel_l2(d["y_owned"], y_ser[lex]) < 1e-13
        # forward scatter refreshed every ghost with its owner's value
        assert np.allclose(d["x_after_fwd"], x_ser[d["lex_all"]], rtol=0, atol=1e-12)
    assert np.all(seen == 1), "every global dof must be owned by exactly one rank"


@pytest.mark.parametrize("overlap", [1, 0], ids=["overlap", "sequential"])
@pytest.mark.parametrize("P,cells,grid", [(2, (4, 3, 2), (2, 1, 1)), (3, (4, 4, 2), (2, 2, 1)), (2, (4, 4, 4), (2, 2, 2))],
                         ids=["2ranks", "4ranks", "8ranks"])
def test_partitioned_apply_gloo_cpu(tmp_path, oracle_c, P, cells, grid, overlap):
    res = run_ranks("cpu", tmp_path, P, cells, grid, overlap)
    check(res, *serial_reference(P, cells, oracle_c))


@pytest.mark.gpu
def test_partitioned_apply_two_ranks_one_gpu(tmp_path, oracle_c):
    """HIP pack/unpack + planned stiffness kernels on sub-ranges, 2 ranks sharing cuda:0,
    gloo transport staged through the host (RCCL cannot connect two ranks of one device)."""
    P, cells, grid = 4, (4, 4, 4), (2, 1, 1)
    res = run_ranks("gpu", tmp_path, P, cells, grid, 1)
    check(res, *serial_reference(P, cells, oracle_c))


@pytest.mark.gpu
def test_rccl_world1_alltoallv():
    """RCCL transport of TorchComm with the bench's process-group options (world size 1)."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_worker.py")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [(2, 1, 1), (1, 2, 2)], ids=["2ranks", "4ranks"])
def test_partitioned_rk4_solver_on_one_gpu(tmp_path, oracle_c, grid):
    """Fused linear RK4 solver on 2 / 4 ranks (all on cuda:0, gloo transport staged through the
    host) against the single-rank oracle-side solver: forward scatters of u_n and v_n, facet
    terms on partition interfaces, reverse scatter, lumped mass assembly."""
    import rk4_oracle

    P, cells, L = 3, (4, 4, 4), 0.012
    res = run_ranks("gpu-solver", tmp_path, P, cells, grid, 1)
    boxmesh = pkg("boxmesh")
    serial = boxmesh.BoxMesh(P, cells, length=L)
    u_ref, _ = rk4_oracle.solve(serial, 8, float(res[0]["dt"]), oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    seen = np.zeros(u_ref.size, dtype=int)
    for d in res:
        seen[d["lex_owned"]] += 1
        assert rel_l2(d["u_owned"], u_ref[d["lex_owned"]]) < 1e-11
    assert np.all(seen == 1)


@pytest.mark.gpu
def test_partitioned_westervelt_solver_on_one_gpu(tmp_path, oracle_c):
    """Fused Westervelt stage on 2 ranks: three quantities cross the partition per stage (u_n and
    v_n forward, b and the solution-dependent lumped mass m reverse)."""
    import rk4_oracle

    P, cells, L = 3, (4, 3, 3), 0.012
    res = run_ranks("gpu-solver-nl", tmp_path, P, cells, (2, 1, 1), 1)
    boxmesh = pkg("boxmesh")
    serial = boxmesh.BoxMesh(P, cells, length=L)
    u_ref, _ = rk4_oracle.solve_westervelt(serial, 8, float(res[0]["dt"]), c0=1500.0, f0=0.5e6, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    for d in res:
        assert rel_l2(d["u_owned"], u_ref[d["lex_owned"]]) < 1e-11
