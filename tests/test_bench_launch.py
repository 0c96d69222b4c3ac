"""bench.py as the driver starts it (missing #1): ``python bench.py --gpus N`` with no
launcher must start its own N ranks before touching the GPU, relay ONE JSON line and propagate
failures; the torch.distributed.run form must keep working.  On CPU the ranks rehearse with
``--dry-run`` (gloo; partition + halo plan + all-to-all-v with the real counts; nothing measured);
on the GPU box the N = 1 line is run through the N > 1 code path (FUS_BENCH_FORCE_DIST=1: HaloApply
+ the RCCL communicator of libfusgpu.so / of torch)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 8])
def test_self_spawn_dry_run(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--steps", "2", "--warmup", "1", "--degree", "2",
                        "--cells", "2"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == n and out["ranks"] == n and out["dry_run"] is True and out["valid"] is False
    assert out["halo_ok"] is True
    assert out["scaling"] == "weak" and out["steps"] == 2 and out["warmup"] == 1


def test_torchrun_form_dry_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--degree", "2", "--cells", "2"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["halo_ok"] is True


def test_failed_rank_fails_the_launch():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--degree", "2", "--cells", "2"],
                       env=_env(FUS_BENCH_TEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_hung_rank_ends_the_launch():
    """A rank that never arrives at a collective: its watchdog leaves with a non-zero code, the launcher
    stops the ranks waiting for it, nothing is printed."""
    import time

    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--degree", "2", "--cells", "2"],
                       env=_env(FUS_BENCH_TEST_HANG_RANK="1", FUS_BENCH_WATCHDOG_S="20"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and time.time() - t0 < 300
    assert "watchdog" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_gpus_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"], env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "launcher started 2 ranks" in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("halo", ["peer", "native", "torch"])
def test_distributed_code_path_on_one_gpu(halo):
    """The N > 1 code path (HaloApply, three cell sub-ranges, halo begin/end, communicator bring-up) in a
    1-rank world on the GPU box, through bench.py itself."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "5", "--warmup", "2", "--cells", "16", "--no-cpu-baseline",
                        "--halo", halo], env=_env(FUS_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    cfg = out["config"]
    assert out["n_gpus"] == 1 and cfg["ranks"] == 1 and cfg["halo"] == "overlapped"
    assert cfg["halo_exposed_ms"] is not None and out["value"] > 0
    assert cfg["halo_split_cost_ms"] is not None and cfg["halo_exchange_exposed_ms"] is not None
    assert ("libfusgpu" in cfg["halo_transport"]) == (halo in ("peer", "native")) and ("PEER" in cfg["halo_transport"]) == (halo == "peer")
    assert cfg["halo_schedule"] == ("concurrent" if halo == "peer" else "split")
    tried = cfg["halo_transports_tried"]
    assert [t["transport"] for t in tried] == [halo] and tried[0]["result"] == "ok" and tried[0]["bring_up_failed_on_ranks"] == []
    assert tried[0]["check_failed_on_ranks"] == [] and (tried[0]["arena_memory_by_rank"] is not None) == (halo == "peer")
    fc = cfg["first_contact"]  # who sits where, before anything is exchanged
    assert len(fc["ranks"]) == 1 and fc["ranks"][0]["rank"] == 0 and ":" in fc["ranks"][0]["pci_bus_id"] and fc["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "first contact: rank 0 ->" in r.stderr and "CHOSEN" in r.stderr
    assert out["check"]["ok"] is True and out["check"]["rel_l2"] <= 1e-12 and cfg["check"]["ok"] is True  # N > 1 form of the result check
    assert cfg["lib_built_from_tree"] is True
    assert cfg["halo_check"]["ok"] is True and cfg["halo_check"]["forward_max_abs_err"] == 0.0
    assert out["roofline"]["kernel_ms"] > 0 and cfg["lib_sha"]


@pytest.mark.gpu
def test_torchrun_form_rehearsal_on_one_gpu():
    """The driver's own launch form (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2) with real
    processes and real HIP kernels sharing the one GPU: PEER transport across the processes, halo check passing."""
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--cells", "10", "--no-cpu-baseline"],
                       env=_env(FUS_BENCH_REHEARSAL="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["valid"] is False and "PEER" in cfg["halo_transport"] and cfg["halo_check"]["ok"] is True
    assert cfg["halo_schedule"] == "concurrent" and cfg["halo_check"]["device_wait_timeouts"] == 0


@pytest.mark.gpu
def test_transport_fallback_after_a_failed_halo_check():
    """A transport whose exchanges fail the run's own check is torn down (halo objects, communicator) and the next one
    is brought up and checked: peer rejected -> native (RCCL) used; peer and native rejected -> torch; the line says
    what was tried.  (The rejection is injected: FUS_BENCH_TEST_REJECT.)"""
    for reject, used in (("peer", "ncclSend"), ("peer,native", "all_to_all_single")):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "3", "--warmup", "1", "--cells", "12", "--no-cpu-baseline"],
                           env=_env(FUS_BENCH_FORCE_DIST="1", FUS_BENCH_TEST_REJECT=reject), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        cfg = _one_json_line(r.stdout)["config"]
        assert used in cfg["halo_transport"] and cfg["halo_check"]["ok"] is True
        tried = cfg["halo_transports_tried"]
        # (a PEER transport rejected ON DATA is tried once more in its fenced form, then with its arenas in fine-grained memory, before
        # RCCL gets its turn)
        assert [t["transport"] for t in tried] == ["peer", "peer:fenced", "peer:finegrained", "native", "torch"][: len(reject.split(",")) + 3]
        assert tried[1]["arena_memory_by_rank"] == ["uncached, fenced"] and tried[2]["arena_memory_by_rank"] == ["fine-grained"]
        assert all("rejected" in t["result"] for t in tried[:-1]) and tried[-1]["result"] == "ok"
    # the middle rung of the ladder: only the fence-free PEER protocol fails its check -> the SAME arenas and kernels
    # with a system-scope release before / acquire after every flag are brought up, checked and used; RCCL is never started
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "3", "--warmup", "1", "--cells", "12", "--no-cpu-baseline"],
                       env=_env(FUS_BENCH_FORCE_DIST="1", FUS_BENCH_TEST_REJECT="=peer"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    cfg = out["config"]
    assert "PEER" in cfg["halo_transport"] and "FENCED" in cfg["halo_transport"] and cfg["halo_check"]["ok"] is True and out["check"]["ok"] is True
    assert [t["transport"] for t in cfg["halo_transports_tried"]] == ["peer", "peer:fenced"] and cfg["halo_transports_tried"][1]["result"] == "ok"
    assert cfg["halo_transports_tried"][1]["arena_memory_by_rank"] == ["uncached, fenced"]
    assert set(cfg["halo_compare"]["transports"]) == {"peer:fenced", "peer", "native"}  # all three rungs timed in the one run
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "3", "--warmup", "1", "--cells", "12", "--no-cpu-baseline"],
                       env=_env(FUS_BENCH_FORCE_DIST="1", FUS_BENCH_TEST_REJECT="peer,native,torch"), capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "no halo transport passed" in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("dist_path", [False, True])
def test_mass_mode_line(dist_path):
    """SURVEY 8d's second operator line (cell mass apply), alone and through the N > 1 code path."""
    r = subprocess.run([sys.executable, BENCH, "--mode", "mass", "--steps", "5", "--warmup", "2", "--cells", "16"],
                       env=_env(FUS_BENCH_FORCE_DIST="1" if dist_path else "0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    assert out["metric"] == "mass_apply_dof_per_s" and out["value"] > 0
    assert out["roofline"]["algorithmic_bytes_per_cell"] == 3044 and "mass" in out["roofline"]["kernel"]
    if not dist_path:
        assert out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,mode", [(2, "stiffness"), (2, "mass"), (2, "rk4"), (4, "stiffness"), (4, "westervelt")])
def test_multi_rank_rehearsal_on_one_gpu(n, mode):
    """bench.py's own N = 2 / N = 4 path end to end with real processes and real HIP kernels on the one-GPU box:
    self-spawned ranks share the GPU, torch.distributed over gloo carries the bootstrap only, the exchange is the
    PEER transport itself -- arenas mapped across the processes with HIP IPC handles, send / receive kernels
    (FUS_BENCH_REHEARSAL=1; the line is marked invalid: the ranks share one GPU).  Covers what the N = 1 forced-dist run cannot: real
    neighbours inside the timed loop, the max-over-ranks reduction, rank 0 printing for the whole job."""
    cmd = [sys.executable, BENCH, "--gpus", str(n), "--mode", mode, "--steps", "3", "--warmup", "1", "--cells", "10", "--no-cpu-baseline"]
    if mode == "westervelt":
        cmd += ["--degree", "6", "--cells", "4"]
    r = subprocess.run(cmd, env=_env(FUS_BENCH_REHEARSAL="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == n and out["valid"] is False and "rehearsal" in out and out["value"] > 0
    if mode in ("stiffness", "mass"):
        cfg = out["config"]
        gy = n // 2
        assert cfg["ranks"] == n and cfg["partition"] == f"2x{gy}x1 blocks" and cfg["halo"] == "overlapped"
        assert cfg["global_dofs"] == (4 * 20 + 1) * (4 * 10 * gy + 1) * 41 and cfg["cells_per_gpu"] == 1000
        # the run checks its own exchanges before timing anything (forward: exact copy; reverse: owned sums)
        assert cfg["halo_check"]["ok"] is True and cfg["halo_check"]["owned_sum_defect_over_sum_abs"] < 1e-9
        assert "PEER" in cfg["halo_transport"] and cfg["halo_check"]["device_wait_timeouts"] == 0
        if mode == "mass":  # the partitioned apply keeps the atomic-free kernel: HaloApply splits it by dof
            assert out["roofline"]["kernel"] == "fus::mass_gather_kernel"
        assert out["check"]["ok"] is True and out["check"]["rel_l2"] <= 1e-12  # every rank's owned dofs against the oracle, reverse-scattered
        fc = cfg["first_contact"]
        assert [r_["rank"] for r_ in fc["ranks"]] == list(range(n)) and len({r_["pid"] for r_ in fc["ranks"]}) == n and fc["rehearsal"] is True
        if mode == "stiffness":
            # the driver's fixed N > 1 command (no extra flags) harvests everything in ONE run -- the transports compared
            # (on by default) and, on the same partition and communicator, the partitioned mass apply, both fused RK4 steps and the Westervelt
            # P = 6 step (BASELINE config 5), each with ms, per-rank min / max, its exposed halo cost and its own halo check
            hc = cfg["halo_compare"]
            assert hc["chosen"] == "peer" and hc["transports"]["peer"]["ms_per_step_median"] > 0 and "rehearsal" in hc["not_compared"]["native"]
            assert hc["transports"]["peer:fenced"]["ms_per_step_median"] > 0 and hc["transports"]["peer:fenced"]["failed_waits_all_ranks"] == 0
            sec = out["roofline"]["secondary"]
            for k in ("westervelt_geom", "rk4_geom", "mass", "rk4"):
                v = sec[k]
                assert v["ms"] > 0 and len(v["rank_ms"]) == 2 and 0 < v["rank_ms"][0] <= v["rank_ms"][1], (k, v)
                assert v["halo_exposed_ms"] is not None and v["halo_ok"] is True, (k, v)
            assert sec["mass"]["chk"][1] is True and sec["mass"]["chk"][0] <= 1e-12  # owned dofs of all ranks against the oracle
            hv = out["harvest"]
            assert hv["westervelt_geom"]["degree"] == 6 and hv["westervelt_geom"]["cells_per_gpu"] == 7**3 and "formed in the cell kernel" in hv["westervelt_geom"]["geometry"]
            assert hv["rk4"]["geometry"] == "general per-quadrature-point G" and hv["mass"]["kernel"] == "fus::mass_gather_kernel"
            assert hv["rk4_geom"]["halo_check"]["forward_wrong_ghosts"] == 0 and hv["rk4_geom"]["local_ms"] > 0 and hv["seconds"] < 150
            assert "skipped" not in json.dumps(hv)
    else:  # the solver lines check the exchange they are about to use, too
        hc = out["config"]["halo_check"]
        assert hc["ok"] is True and hc["forward_wrong_ghosts"] == 0 and hc["reverse_sum"] == hc["global_ghosts"] > 0
        assert out["config"]["halo_schedule"] == "concurrent"


@pytest.mark.gpu
def test_hung_optional_phase_does_not_take_the_headline_with_it():
    """N > 1: the transport comparison and the harvest are optional next to the headline, which is measured and checked before them.  A rank
    that never reaches them (test hook) leaves the others inside a collective: every rank's timer ends the run, rank 0 prints the line it has --
    valid headline, its check, ``extras_timed_out`` -- and the exit code is 0."""
    import time

    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--cells", "8", "--no-cpu-baseline"],
                       env=_env(FUS_BENCH_REHEARSAL="1", FUS_BENCH_TEST_HANG_EXTRAS="1", FUS_BENCH_EXTRAS_TIMEOUT_S="25"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert time.time() - t0 < 300
    out = _one_json_line(r.stdout)
    assert out["extras_timed_out"]["phase"] == "halo_compare" and out["check"]["ok"] is True and out["value"] > 0
    assert out["config"]["halo_check"]["ok"] is True and "the line is emitted without the rest" in r.stderr


@pytest.mark.gpu
def test_halo_compare_one_run_times_every_transport():
    """``--halo-compare``: ONE run times the apply over every transport that comes up, in alternating rounds, and puts each one's
    exposed cost in ``config.halo_compare``.  1-rank world through the N > 1 code path: PEER and RCCL both come up; 2 rehearsal
    ranks sharing the GPU: RCCL cannot (two ranks on one device) and the line says so."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "4", "--warmup", "1", "--cells", "12", "--no-cpu-baseline", "--halo-compare"],
                       env=_env(FUS_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    hc = _one_json_line(r.stdout)["config"]["halo_compare"]
    assert hc["chosen"] == "peer" and set(hc["transports"]) == {"peer", "peer:fenced", "native"} and hc["rounds"] == 5
    for k, v in hc["transports"].items():
        assert v["ms_per_step_median"] > 0 and len(v["ms_per_step_rounds"]) == 5 and v["failed_waits_all_ranks"] == 0
        assert abs(v["exposed_ms"] - (v["ms_per_step_median"] - hc["one_launch_ms"])) < 1e-12
    assert hc["transports"]["native"]["max_rel_diff_vs_chosen"] < 1e-12 and "halo compare: peer:" in r.stderr
    assert hc["transports"]["peer:fenced"]["max_rel_diff_vs_chosen"] < 1e-12 and "FENCED" in hc["transports"]["peer:fenced"]["transport"]
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--cells", "8", "--no-cpu-baseline", "--halo-compare"],
                       env=_env(FUS_BENCH_REHEARSAL="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    hc = _one_json_line(r.stdout)["config"]["halo_compare"]
    assert set(hc["transports"]) == {"peer", "peer:fenced"} and "rehearsal" in hc["not_compared"]["native"]


@pytest.mark.gpu
def test_scatter_mode_line():
    """``bench.py --mode scatter``: the reference's third timing script (numba-cpu/time_scatterer.py:126-210) at N = 1 -- a rank
    that is its own neighbour with config-4-shaped messages, per transport and direction both protocols, the numpy oracle
    beside them."""
    r = subprocess.run([sys.executable, BENCH, "--mode", "scatter", "--steps", "10", "--cells", "12"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    assert out["metric"] == "scatter_forward_reverse_us" and out["higher_is_better"] is False
    tr = out["scatter"]["transports"]
    for kind in ("peer", "native", "torch"):
        assert kind in tr and "error" not in tr[kind], tr.get(kind)
        for d in ("scatter_forward", "scatter_reverse"):
            assert tr[kind][d]["us_per_call_stream"] > 0 and tr[kind][d]["us_per_call_sync_mean"] > 0
    assert tr["peer"]["scatter_forward"]["failed_waits"] == 0
    assert out["cpu_baseline"]["scatter_reverse"]["us_per_call_mean"] > 0


@pytest.mark.gpu
def test_default_line_carries_the_aux_entries():
    """The driver's default command at a small size: the headline line with every ``aux`` entry of DESIGN.md section 6 (mass, its
    cached-diagonal form, sustained applies, in-kernel geometry, the RK4 step with and without in-kernel geometry -- with its
    CPU oracle leg --, the Westervelt step of config 5's shape, scatter)."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "5", "--warmup", "2", "--cells", "12"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    aux = out["aux"]
    for k in ("mass", "mass_cached_diagonal", "sustained", "stiffness_in_kernel_geometry", "rk4_step", "rk4_step_in_kernel_geometry",
              "westervelt_step", "scatter"):
        assert aux.get(k) is not None, k
    assert aux["westervelt_step"]["config"]["degree"] == 6 and aux["westervelt_step"]["value"] > 0
    assert aux["mass"]["roofline"]["kernel"] == "fus::mass_gather_kernel" and aux["mass"]["roofline"]["atomic_kernel_ms"] > 0
    assert aux["sustained"]["applies"] >= 2000 and len(aux["sustained"]["window_ms_per_apply"]) == 10
    assert aux["rk4_step"]["cpu_baseline"]["value"] > 0 and aux["rk4_step"]["cpu_baseline"]["single_thread_value"] > 0
    assert aux["rk4_step"]["roofline"]["algorithmic_bytes_per_step"] > 0
    assert aux["stiffness_in_kernel_geometry"]["roofline"]["algorithmic_bytes_per_cell"] == 2100
    assert out["cpu_baseline"]["value"] > 0 and out["roofline"]["frac"] > 0
    # a result check bound to the timed region, the aux scalars inside a key the driver keeps, the halo proxy
    for ck in (out["check"], out["config"]["check"]):
        assert ck["ok"] is True and ck["rel_l2"] <= 1e-12 and ck["rel_max"] <= 1e-11 and abs(ck["sum_y"] - ck["sum_y_oracle"]) <= 1e-9 * ck["norm_y_oracle"]
    # every line the driver stores carries a check of what ITS OWN timed launches computed, against the oracle
    for k in ("mass", "mass_cached_diagonal", "stiffness_in_kernel_geometry"):
        assert aux[k]["check"]["ok"] is True and aux[k]["check"]["rel_l2"] <= 1e-12, (k, aux[k]["check"])
    assert aux["mass"]["check"]["static_detJ_rel_l2"] <= 1e-12 and aux["mass"]["check"]["atomic_rel_l2"] <= 1e-12
    for k in ("rk4_step", "rk4_step_in_kernel_geometry", "westervelt_step", "westervelt_step_in_kernel_geometry", "westervelt_step_single_gather"):
        assert aux[k]["check"]["ok"] is True and aux[k]["check"]["rel_l2"] <= 1e-11, (k, aux[k]["check"])
    sec = out["roofline"]["secondary"]
    assert len(json.dumps(sec)) <= 1536, len(json.dumps(sec))
    for k in ("mass", "geom", "rk4", "rk4_geom", "westervelt", "westervelt_geom"):
        assert sec[k]["chk"][1] is True, (k, sec[k])
    for k in ("mass", "geom", "rk4", "rk4_geom", "westervelt", "westervelt_geom", "sustained", "halo_proxy", "check"):
        assert k in sec, k
    assert sec["check"]["ok"] is True and sec["mass"]["ms"] > 0 and 0 < sec["rk4"]["frac"] < 1
    hp = aux["halo_proxy"]["transports"]
    assert hp["peer"]["failed_waits"] == 0 and hp["peer"]["single_launch_us"] > 0 and "exposed_pct" in hp["peer"]
    assert sec["halo_proxy"]["peer"]["pct"] == round(hp["peer"]["exposed_pct"], 1)
    # the three rungs of the transport ladder on the one-GPU proxy: fence-free PEER, its fenced form, RCCL
    assert set(hp) == {"peer", "peer:fenced", "native"} and hp["peer:fenced"]["failed_waits"] == 0 and "pct" in sec["halo_proxy"]["peer:fenced"]
    assert aux["westervelt_step_in_kernel_geometry"]["value"] > 0


@pytest.mark.gpu
def test_bench_result_check_fails_loudly(monkeypatch):
    """A wrong result must not produce a valid line: with the oracle's constants perturbed (test hook) the check fails, the line
    says ``valid: false`` and the exit code is non-zero."""
    env = dict(_env(), FUS_BENCH_TEST_BREAK_CHECK="1")
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--cells", "8", "--no-aux", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    out = _one_json_line(r.stdout)
    assert out["valid"] is False and out["check"]["ok"] is False and out["check"]["rel_l2"] > 1e-6
    assert "RESULT CHECK FAILED" in r.stderr


@pytest.mark.gpu
def test_time_scatterer_script():
    """fenicsx-fus-gpu_amd/time_scatterer.py: the reference's numba-cpu/time_scatterer.py protocol (50 / 10 timed calls, mean +/-
    std) -- a rank that is its own neighbour with config-4 messages, and the one-rank no-neighbour case the reference runs on a
    single MPI rank."""
    script = os.path.join(ROOT, "fenicsx-fus-gpu_amd", "time_scatterer.py")
    for extra in (["--self-neighbour"], []):
        r = subprocess.run([sys.executable, script, *extra], env=_env(), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("Elapsed time")]
        assert len(lines) == 2 and "scatter reverse" in lines[0] and "scatter forward" in lines[1]
        for l in lines:
            mean = float(l.split(":")[1].split("±")[0])
            assert 0 < mean < 1e-3, l  # microseconds, not milliseconds


@pytest.mark.gpu
def test_config4_topology_rehearsal_tool():
    """tools/rehearse_8_ranks.py at a small size: the 2x2x2 partition as 8 ranks on 4 real processes x 2 ranks over the PEER
    transport, with bench.py's halo checks (exact ghosts, owned sums = 1^T K x = 0, no failed device-side wait)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_8_ranks.py"), "--cells", "6", "--applies", "2"], env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = _one_json_line(r.stdout)
    assert out["ok"] is True and out["ranks"] == 8 and out["processes"] == 4 and out["failed_device_waits"] == 0
    assert out["forward_max_abs_err"] == 0.0 and out["owned_sum_defect_over_sum_abs"] < 1e-9
    assert sorted(out["rank0_ghosted_by"]) == [str(k) for k in range(1, 8)] and out["rank0_ghosted_by"]["7"] == 1
