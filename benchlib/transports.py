"""Halo transports of a run: candidates and bring-up, the first-contact report, the in-run PEER / RCCL comparison, the one-GPU
halo proxy and the stand-alone scatter timing (numba-cpu/time_scatterer.py)."""
import os
import time

import numpy as np

from .common import _free_port, coll_device, log, rehearsal
from .launch import _StagedGlooComm


_VARIANT_ENV = {"finegrained": ("FUS_IPC_MEMORY", "finegrained"), "fenced": ("FUS_IPC_FENCED", "1")}
_ENV_AT_START = {}


def apply_variant_env(kind):
    """Point the environment at one variant of a transport (``"peer:fenced"``, ``"peer:finegrained"``; a plain kind restores what the
    process started with): the PEER transport reads FUS_IPC_MEMORY / FUS_IPC_FENCED when a halo object is CREATED, so the variant
    holds for every closure built until the next call -- the chosen transport's setting stays for the rest of the run (the harvest's
    solvers build their closures on the same communicator)."""
    for name in ("FUS_IPC_MEMORY", "FUS_IPC_FENCED"):
        if name not in _ENV_AT_START:
            _ENV_AT_START[name] = os.environ.get(name)
        if _ENV_AT_START[name] is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = _ENV_AT_START[name]
    variant = kind.partition(":")[2]
    if variant:
        name, value = _VARIANT_ENV[variant]
        os.environ[name] = value


def transport_text(kind):
    base, _, variant = kind.partition(":")
    extra = {"": "", "finegrained": " [arenas in fine-grained memory]",
             "fenced": " [FENCED: system-scope release before / acquire after every flag, one lane per workgroup]"}[variant]
    return TRANSPORT_TEXT[base] + extra


TRANSPORT_TEXT = {
    "peer": "libfusgpu.so PEER transport: peer-mapped arenas (HIP IPC), send / receive kernels with sequence flags, no RCCL kernel",
    "native": "libfusgpu.so: grouped ncclSend/ncclRecv on a library-owned stream",
    "torch": "torch.distributed.all_to_all_single (RCCL)",
}


def transport_candidates(args):
    """Transports this run may use, in order of preference; the first one that comes up on EVERY rank and passes the
    run's own halo check is used (decided collectively, recorded in the line)."""
    # "peer:finegrained": the PEER transport once more with its receive arenas in fine-grained instead of uncached device memory
    # (FUS_IPC_MEMORY) -- export / open of UNCACHED memory between two different devices has never run on this pool (one GPU per box),
    # and falling straight back to RCCL would cost 22 % per apply where another memory kind might cost nothing
    # "peer:fenced" (round 6): the same arenas and kernels, conservatively ordered (csrc/halo_ipc.hpp ipc_release_system) -- the middle rung,
    # tried only after a PEER transport that CAME UP failed the halo check on data (a bring-up failure is not an ordering problem)
    return {"peer": ["peer", "peer:fenced", "peer:finegrained", "native", "torch"], "native": ["native", "torch"], "torch": ["torch"]}[args.halo]


def make_comm(kind, scat, world, device):
    """One candidate transport, created on all ranks or on none: returns (comm, None) or (None, reason)."""
    import torch
    import torch.distributed as dist

    if kind == "torch":
        return (_StagedGlooComm(scat.TorchComm()) if rehearsal() else scat.TorchComm()), None
    if kind == "native" and rehearsal():
        return None, "RCCL refuses two ranks on one device (rehearsal)"
    comm, err = None, None
    try:
        comm = scat.NativeComm(transport="peer" if kind == "peer" else "rccl")
    except Exception as e:  # noqa: BLE001  (NativeComm itself fails on all ranks or on none; this is the belt to its braces)
        err = repr(e)
        log(f"{kind} communicator failed on rank {dist.get_rank() if dist.is_initialized() else 0}: {err}")
    ok = 1.0 if err is None else 0.0
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = float(flag.item())
    if ok == 1.0:
        return comm, None
    if comm is not None:
        comm.close()
    return None, err or "failed on another rank"


def first_comm(args, scat, world, device):
    """(comm, kind) of the first candidate transport that comes up on every rank."""
    for kind in transport_candidates(args):
        if ":" in kind:  # arena-memory variants of a transport are retried by the apply modes' own halo check, not here
            continue
        comm, why = make_comm(kind, scat, world, device)
        if comm is not None:
            return comm, kind
        log(f"halo transport {kind!r} not available ({why}); trying the next one")
    raise SystemExit("no halo transport came up")


def gather_verdicts(rank, world, mine):
    """Every rank's view of one bring-up / check step, so that the line and the log name the rank that failed."""
    import torch.distributed as dist

    every = [None] * world
    dist.all_gather_object(every, dict(mine, rank=rank))
    return every


def first_contact_report(rank, world, device):
    """N > 1, before anything is exchanged: what the transports will find, one block on rank 0's stderr and the same facts in the
    line (``config.first_contact``) -- the devices by PCI bus id (ordinals are process-local), which of the devices visible to a
    rank it can reach peer-to-peer, the IPC mode of the environment.  The first run on a real 8-GPU node must explain itself."""
    import torch
    import torch.distributed as dist

    def pci(d):
        p = torch.cuda.get_device_properties(d)
        try:
            return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except AttributeError:
            return f"ordinal-{d}"

    me = {"rank": rank, "pid": os.getpid(), "device_ordinal": device.index, "pci_bus_id": pci(device.index),
          "name": torch.cuda.get_device_properties(device.index).name, "visible_devices": torch.cuda.device_count(), "peer_access": {}}
    for d in range(torch.cuda.device_count()):
        if d != device.index:
            try:
                me["peer_access"][pci(d)] = bool(torch.cuda.can_device_access_peer(device.index, d))
            except Exception as e:  # noqa: BLE001
                me["peer_access"][pci(d)] = f"error: {e!r}"
    every = [None] * world
    dist.all_gather_object(every, me)
    report = {"ranks": every, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "FUS_IPC_MEMORY": os.environ.get("FUS_IPC_MEMORY"),
              "rehearsal": rehearsal()}
    if rank == 0:
        log("first contact: " + ", ".join(f"rank {r['rank']} -> {r['pci_bus_id']} ({r['name']}, ordinal {r['device_ordinal']} of {r['visible_devices']})" for r in every))
        shared = len({r["pci_bus_id"] for r in every}) < world
        if shared:
            log("first contact: several ranks share one device (rehearsal): peer access is not the question here")
        for r in every:
            no = [k for k, v in r["peer_access"].items() if v is not True]
            log(f"first contact: rank {r['rank']} peer access to the other visible devices: " + ("all" if not no else f"NOT to {no}") + f" ({len(r['peer_access'])} checked)")
        log(f"first contact: HSA_ENABLE_IPC_MODE_LEGACY={report['HSA_ENABLE_IPC_MODE_LEGACY']!r} (must be '0': dmabuf IPC), FUS_IPC_MEMORY={report['FUS_IPC_MEMORY']!r}")
    return report


def compare_transports(args, rank, world, device, scat, mesh, op, dt, x_d, cc_d, y_d, G_d, dm_d, chosen_kind, chosen_halo, kern_ms, rounds=5):
    """``--halo-compare``: the apply over every transport that comes up (the chosen one + the other of peer / native), timed in
    ALTERNATING rounds of K steps in this one process (barrier + synchronise on both sides, max over ranks), each one's exposed cost
    against ONE launch over all local cells; the result of one apply through each extra transport is compared with the chosen
    transport's.  One ``bench.py --gpus 8 --halo-compare`` run answers "PEER or RCCL, and by how much" (VERDICT r4 item 7)."""
    import torch
    import torch.distributed as dist

    ops_mod = __import__("fusgpu_loader").submodule("operators")
    halos, comms, notes = {chosen_kind: chosen_halo}, {}, {}
    for kind in ("peer", "peer:fenced", "native"):  # the three rungs of the ladder: fence-free PEER, its fenced form, RCCL
        if kind in halos:
            continue
        if kind.startswith("peer") and chosen_kind.partition(":")[0] != "peer":
            notes[kind] = "the run's own ladder rejected the PEER transport"
            continue
        apply_variant_env(kind)
        comm, why = make_comm(kind.partition(":")[0], scat, world, device)
        if comm is None:
            notes[kind] = f"did not come up: {why}"
            continue
        err, h = None, None
        try:
            h = scat.HaloApply(mesh, op, comm, dt, overlap=os.environ.get("FUS_HALO_OVERLAP", "1") != "0")
            h.prepare(x_d, cc_d, G_d, dm_d)
            h.apply(x_d, cc_d, y_d, G_d, dm_d)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        every = gather_verdicts(rank, world, {"error": err})
        if any(v["error"] for v in every):
            notes[kind] = f"bring-up failed on rank(s) {[v['rank'] for v in every if v['error']]}: {[v['error'] for v in every if v['error']][:2]}"
            try:
                dist.barrier()
                if h is not None:
                    h.fwd.close(), h.rev.close()
                comm.close()
            except Exception:  # noqa: BLE001
                pass
            continue
        halos[kind], comms[kind] = h, comm
    apply_variant_env(chosen_kind)  # what the rest of the run builds (harvest) is the chosen transport's variant again
    # one apply through each transport into a zeroed y: the extra transports against the chosen one
    ref, diffs = None, {}
    for kind, h in halos.items():
        ops_mod.fill(0.0, y_d)
        h.apply(x_d, cc_d, y_d, G_d, dm_d)
        torch.cuda.synchronize()
        owned = y_d[: mesh.nlocal].clone()
        if ref is None:
            ref = owned
        else:
            t = torch.stack([(owned - ref).abs().max() if owned.numel() else owned.new_zeros(()), ref.abs().max() if ref.numel() else ref.new_zeros(())]).double().to(coll_device(device))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            diffs[kind] = float(t[0].item()) / max(float(t[1].item()), 1e-300)
    times = {k: [] for k in halos}
    for _ in range(rounds):
        for kind, h in halos.items():
            h.apply(x_d, cc_d, y_d, G_d, dm_d)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                h.apply(x_d, cc_d, y_d, G_d, dm_d)
            torch.cuda.synchronize()
            dist.barrier()
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_device(device))
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            times[kind].append(float(el.item()) / args.steps * 1e3)
    out = {"rounds": rounds, "steps_per_round": args.steps, "one_launch_ms": kern_ms, "chosen": chosen_kind, "transports": {}, "not_compared": notes or None}
    for kind, h in halos.items():
        med = float(np.median(times[kind]))
        late = torch.tensor([float(h.health())], dtype=torch.float64, device=coll_device(device))
        dist.all_reduce(late)
        out["transports"][kind] = {"transport": transport_text(kind), "schedule": h.schedule_kind, "ms_per_step_median": med,
                                   "ms_per_step_rounds": times[kind], "exposed_ms": med - kern_ms, "exposed_frac": (med - kern_ms) / kern_ms,
                                   "failed_waits_all_ranks": int(late.item()), "max_rel_diff_vs_chosen": diffs.get(kind)}
    if rank == 0:
        log("halo compare: " + "; ".join(f"{k}: {v['ms_per_step_median']:.4f} ms/step = one launch {v['exposed_ms'] * 1e3:+.1f} us ({100 * v['exposed_frac']:+.1f} %)"
                                         for k, v in out["transports"].items()) + (f"; not compared: {notes}" if notes else ""))
    try:
        torch.cuda.synchronize()
        dist.barrier()
        for kind, comm in comms.items():
            halos[kind].fwd.close(), halos[kind].rev.close()
            comm.close()
    except Exception as e:  # noqa: BLE001
        log(f"rank {rank}: halo compare teardown: {e!r}")
    return out


def config4_self_plan(n1, permuted=False, seed=0):
    """``utils.config4_self_plan`` of the package (the halo plan of one config-4 rank that is its own neighbour)."""
    import fusgpu_loader

    return fusgpu_loader.submodule("utils").config4_self_plan(n1, permuted, seed)


def measure_scatter(device, dtype_np, kinds=("peer", "native", "torch"), reps=100, P=4, cells=54):
    """The reference's third timing script (numba-cpu/time_scatterer.py:126-210: scatter_reverse / scatter_forward alone, one
    call at a time between two clock reads) at N = 1: a rank that is its own neighbour with config-4 message sizes.  Per
    transport and direction: ``us_per_call_sync`` = mean / std of host clock around call + device synchronise (the reference's
    protocol -- its closures block), ``us_per_call_stream`` = ``reps`` calls back to back between one HIP-event pair."""
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    scat = fusgpu_loader.submodule("scatterer")
    n1 = P * cells + 1
    od, gd, N = config4_self_plan(n1)
    ng = int(od[1][0])
    tdt = torch.float64 if np.dtype(dtype_np) == np.float64 else torch.float32
    buf = torch.randn(N + ng, dtype=tdt, device=device)
    out = {"workload": f"one rank, its own neighbour, config-4 messages ({ng} elements = {ng * np.dtype(dtype_np).itemsize / 1e6:.2f} MB per direction: "
                       f"3 faces of {n1 * n1}, 3 edges of {n1}, 1 corner), vector of {N + ng} dofs", "reps": reps, "transports": {}}
    own_pg = False
    for kind in kinds:
        comm = None
        try:
            if kind == "torch":
                if not dist.is_initialized():
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", str(_free_port()))
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
                    own_pg = True
                comm = scat.TorchComm()
            else:
                comm = scat.NativeComm(transport="peer" if kind == "peer" else "rccl")
            row = {}
            for dname, mk in (("scatter_forward", scat.scatter_forward), ("scatter_reverse", scat.scatter_reverse)):
                sc = mk(comm, od, gd, N, dtype_np)
                for _ in range(3):
                    sc(buf)
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    sc(buf)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    sc(buf)
                e1.record()
                torch.cuda.synchronize()
                row[dname] = {"us_per_call_sync_mean": float(np.mean(ts)) * 1e6, "us_per_call_sync_std": float(np.std(ts)) * 1e6,
                              "us_per_call_sync_min": float(np.min(ts)) * 1e6, "us_per_call_stream": e0.elapsed_time(e1) / reps * 1e3}
                if hasattr(sc, "status"):
                    row[dname]["failed_waits"] = int(sc.status().get("failures", 0))
                if hasattr(sc, "close"):
                    sc.close()
                buf.normal_()  # reverse adds: keep the values bounded
            row["transport"] = TRANSPORT_TEXT[kind]
            out["transports"][kind] = row
        except Exception as e:  # noqa: BLE001
            out["transports"][kind] = {"error": repr(e)}
            log(f"scatter timing, transport {kind!r}: {e!r}")
        finally:
            if comm is not None and hasattr(comm, "close"):
                try:
                    comm.close()
                except Exception:  # noqa: BLE001
                    pass
    if own_pg:
        dist.destroy_process_group()
    # CPU beside it: the oracle's numpy restatement of the reference's closures (pack, copy, unpack), same plan
    try:
        from oracle import oracle_np

        h = np.random.default_rng(0).standard_normal(N + ng)
        cpu = {}
        for dname, fn in (("scatter_forward", oracle_np.scatter_forward_all), ("scatter_reverse", oracle_np.scatter_reverse_all)):
            fn([h], [od], [gd], [N])
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                fn([h], [od], [gd], [N])
                ts.append(time.perf_counter() - t0)
            cpu[dname] = {"us_per_call_mean": float(np.mean(ts)) * 1e6, "us_per_call_std": float(np.std(ts)) * 1e6}
        out["cpu_baseline"] = dict(cpu, kind="port", cores=1, impl="oracle/oracle_np.py (numba-cpu/scatterer.py:78-207 restated, no MPI: one rank)",
                                   sample="the same plan, 10 calls per direction")
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline"] = None
        log(f"scatter cpu leg failed: {e!r}")
    return out


def measure_halo_proxy(op, mesh, cc_d, G_d, dm_d, y_d, device, dt_np, P, cells, kinds=("peer", "peer:fenced", "native"), rounds=5, reps=40):
    """north_star's "< 5 % halo-exchange overhead" on the only proxy a one-GPU box has (tools/overlap_probe.py --paired, the
    measurement DESIGN 4.3 quotes): ONE rank that is its own neighbour with the messages of a config-4 rank (3 faces + 3 edges
    + 1 corner of a 54^3-cell P = 4 block: 1.14 MB per direction; it sends AND receives every message -- the upper bound of what
    a rank of a 2x2x2 partition does), 8 590 boundary cells first.  ``rounds`` alternating rounds of ``reps`` applies each of
    (single launch over all cells | HaloApply's own launch schedule without exchange | the same with both exchanges);
    medians, and medians of the per-round differences.  Per transport: PEER (the default) and RCCL grouped send / recv."""
    import torch

    import fusgpu_loader

    scat = fusgpu_loader.submodule("scatterer")
    n1 = P * cells + 1
    od, gd, N = config4_self_plan(n1)
    ng = int(od[1][0])
    if N + ng != mesh.ndofs:
        raise ValueError("halo proxy: the self-neighbour plan is sized for the serial box")

    class _RankView:  # the attributes HaloApply reads from a mesh
        pass

    m = _RankView()
    nb = cells * cells + cells * (cells - 1) + (cells - 1) * (cells - 1)  # the cells on three faces of a cells^3 block ...
    nb = (nb + 9) // 10 * 10  # ... in whole batches of the plan (8 590 at config 4, as tools/overlap_probe.py)
    m.num_boundary_cells, m.ncells, m.nlocal, m.dofmap, m.index_map = nb, mesh.ncells, N, mesh.dofmap, None
    tdt = torch.float64 if np.dtype(dt_np) == np.float64 else torch.float32
    xg = torch.randn(mesh.ndofs, dtype=tdt, device=device)
    out = {"workload": f"one rank, its own neighbour, config-4 messages ({ng} elements per direction), {nb} boundary cells of {mesh.ncells}; "
                       f"{rounds} alternating rounds x {reps} applies, medians of per-round differences",
           "rounds": rounds, "reps": reps, "transports": {}}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for kind in kinds:
        comm = halo = None
        try:
            apply_variant_env(kind)
            comm = scat.NativeComm(transport="peer" if kind.startswith("peer") else "rccl")
            halo = scat.HaloApply(m, op, comm, dt_np, plan=(od, gd))
            halo.prepare(xg, cc_d, G_d, dm_d)
            fns = (("single", lambda: op(xg, cc_d, y_d, G_d, dm_d)),
                   ("schedule", lambda: halo.apply_no_exchange(xg, cc_d, y_d, G_d, dm_d)),
                   ("halo", lambda: halo.apply(xg, cc_d, y_d, G_d, dm_d)))
            res = {k: [] for k, _ in fns}
            for _ in range(rounds):
                for k, fn in fns:
                    res[k].append(timed(fn))
            a = {k: np.array(v) for k, v in res.items()}
            single = float(np.median(a["single"]))
            d_halo, d_split = float(np.median(a["halo"] - a["single"])), float(np.median(a["schedule"] - a["single"]))
            out["transports"][kind] = {
                "transport": transport_text(kind), "schedule": halo.schedule_kind, "lead_cells": halo.lead_cells,
                "single_launch_us": single, "schedule_without_exchange_us": float(np.median(a["schedule"])),
                "with_both_exchanges_us": float(np.median(a["halo"])), "exposed_us": d_halo, "exposed_pct": 100.0 * d_halo / single,
                "split_us": d_split, "exchanges_us": float(np.median(a["halo"] - a["schedule"])),
                "exposed_us_per_round": [float(v) for v in (a["halo"] - a["single"])], "failed_waits": int(halo.health())}
        except Exception as e:  # noqa: BLE001
            out["transports"][kind] = {"error": repr(e)}
            log(f"halo proxy, transport {kind!r}: {e!r}")
        finally:
            try:
                torch.cuda.synchronize()
                if halo is not None:
                    halo.fwd.close(), halo.rev.close()
                if comm is not None:
                    comm.close()
            except Exception:  # noqa: BLE001
                pass
    apply_variant_env("peer")  # back to what the process started with
    return out
