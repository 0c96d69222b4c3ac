"""Bytes contracts, replay of the committed rocprofv3 --pmc passes as ``roofline.traffic``, and the
compact ``roofline.secondary`` the driver's record keeps verbatim."""
import json
import os

from .common import HBM_PEAK_GBS, ROOT, kernel_src_sha, lib_built_from_tree


def stiffness_bytes_per_cell(P, T):
    """Algorithmic HBM bytes per cell: G + dofmap + x once + y RMW + constant."""
    n = P + 1
    nd = n**3
    return 6 * nd * T + 4 * nd + T * P**3 + 2 * T * P**3 + T


def mass_bytes_per_cell(P, T):
    """Algorithmic HBM bytes per cell of the cell mass apply: detJ + dofmap + x once + y RMW + constant."""
    nd = (P + 1) ** 3
    return nd * T + 4 * nd + 3 * T * P**3 + T


def geom_bytes_per_cell(P, T):
    """Algorithmic HBM bytes per cell of the in-kernel-geometry apply: dofmap + x once + y RMW +
    constant + the cell's vertex ids (8 int32) + vertex coordinates, each vertex read once
    (3 T per cell asymptotically).  DESIGN.md 3.3."""
    nd = (P + 1) ** 3
    return 4 * nd + 3 * T * P**3 + T + 32 + 3 * T


def rk4_step_bytes(P, T, ncells, ndofs, nfacets_source, nfacets_absorbing, mode, affine, in_kernel_geometry, single_gather=False, lean=True):
    """Algorithmic HBM bytes of ONE fused RK4 step (4 stages), DESIGN.md 3.4 / 6:
    linear:      4 x [cell pass + facet terms] + the vector passes: 34 touches with the LEAN stage kinds 4-7 of csrc/rk4.hpp (7 + 10 + 12 + 5;
                 the default since round 6), 41 with kinds 2, 0, 0, 3 (9 + 12 + 12 + 8)
    Westervelt:  4 x [cell pass: stiffness part, two gathers unless c4 / c3 is uniform] + the vector passes of csrc/westervelt.hpp
                 rk4_stage_nl2_kernel, which also read un, ku, m0, w2, w5: 46 touches lean (9 + 13 + 15 + 9), 52 otherwise (11 + 15 + 15 + 11;
                 rounds 2-5 priced this pass at 4 x 15 = 60: 8 touches too many, corrected in round 6); single-gather form: + 1 per pass for w
                 (+ 1 for u0 in the lean last pass)
    cell pass per cell: G (or the 48-byte affine record, or vertex ids + coordinates) + dofmap + x once per gather +
    y read-modify-write + constants;  facet terms per facet: detJ + dofmap + y RMW (+ x for the absorbing set)."""
    n = P + 1
    nd = n**3
    if mode == "rk4":
        if affine:
            cell = 48 + 4 * nd + 3 * T * P**3 + T
        elif in_kernel_geometry:
            cell = geom_bytes_per_cell(P, T)
        else:
            cell = stiffness_bytes_per_cell(P, T)
        touches = 34 if lean else 41
    else:
        gathers = 1 if single_gather else 2
        geo = (32 + 3 * T) if in_kernel_geometry else 6 * nd * T
        cell = geo + 4 * nd + gathers * T * P**3 + 2 * T * P**3 + gathers * T
        touches = (46 if lean else 52) + ((5 if lean else 4) if single_gather else 0)
    facets = nfacets_source * n * n * (T + 4 + 2 * T) + nfacets_absorbing * n * n * (T + 4 + 3 * T)
    return {"cell_pass_bytes_per_cell": cell, "vector_touches_per_step": touches,
            "bytes_per_step": 4 * (ncells * cell + facets) + touches * T * ndofs}


def load_traffic(P, ncell, sha, dtype="f64"):
    """(per-launch HBM bytes, source) from the committed rocprofv3 PMC passes (profiles/), or
    (None, reason).  PMC counters cannot be read from inside the run, so this is a REPLAYED figure:
    it is reported only when the profiled library is the one loaded now (same hash), and the
    line names its source."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except Exception:
        return None, "no profiles/traffic_latest.json"
    if int(t.get("P", -1)) != P or int(t.get("ncell", -1)) != ncell or t.get("dtype", "f64") != dtype:
        return None, "profiled workload differs from this run"
    if t.get("lib_sha") == sha:
        return float(t["hbm_bytes_per_launch"]), f"replayed from {t.get('source')} (rocprofv3 --pmc, same library hash)"
    ks = kernel_src_sha()
    if ks is not None and t.get("kernel_src_sha") == ks and lib_built_from_tree():
        return float(t["hbm_bytes_per_launch"]), (f"replayed from {t.get('source')} (rocprofv3 --pmc; library {t.get('lib_sha')} then, {sha} now: "
                                                  f"same kernel sources and compile flags {ks}, the library differs elsewhere)")
    return None, f"profiled library {t.get('lib_sha')} is not the loaded one ({sha}) and the kernel's sources differ"


def aux_traffic(key, P, ncell, dtype):
    """(HBM bytes per launch / per step of the ``aux.<key>`` entry of profiles/traffic_latest.json, source) or (None, reason):
    a REPLAYED figure of separate rocprofv3 --pmc passes, reported only when the workload is the profiled one and every
    kernel source it names (and the compile flags) are the profiled ones -- the rule of the headline's ``roofline.traffic``."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            t = json.load(f).get("aux", {}).get(key)
    except Exception:
        return None, "no profiles/traffic_latest.json"
    if not t:
        return None, f"no PMC passes in profiles/traffic_latest.json (aux.{key})"
    if int(t.get("P", -1)) != P or int(t.get("ncell", -1)) != ncell or t.get("dtype", "f64") != dtype:
        return None, "profiled workload differs from this run"
    files = tuple(t.get("kernel_src_files", ()))
    if not files or t.get("kernel_src_sha") != kernel_src_sha(files) or not lib_built_from_tree():
        return None, "the kernel sources differ from the profiled ones"
    val = t.get("hbm_bytes_per_step", t.get("hbm_bytes_per_launch"))
    src = f"(2 FETCH_SIZE + WRITE_SIZE) x 1024 of separate rocprofv3 --pmc passes ({t.get('source')}"
    if t.get("breakdown"):
        src += f"; per launch: {t['breakdown']}, 4 launches of each per step"
    return float(val), src + "); same kernel sources and compile flags"


def rk4_step_traffic(P, ncell, dtype, in_kernel_geometry):
    """(HBM bytes per fused RK4 step from the committed per-kernel PMC passes, source) or (None, reason): the sum over the
    step's launches of each kernel's per-launch bytes, replayed only when every kernel's sources and the compile flags are
    the profiled ones (as the headline's traffic)."""
    key = "rk4_step_in_kernel_geometry" if in_kernel_geometry else "rk4_step"
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            t = json.load(f).get("aux", {}).get(key)
    except Exception:
        return None, "no profiles/traffic_latest.json"
    if not t:
        return None, f"no PMC passes of the step's kernels in profiles/traffic_latest.json (aux.{key})"
    if int(t.get("P", -1)) != P or int(t.get("ncell", -1)) != ncell or t.get("dtype", "f64") != dtype:
        return None, "profiled workload differs from this run"
    files = tuple(t.get("kernel_src_files", ()))
    if not files or t.get("kernel_src_sha") != kernel_src_sha(files) or not lib_built_from_tree():
        return None, "the step's kernel sources differ from the profiled ones"
    return float(t["hbm_bytes_per_step"]), (f"sum over the step's launches of the per-launch (2 FETCH_SIZE + WRITE_SIZE) x 1024 of separate rocprofv3 --pmc "
                                            f"passes ({t.get('source')}): {t.get('breakdown')}; same kernel sources and compile flags")


def secondary_summary(out):
    """The scalars of ``aux`` that matter, mirrored into ``roofline.secondary`` (<= 1 kB): the driver's record keeps
    ``config``, ``roofline`` and ``cpu_baseline`` verbatim and only the NAME of ``aux``."""
    aux = out.get("aux") or {}
    r3, r1 = (lambda v: None if v is None else round(float(v), 3)), (lambda v: None if v is None else round(float(v), 1))
    r4 = lambda v: None if v is None else round(float(v), 4)  # noqa: E731

    def e2(v):
        try:
            return float(f"{float(v):.1e}")
        except (TypeError, ValueError):
            return None
    sec = {}

    def line(key, name):
        a = aux.get(name)
        if not a:
            return
        rf = a.get("roofline") or {}
        alg = rf.get("algorithmic_bytes_per_step") or ((rf.get("algorithmic_bytes_per_cell") or 0) * (rf.get("cells_per_launch") or 0)) or rf.get("algorithmic_bytes_per_launch")
        tr = rf.get("traffic")
        sec[key] = {"ms": None if rf.get("kernel_ms") is None else round(float(rf["kernel_ms"]), 4), "frac": r3(rf.get("frac")),
                    "tr": r3(tr / alg) if (tr and alg) else None}
        ck = a.get("check")  # the line's own oracle check (round 6): rel l2 and verdict
        if isinstance(ck, dict):
            sec[key]["chk"] = [e2(ck.get("rel_l2")), bool(ck.get("ok"))]

    line("mass", "mass")
    mrf = (aux.get("mass") or {}).get("roofline") or {}
    if mrf.get("static_detJ_kernel_ms"):
        sec["mass_static"] = {"ms": round(float(mrf["static_detJ_kernel_ms"]), 4), "frac": r3(mrf.get("static_detJ_frac"))}
    line("mass_diag", "mass_cached_diagonal")
    line("geom", "stiffness_in_kernel_geometry")
    line("rk4", "rk4_step")
    line("rk4_geom", "rk4_step_in_kernel_geometry")
    line("westervelt", "westervelt_step")
    line("westervelt_geom", "westervelt_step_in_kernel_geometry")
    line("westervelt_1g", "westervelt_step_single_gather")
    su = aux.get("sustained")
    if su:
        sec["sustained"] = {"ms": round(float(su["ms_per_apply"]), 4), "frac": r3(su["frac_of_hbm_roofline"])}
    hp = (aux.get("halo_proxy") or {}).get("transports") or {}
    if hp:
        sec["halo_proxy"] = {k: ({"us": r1(v.get("exposed_us")), "pct": r1(v.get("exposed_pct"))} if "exposed_us" in v else {"error": True})
                             for k, v in hp.items()}
    sc = (aux.get("scatter") or {}).get("transports") or {}
    if "peer" in sc and "scatter_forward" in sc["peer"]:
        sec["scatter_peer_us"] = [r1(sc["peer"]["scatter_forward"]["us_per_call_sync_mean"]), r1(sc["peer"]["scatter_reverse"]["us_per_call_sync_mean"])]
    ck = out.get("check")
    if ck:
        sec["check"] = {"rel_l2": float(f"{ck['rel_l2']:.2e}"), "ok": ck["ok"]}
    # N > 1: the harvested lines (harvest.py) -- ms per step (max over ranks), per-rank device time [min, max], the exposed halo cost against
    # the same work without exchange, the line's own halo check (and the oracle check of the partitioned mass apply)
    hv = out.get("harvest")
    if isinstance(hv, dict):
        for name, v in hv.items():
            if not isinstance(v, dict):
                continue
            if "ms" not in v:
                sec[name] = {k: v[k] for k in ("skipped", "error") if k in v}
                continue
            sec[name] = {"ms": r4(v["ms"]), "rank_ms": [r4(t) for t in v.get("rank_ms") or []], "halo_exposed_ms": r4(v.get("halo_exposed_ms")),
                         "frac": r3(v.get("frac")), "halo_ok": bool((v.get("halo_check") or {}).get("ok"))}
            if isinstance(v.get("check"), dict):
                sec[name]["chk"] = [e2(v["check"].get("rel_l2")), bool(v["check"].get("ok"))]
        sec["harvest_s"] = r1(hv.get("seconds"))
    return sec
