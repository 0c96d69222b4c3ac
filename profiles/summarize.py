#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (from profiles/run_profile.sh) into tracked files:
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_counters.json        per-launch averages of the PMC passes for the stiffness kernel
  profiles/<tag>_bench.json           the bench line of the traced run
  profiles/traffic_latest.json        what bench.py reports as roofline.traffic
HBM traffic = 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads,
guides/MI355X_MICROARCH.md 'HBM') + WRITE_SIZE, both in KiB per launch."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag = sys.argv[1]
kernel_key = sys.argv[2] if len(sys.argv) > 2 else "stiffness"
out_tag = tag
for a_ in sys.argv[3:]:
    if a_.startswith("--as="):
        out_tag = a_[5:]  # several kernels of one profiled run: one summary file each
# traffic_latest.json (what bench.py replays as roofline.traffic of the HEADLINE line) is rewritten only by the pass
# over the headline kernel in the default mode
update_latest = "--headline" in sys.argv[3:]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
out = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
_tag_for_files = out_tag
counters = {}
meta = {}
for sub, name in (("pmc_fetch", "fetch"), ("pmc_write", "write"), ("pmc_tcc", "tcc"), ("pmc_sq", "sq")):
    p = os.path.join(src, sub, f"{name}_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if kernel_key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"]
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
    for k, v in agg.items():
        counters[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
stats = {}
for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))):
    if kernel_key in r["Name"]:
        stats = {"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])}
bench = json.load(open(os.path.join(src, "bench_trace.json")))
res = {"tag": tag, "kernel_key": kernel_key, "kernel_stats": stats, "dispatch": meta, "counters": counters,
       "lib_sha": bench.get("config", {}).get("lib_sha"), "bench_metric": bench.get("metric"), "bench_ms_per_step": bench.get("ms_per_step")}
# back-to-back launches overlap at their tails: besides the per-dispatch durations of --stats, take the
# longest run of consecutive dispatches of the kernel from the trace and divide its span by its length
tr = os.path.join(src, "trace", "trace_kernel_trace.csv")
if os.path.exists(tr):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr)) if kernel_key in r["Kernel_Name"]]
    rows.sort()
    best = (0, 0, 0)
    i = 0
    while i < len(rows):
        j = i
        while j + 1 < len(rows) and rows[j + 1][0] - rows[j][1] < 3000:  # next starts within 3 us of this one's end (or before it)
            j += 1
        if j - i + 1 > best[0]:
            best = (j - i + 1, rows[i][0], rows[j][1])
        i = j + 1
    if best[0] > 1:
        res["back_to_back_run"] = {"launches": best[0], "span_ns_per_launch": (best[2] - best[1]) / best[0],
                                   "mean_dispatch_ns_in_trace": sum(e - b for b, e in rows) / len(rows)}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    f, w = counters["FETCH_SIZE"]["mean_per_launch"], counters["WRITE_SIZE"]["mean_per_launch"]
    res["hbm_bytes_per_launch"] = (2 * f + w) * 1024
    res["fetch_bytes_corrected"] = 2 * f * 1024
    res["write_bytes"] = w * 1024
    bytes_model = {"stiffness_plan_kernel": lambda nc, P: nc * (6 * (P + 1) ** 3 * 8 + 4 * (P + 1) ** 3 + 3 * 8 * P ** 3 + 8),
                   "mass_plan_kernel": lambda nc, P: nc * ((P + 1) ** 3 * 8 + 4 * (P + 1) ** 3 + 3 * 8 * P ** 3 + 8),
                   "mass_gather_kernel": lambda nc, P: nc * ((P + 1) ** 3 * 8 + 4 * (P + 1) ** 3 + 3 * 8 * P ** 3 + 8)}
    ncell = bench["config"].get("cells_per_gpu") or bench.get("roofline", {}).get("cells_per_launch")
    P = bench["config"].get("degree")
    if "algorithmic_bytes_per_cell" in (bench.get("roofline") or {}) and ncell:
        res["algorithmic_bytes_per_launch"] = ncell * bench["roofline"]["algorithmic_bytes_per_cell"]
    for key, fn in bytes_model.items():  # a kernel inside a larger step (RK4): its own model
        if key in kernel_key and ncell and P and "algorithmic_bytes_per_launch" not in res:
            res["algorithmic_bytes_per_launch"] = fn(ncell, P)
    if "algorithmic_bytes_per_launch" in res:
        res["traffic_over_algorithmic"] = res["hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    if stats:
        res["hbm_gbs_from_counters"] = res["hbm_bytes_per_launch"] / stats["avg_ns"]
    latest_path = os.path.join(out, "traffic_latest.json")
    try:
        latest = json.load(open(latest_path))
    except Exception:
        latest = {}
    if update_latest:
        latest = {"P": bench["config"]["degree"], "ncell": ncell, "hbm_bytes_per_launch": res["hbm_bytes_per_launch"],
                  "source": f"profiles/{_tag_for_files}_counters.json", "lib_sha": res["lib_sha"],
                  "kernel_src_sha": bench.get("config", {}).get("kernel_src_sha"), "dtype": bench.get("dtype", "f64"),
                  "aux": latest.get("aux", {})}
        json.dump(latest, open(latest_path, "w"), indent=1)
    if "--aux=mass" in sys.argv[3:]:  # what bench.py replays as aux.mass.roofline.traffic (gated on the mass kernel's sources)
        sys.path.insert(0, ROOT)
        import bench as bench_py

        latest.setdefault("aux", {})["mass"] = {
            "P": bench["config"]["degree"], "ncell": ncell, "dtype": bench.get("dtype", "f64"), "hbm_bytes_per_launch": res["hbm_bytes_per_launch"],
            "source": f"profiles/{_tag_for_files}_counters.json", "lib_sha": res["lib_sha"],
            "kernel": "fus::mass_gather_kernel" if "gather" in kernel_key else "fus::mass_plan_kernel",
            "kernel_src_files": ["mass_gather.hpp", "vecops.hpp"] if "gather" in kernel_key else ["plan.hpp", "mass.hpp"],
            "kernel_src_sha": bench_py.kernel_src_sha(("mass_gather.hpp", "vecops.hpp") if "gather" in kernel_key else ("plan.hpp", "mass.hpp")),
            "atomic_requests_per_launch": counters.get("TCC_EA0_ATOMIC_sum", {}).get("mean_per_launch")}
        json.dump(latest, open(latest_path, "w"), indent=1)
for a_ in sys.argv[3:]:
    # --aux=rk4_step / --aux=rk4_step_in_kernel_geometry: HBM bytes of one fused RK4 step = sum over the step's kernels of
    # (mean per-launch bytes) x (launches per step: 4 of each), from the FETCH_SIZE / WRITE_SIZE passes of a --mode rk4 run
    if a_ == "--aux=geom":  # what bench.py replays as aux.stiffness_in_kernel_geometry.roofline.traffic
        sys.path.insert(0, ROOT)
        import bench as bench_py

        files = ("plan.hpp", "stiffness.hpp", "stiffness_plan.hpp", "stiffness_geom.hpp")
        latest_path = os.path.join(out, "traffic_latest.json")
        latest = json.load(open(latest_path))
        latest.setdefault("aux", {})["stiffness_in_kernel_geometry"] = {
            "P": bench["config"]["degree"], "ncell": bench["config"].get("cells_per_gpu"), "dtype": bench.get("dtype", "f64"),
            "hbm_bytes_per_launch": res.get("hbm_bytes_per_launch"), "source": f"profiles/{_tag_for_files}_counters.json", "lib_sha": res["lib_sha"],
            "kernel": "fus::stiffness_plan_geom_kernel", "kernel_src_files": list(files), "kernel_src_sha": bench_py.kernel_src_sha(files),
            "atomic_requests_per_launch": counters.get("TCC_EA0_ATOMIC_sum", {}).get("mean_per_launch")}
        json.dump(latest, open(latest_path, "w"), indent=1)
    if a_.startswith("--aux=rk4_step") or a_.startswith("--aux=westervelt_step"):
        key = a_[6:]
        per = {}
        for sub, name in (("pmc_fetch", "fetch"), ("pmc_write", "write")):
            pth = os.path.join(src, sub, f"{name}_counter_collection.csv")
            for r in csv.DictReader(open(pth)):
                for kn in ("stiffness_plan_geom_kernel", "stiffness_plan_kernel", "westervelt_cell_geom_kernel", "westervelt_cell_kernel",
                           "facet_terms_kernel", "rk4_stage_nl2_kernel", "rk4_stage_kernel"):
                    if kn + "<" in r["Kernel_Name"] or r["Kernel_Name"].split("(")[0].endswith(kn):
                        per.setdefault(kn, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                        break
        breakdown, total = {}, 0.0
        for kn, c in per.items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                b = (2 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) + sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])) * 1024
                breakdown[kn] = {"hbm_bytes_per_launch": b, "launches_profiled": len(c["FETCH_SIZE"])}
                total += 4 * b
        sys.path.insert(0, ROOT)
        import bench as bench_py

        geo = "geometry" in key
        files = ("plan.hpp", "stiffness.hpp", "stiffness_plan.hpp") + (("stiffness_geom.hpp",) if geo else ()) + ("mass.hpp", "rk4.hpp", "vecops.hpp")
        if key.startswith("westervelt_step"):
            files = files + ("westervelt.hpp",) + (("westervelt_geom.hpp",) if geo else ())
        latest_path = os.path.join(out, "traffic_latest.json")
        latest = json.load(open(latest_path))
        latest.setdefault("aux", {})[key] = {
            "P": bench["config"]["degree"], "ncell": bench["config"].get("cells_per_gpu"), "dtype": bench.get("dtype", "f64"),
            "hbm_bytes_per_step": total, "breakdown": {k: round(v["hbm_bytes_per_launch"]) for k, v in breakdown.items()},
            "source": f"profiles/{_tag_for_files}_counters.json", "lib_sha": res["lib_sha"], "kernel_src_files": list(files),
            "kernel_src_sha": bench_py.kernel_src_sha(files)}
        json.dump(latest, open(latest_path, "w"), indent=1)
        res["rk4_step"] = {"hbm_bytes_per_step": total, "breakdown": breakdown,
                           "algorithmic_bytes_per_step": (bench.get("roofline") or {}).get("algorithmic_bytes_per_step")}
# the K dispatches of the timed region, in order (is the first launch after the synchronise slower than the steady state?)
if os.path.exists(tr):
    seq = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr)) if kernel_key in r["Kernel_Name"])
    K = int(bench.get("steps") or 0)
    W = int(bench.get("warmup") or 0)
    if K and len(seq) >= W + K:
        region = seq[W:W + K]
        res["timed_region_dispatches"] = {"duration_us": [round((e - b) / 1e3, 1) for b, e in region],
                                          "start_to_start_us": [round((region[i + 1][0] - region[i][0]) / 1e3, 1) for i in range(K - 1)],
                                          "span_us_per_launch": (region[-1][1] - region[0][0]) / 1e3 / K}
if "TCC_EA0_ATOMIC_sum" in counters and stats:
    res["atomic_requests_per_s"] = counters["TCC_EA0_ATOMIC_sum"]["mean_per_launch"] / (stats["avg_ns"] * 1e-9)
if "SQ_LDS_BANK_CONFLICT" in counters and "SQ_LDS_IDX_ACTIVE" in counters:
    res["lds_bank_conflict_fraction"] = counters["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / max(counters["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"], 1.0)
if "SQ_WAVE_CYCLES" in counters and "SQ_WAIT_ANY" in counters:
    wc = counters["SQ_WAVE_CYCLES"]["mean_per_launch"]
    res["wave_cycle_shares"] = {k: counters[k]["mean_per_launch"] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in counters}
json.dump(res, open(os.path.join(out, f"{_tag_for_files}_counters.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
