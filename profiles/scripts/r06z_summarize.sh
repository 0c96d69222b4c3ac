#!/bin/bash
# in the build container, after gpurun merged gpurun_out/prof_r06z*: condense into profiles/r06z_* and profiles/traffic_latest.json
set -e
python profiles/summarize.py r06z stiffness_plan_kernel --headline > /dev/null
python profiles/summarize.py r06z_mass mass_gather_kernel --aux=mass > /dev/null
python profiles/summarize.py r06z_mass_static mass_gather_kernel > /dev/null
python profiles/summarize.py r06z_geom stiffness_plan_geom_kernel --aux=geom > /dev/null
python profiles/summarize.py r06z_rk4 stiffness_plan_kernel --aux=rk4_step > /dev/null
python profiles/summarize.py r06z_rk4_geom stiffness_plan_geom_kernel --aux=rk4_step_in_kernel_geometry > /dev/null
python profiles/summarize.py r06z_westervelt westervelt_cell_kernel --aux=westervelt_step > /dev/null
python profiles/summarize.py r06z_westervelt_geom westervelt_cell_geom_kernel --aux=westervelt_step_in_kernel_geometry > /dev/null
python profiles/summarize.py r06z_westervelt_geom_single_gather stiffness_plan_geom_kernel --aux=westervelt_step_in_kernel_geometry_single_gather > /dev/null
python - <<'PY'
import json
t = json.load(open("profiles/traffic_latest.json"))
print("headline", t["hbm_bytes_per_launch"], t["source"], t["kernel_src_sha"])
for k, v in t["aux"].items():
    print(k, v.get("hbm_bytes_per_launch", v.get("hbm_bytes_per_step")), v["source"])
PY
