#!/bin/bash
# round 5, library after the preamble restructure: full-size rehearsals of bench.py's N > 1 path with REAL processes sharing the one GPU (PEER transport): 6 ranks (3x2x1 blocks),
# stiffness with --halo-compare and the N > 1 result check, the mass mode on the atomic-free kernel (row split), the default RK4 and
# Westervelt steps; config 4's 8-rank topology at full size on 4 processes x 2 ranks
O=gpurun_out/r05w
mkdir -p $O
export FUS_BENCH_REHEARSAL=1 FUS_IPC_SPIN_SECONDS=60
timeout -k 10 500 python bench.py --gpus 6 --steps 10 --warmup 3 --no-cpu-baseline --halo-compare > $O/bench_rehearsal_6_processes_stiffness.json 2> $O/b6.err || { tail -30 $O/b6.err; exit 1; }
grep "first contact\|halo transport\|halo compare" $O/b6.err | cut -c1-260
timeout -k 10 500 python bench.py --gpus 6 --mode mass --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rehearsal_6_processes_mass.json 2> $O/b6m.err || { tail -30 $O/b6m.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 4 --mode rk4 --perturbed --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rehearsal_4_processes_rk4.json 2> $O/b4r.err || { tail -30 $O/b4r.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 4 --mode westervelt --degree 6 --cells 18 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rehearsal_4_processes_westervelt.json 2> $O/b4w.err || { tail -30 $O/b4w.err; exit 1; }
unset FUS_BENCH_REHEARSAL
timeout -k 10 600 python tools/rehearse_8_ranks.py > $O/rehearsal_8_ranks_full_size.json 2> $O/r8.err || { tail -30 $O/r8.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_rehearsal_6_processes_stiffness", "bench_rehearsal_6_processes_mass", "bench_rehearsal_4_processes_rk4", "bench_rehearsal_4_processes_westervelt"):
    d = json.load(open(f"gpurun_out/r05w/{f}.json"))
    c = d["config"]
    print(f, d["n_gpus"], d.get("valid"), c.get("partition"), c.get("halo_check"), c.get("halo_schedule"), (c.get("halo_transport") or "")[:30], round(d["ms_per_step"], 3),
          (d.get("check") or {}).get("rel_l2"), d["roofline"].get("kernel"), (c.get("geometry") or "")[:40])
    if c.get("halo_compare"):
        print("   halo_compare:", {k: round(v["ms_per_step_median"], 3) for k, v in c["halo_compare"]["transports"].items()}, c["halo_compare"]["not_compared"])
print(open("gpurun_out/r05w/rehearsal_8_ranks_full_size.json").read()[:600])
PY
echo done
