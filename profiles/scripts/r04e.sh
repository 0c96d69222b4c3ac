#!/bin/bash
# r04e: full-size rehearsals of bench.py's N > 1 path with REAL processes sharing the one GPU over the PEER transport.
# The pool's process guard allows at most 6 processes on the card, so --gpus 8 cannot run here: 6 ranks (3x2x1 blocks: two
# ranks with neighbours on BOTH sides in x) is the largest process-per-rank world; the 8-rank 2x2x2 topology runs on 4
# processes of 2 ranks in tests/test_halo_gpu.py::test_peer_transport_8_ranks_2x2x2_on_4_processes.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
export FUS_BENCH_REHEARSAL=1 FUS_IPC_SPIN_SECONDS=60
timeout -k 10 500 python bench.py --gpus 6 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04e/bench_rehearsal_6_processes_peer.json 2> gpurun_out/r04e/bench6.err || { tail -30 gpurun_out/r04e/bench6.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 6 --mode rk4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04e/bench_rehearsal_6_processes_rk4.json 2> gpurun_out/r04e/bench6rk4.err || { tail -30 gpurun_out/r04e/bench6rk4.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04e/bench_rehearsal_4_processes_peer.json 2> gpurun_out/r04e/bench4.err || { tail -30 gpurun_out/r04e/bench4.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_rehearsal_6_processes_peer", "bench_rehearsal_6_processes_rk4", "bench_rehearsal_4_processes_peer"):
    d = json.load(open(f"gpurun_out/r04e/{f}.json"))
    c = d["config"]
    print(f, d["n_gpus"], d.get("valid"), c.get("partition"), c.get("halo_check"), c.get("halo_schedule"), (c.get("halo_transport") or "")[:40], d["ms_per_step"])
PY
