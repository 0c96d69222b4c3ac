#!/bin/bash
# round 5: wall time of the no-flag bench.py (contract: "finishes within minutes")
O=gpurun_out/r05m
mkdir -p $O
t0=$(date +%s)
python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err; echo "bench rc=$?"
t1=$(date +%s); echo "bench.py without flags: $((t1 - t0)) s wall"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05m/bench_noflags.json"))
print(d["steps"], d["warmup"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["check"]["ok"], len(json.dumps(d)))
PY
echo done
