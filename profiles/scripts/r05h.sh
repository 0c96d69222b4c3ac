#!/bin/bash
# round 5: the whole GPU suite on the library without the chained-plan experiment; default bench line
O=gpurun_out/r05h
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05h/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["check"]["rel_l2"])
print(json.dumps(d["roofline"].get("secondary")), len(json.dumps(d["roofline"].get("secondary"))))
PY
echo done
