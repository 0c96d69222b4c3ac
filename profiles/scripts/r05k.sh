#!/bin/bash
# round 5, final: the whole GPU suite, the driver's command on the final library, the degree sweep
O=gpurun_out/r05k
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05k/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["check"]["rel_l2"])
print(json.dumps(d["roofline"].get("secondary")), len(json.dumps(d["roofline"].get("secondary"))))
PY
timeout -k 10 900 python tools/sweep.py --degrees 2,3,4,5,6,7,8 > $O/sweep_degrees.log 2>&1; echo "sweep rc=$?"; grep "^P=" $O/sweep_degrees.log | cut -c1-150
echo done
