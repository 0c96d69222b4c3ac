#!/bin/bash
# round 5: GPU suite + smoke + default bench line of the restructured-preamble library (7 translation units)
O=gpurun_out/r05r
mkdir -p $O
date
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
date
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05r/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "check", d.get("check"), "lib", d.get("config", {}).get("library"))
PY
date
