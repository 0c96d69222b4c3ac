#!/bin/bash
# round 5, first contact: the whole GPU suite, then the default bench line (check vs oracle, roofline.secondary, aux.halo_proxy)
set -e
O=gpurun_out/r05a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -30 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a/bench_default.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "check")}, indent=1))
print(json.dumps(d["roofline"].get("secondary"), indent=1))
print(len(json.dumps(d["roofline"].get("secondary"))), "bytes of secondary")
PY
echo done
