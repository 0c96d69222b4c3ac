#!/bin/bash
# round 5: whole GPU suite (no -x: a single failure must not hide the rest), the stream-visibility probe, the default bench line
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python tools/stream_visibility_probe.py --reps 8 > $O/probe.log 2>&1; echo "probe rc=$?"; cat $O/probe.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -5 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05b/bench_default.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "check")}, indent=1))
print(json.dumps(d["roofline"].get("secondary"), indent=1))
print(len(json.dumps(d["roofline"].get("secondary"))), "bytes of secondary")
print(json.dumps(d["aux"].get("halo_proxy"), indent=1)[:3000])
PY
echo done
