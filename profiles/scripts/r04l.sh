#!/bin/bash
# r04l: stand-alone scatter on the caller's stream + the streaming-mode parity test; halo tests; scatter line
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
timeout -k 10 1000 python -m pytest tests/test_halo_gpu.py tests/test_operators_gpu.py tests/test_rk4_golden.py tests/test_abi.py -m gpu -x -q -k "not maximum_size" > gpurun_out/r04l/tests.log 2>&1
rc=$?
tail -4 gpurun_out/r04l/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --mode scatter --steps 100 > gpurun_out/r04l/bench_scatter.json 2> gpurun_out/r04l/bench_scatter.err || { tail -20 gpurun_out/r04l/bench_scatter.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04l/bench_scatter.json"))["scatter"]
for t, r in d["transports"].items():
    print(t, {k: {a: round(b, 1) for a, b in v.items()} for k, v in r.items() if isinstance(v, dict)})
PY
