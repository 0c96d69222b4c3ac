#!/bin/bash
# round 5: PEER send / receive kernels with their loads issued together (two round trips per chunk instead of two per element) vs the
# committed form: halo GPU tests, then the default bench line's halo proxy and stand-alone scatters, interleaved
O=gpurun_out/r05y
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_halo_gpu.py -x -q > $O/pytest_halo.log 2>&1 || { tail -20 $O/pytest_halo.log; exit 1; }
tail -2 $O/pytest_halo.log
for rep in 1 2 3; do
  for v in prev tree; do
    if [ $v = prev ]; then lib=$PWD/tools/_bin/libfusgpu_prev.so; else lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; fi
    FUS_LIB_PATH=$lib timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_halo_${v}_$rep.json 2> $O/bench_halo_${v}_$rep.err || { tail -5 $O/bench_halo_${v}_$rep.err; exit 1; }
    python - $O/bench_halo_${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["roofline"]["secondary"]
print(f"{sys.argv[2]:5s} halo_proxy {s.get('halo_proxy')} scatter_peer_us {s.get('scatter_peer_us')}", flush=True)
PY
  done
done | tee $O/ab_ipc_loads.log
