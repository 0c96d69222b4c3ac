set -o pipefail
O=gpurun_out/r02e
mkdir -p $O
python tools/host_overhead.py > $O/host_overhead.log 2>&1 || { tail -30 $O/host_overhead.log; exit 1; }
grep -v amdgpu.ids $O/host_overhead.log
FUS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --halo native > $O/bench_forcedist_native.json 2> $O/bench_forcedist_native.err || exit 2
FUS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --halo torch > $O/bench_forcedist_torch.json 2> $O/bench_forcedist_torch.err || exit 3
python - <<'PY'
import json
for t in ("native","torch"):
    d=json.load(open(f"gpurun_out/r02e/bench_forcedist_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["halo_exposed_ms"], d["config"]["halo_transport"])
PY
