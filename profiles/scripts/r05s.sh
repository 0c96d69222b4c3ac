#!/bin/bash
# round 5: GPU suite, then A/B of the default bench line: library at the previous commit (tools/_bin/libfusgpu_prev.so, if present) vs this tree
O=gpurun_out/r05s
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for v in prev tree; do
    if [ $v = prev ]; then lib=$PWD/tools/_bin/libfusgpu_prev.so; [ -f $lib ] || continue; else lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; fi
    FUS_LIB_PATH=$lib timeout -k 10 300 python bench.py > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err || { echo "bench $v $rep failed"; tail -5 $O/bench_${v}_$rep.err; exit 1; }
    python - $O/bench_${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
out = [f"{sys.argv[2]:5s} region {d['ms_per_step']:.4f} ms frac {d['roofline']['frac']:.3f} check {d['check']['rel_l2']:.2e}"]
for k, val in d["aux"].items():
    if isinstance(val, dict):
        for f in ("ms_per_step", "ms_per_apply"):
            if isinstance(val.get(f), (int, float)):
                out.append(f"{k}={val[f]:.4f}")
print(" ".join(out), flush=True)
PY
  done
done | tee $O/ab_preamble.log
