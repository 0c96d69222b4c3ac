#!/bin/bash
# round 5: A/B of the default bench line: this tree's library vs scratch builds with (x1) the fork-signal publish after the load issue and a
# branch-free block remap (kernel-argument loads in one batch) and (x2) x1 + -mllvm -amdgpu-kernarg-preload-count=16
O=gpurun_out/r05u
mkdir -p $O
for rep in 1 2; do
  for v in tree x1 x2; do
    if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_$v.so; fi
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
done | tee $O/ab_kernarg.log
