#!/bin/bash
# round 5: A/B of the restructured preamble (2 round trips instead of ~8; built from a scratch copy into tools/_bin/libfusgpu_v2.so)
# against the shipped library: the default bench line with each, twice, interleaved.
O=gpurun_out/r05q
mkdir -p $O
for rep in 1 2; do
  for v in shipped v2; do
    if [ $v = shipped ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_v2.so; fi
    FUS_LIB_PATH=$lib timeout -k 10 300 python bench.py > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err || { echo "bench $v $rep failed"; tail -5 $O/bench_${v}_$rep.err; exit 1; }
    python - $O/bench_${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
a = d.get("aux", {})
def g(k, f): 
    v = a.get(k, {}); return v.get(f) if isinstance(v, dict) else None
print(f"{sys.argv[2]:8s} value {d['value']:.3f} ms_per_step {d['ms_per_step']:.4f} frac {d['roofline']['frac']:.3f} check {d.get('check',{}).get('rel_l2')}"
      f" | sustained {g('sustained','ms_per_step')} mass {g('mass','kernel_ms')} geom {g('stiffness_in_kernel_geometry','kernel_ms')}"
      f" rk4 {g('rk4_step','ms_per_step')} rk4_geom {g('rk4_step_in_kernel_geometry','ms_per_step')} wv {g('westervelt_step','ms_per_step')}"
      f" wv_geom {g('westervelt_step_in_kernel_geometry','ms_per_step')} wv_1g {g('westervelt_step_single_gather','ms_per_step')}", flush=True)
PY
  done
done | tee $O/ab_preamble.log
