#!/bin/bash
# round 6, VERDICT r5 item 3: upper bound of what removing the cell-local barriers of the in-kernel-geometry kernel could buy
# (tools/exp_geom_barriers.py: RACY builds of a copy of csrc/, timing only), interleaved with the shipped library, P = 4, 3, 7;
# then the wave-state counters of the shipped and the nobar23 build at P = 4 (separate --pmc pass, --kernel-trace only).
O=gpurun_out/r06f
mkdir -p $O
for pn in "4 54" "3 71" "7 31"; do
  set -- $pn
  for a in "" nobar23 nobar "" nobar23 nobar; do
    if [ -z "$a" ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_$a.so; fi
    ABLATE_P=$1 ABLATE_N=$2 FUS_LIB_PATH=$lib timeout -k 10 200 python tools/ablate_geom.py 2>&1 | grep -E "^P=" | sed "s/^/${a:-shipped}: /"
  done
done | tee $O/barrier_bound.log
cd /tmp && export TMPDIR=/tmp
for a in shipped nobar23; do
  if [ $a = shipped ]; then lib=$GRAFT_REPO_ROOT/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$GRAFT_REPO_ROOT/tools/_bin/libfusgpu_$a.so; fi
  export FUS_LIB_PATH=$lib ABLATE_P=4 ABLATE_N=54 ABLATE_SHORT=1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv \
    -d $GRAFT_REPO_ROOT/$O/pmc_$a -o sq -- python3 $GRAFT_REPO_ROOT/tools/ablate_geom.py > $GRAFT_REPO_ROOT/$O/pmc_$a.out 2>&1 || echo "pmc pass $a failed"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for a in ("shipped", "nobar23"):
    files = glob.glob(f"gpurun_out/r06f/pmc_{a}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if "stiffness_plan_geom_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        print(a, "no counters found in", files)
        continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{a}: launches {len(next(iter(acc.values())))}  " + "  ".join(f"{k} {v:.4g}" for k, v in sorted(m.items())))
    print(f"{a}: of SQ_WAVE_CYCLES: WAIT_ANY {100 * m.get('SQ_WAIT_ANY', 0) / wc:.1f} %  WAIT_INST_ANY {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} %  "
          f"ACTIVE_INST_ANY {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} %  WAIT_INST_LDS {100 * m.get('SQ_WAIT_INST_LDS', 0) / wc:.1f} %")
PY
