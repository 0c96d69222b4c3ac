#!/bin/bash
# texture-addresser / L1 counters of the atomic-free mass kernel (what bounds it once HBM traffic is the algorithmic 1.06 x).
# Two counters of a block per pass (a pass asking for five TA counters aborts with "exceeds the capabilities of the hardware"
# and then hangs in rocprofv3's finalisation: every pass runs under its own timeout).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r04u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--mode mass --steps 20 --warmup 3 --no-cpu-baseline --no-aux"
pass() {  # name, counters...
  local name=$1; shift
  echo "pass $name: $*"
  timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -o $name -- python3 $REPO/bench.py $ARGS > $O/bench_$name.json 2> $O/bench_$name.err || { echo "pass $name failed"; grep -m2 "error code\|capabilities" $O/bench_$name.err; return 1; }
}
pass ta1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 2
pass ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum || exit 3
pass ta3 TA_FLAT_WRITE_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || exit 4
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum || exit 5
find $O -name "*.db" -delete
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/pmc_*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mass_gather_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f.split("/")[-1], k, sum(v) / len(v), len(v))
PY
