set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r02w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for DT in f32 f64; do
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/lds_$DT -o lds -- python3 $R/bench.py --dtype $DT --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$DT.json 2> $O/err_$DT || echo "pass failed $DT"
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02w/lds_*")):
    for f in glob.glob(d+"/*counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "stiffness_plan" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d.split("/")[-1], {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
tail -3 $O/err_f32
