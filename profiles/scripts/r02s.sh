set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r02s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for MODE in "stiffness_geom f64" "stiffness f32" "stiffness f64"; do set -- $MODE
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_$1_$2 -o sq -- python3 $R/bench.py --mode $1 --dtype $2 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/err_$1_$2 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/inst_$1_$2 -o inst -- python3 $R/bench.py --mode $1 --dtype $2 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/err2_$1_$2 || echo "inst pass failed"
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02s/*_*_f*")):
    for f in glob.glob(d+"/*counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "stiffness_plan" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d.split("/")[-1], {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
