#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/run_profile.sh <tag> [bench args...]
# Produces under gpurun_out/prof_<tag>/: kernel-trace stats and two PMC passes (FETCH_SIZE, WRITE_SIZE
# in separate runs, as the TCC slots require), all of the SAME bench command.
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-aux --no-check $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 3
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o tcc -- python3 $REPO/bench.py $ARGS > $OUT/bench_tcc.json 2> $OUT/bench_tcc.err || echo "tcc pass failed (non-fatal)"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/bench_sq.err || echo "sq pass failed (non-fatal)"
find $OUT -name "*.csv" | head -50
# large per-dispatch traces are not needed for the summary: keep only stats + counter csvs small
find $OUT -name "*.db" -delete
ls -la $OUT/*
