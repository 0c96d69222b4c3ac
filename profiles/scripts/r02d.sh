set -o pipefail
O=gpurun_out/r02d
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python tools/exp_numbering.py > $O/exp_numbering_p4.log 2>&1 || { tail -20 $O/exp_numbering_p4.log; exit 2; }
cat $O/exp_numbering_p4.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err || exit 3
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['isolated_frac'])"
