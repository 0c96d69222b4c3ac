set -o pipefail
O=gpurun_out/r02c
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_abi.py tests/test_solver_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
./examples/c_abi_golden tests/golden/ops_P4_2x2x2_pert_float64.bin > $O/c_abi_golden.log 2>&1; cat $O/c_abi_golden.log
python tools/exp_numbering.py > $O/exp_numbering_p4.log 2>&1 || { tail -20 $O/exp_numbering_p4.log; exit 2; }
cat $O/exp_numbering_p4.log
