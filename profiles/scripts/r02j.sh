set -o pipefail
O=gpurun_out/r02j
mkdir -p $O
for i in 1 2; do
for L in libfusgpu.so _ab/libfusgpu_excl35.so _ab/libfusgpu_excl100.so; do
FUS_LIB_PATH=$PWD/fenicsx-fus-gpu_amd/csrc/$L python tools/ab_stiffness.py --degree 4 --rounds 5 plan >> $O/ab_excl_p4.log 2>&1 || exit 1
done
done
grep -E "^plan|lib=" $O/ab_excl_p4.log | sed 's/.*lib=//' 
for L in libfusgpu.so _ab/libfusgpu_excl35.so _ab/libfusgpu_excl100.so; do
FUS_LIB_PATH=$PWD/fenicsx-fus-gpu_amd/csrc/$L python tools/ab_stiffness.py --degree 6 --cells 36 --rounds 5 plan >> $O/ab_excl_p6.log 2>&1 || exit 1
done
grep -E "^plan|lib=" $O/ab_excl_p6.log | sed 's/.*lib=//'
