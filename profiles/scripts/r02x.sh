set -o pipefail
O=gpurun_out/r02x
mkdir -p $O
for L in libfusgpu.so _ab/libfusgpu_noldsatomic.so; do
FUS_LIB_PATH=$PWD/fenicsx-fus-gpu_amd/csrc/$L python tools/ab_stiffness.py --degree 4 --dtype f32 --rounds 5 plan >> $O/ab_f32.log 2>&1 || exit 1
FUS_LIB_PATH=$PWD/fenicsx-fus-gpu_amd/csrc/$L python tools/ab_stiffness.py --degree 4 --dtype f64 --rounds 5 plan geom >> $O/ab_f64.log 2>&1 || exit 1
done
grep -E "^plan|^geom|lib=" $O/ab_f32.log $O/ab_f64.log | sed 's/.*csrc\///'
