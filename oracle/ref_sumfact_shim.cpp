// ORACLE support (test infrastructure).  Thin extern "C" shim around the REFERENCE's own
// C++ primitives, compiled from the header where it lies under /root/reference
// (-I$(REF)/cpp/common; nothing is copied into this repo).  Output: oracle/_ref/libref_sumfact.so.
// The header needs only <array> included first (it uses std::array without including it).
// The rest of the reference's C++ path (spectral_op.hpp, precompute.hpp, ...) needs
// dolfinx/basix headers that this image lacks => unbuildable here.
#include <array>
#include <cstdint>

#include "sum_factorisation.hpp"  // reference: cpp/common/sum_factorisation.hpp

namespace {
template <int n>
void contract_n(int tr, const double* A, const double* B, double* C) {
  if (tr)
    contract<double, n, n, n, n, true>(A, B, C);  // :70-86
  else
    contract<double, n, n, n, n, false>(A, B, C);
}
template <int n>
void transpose_n(int which, double* A, double* B) {
  if (which == 0)
    transpose<double, n, n, n, n, n * n, 1>(A, B);  // "transpose_y" of numba-cpu/operators.py:88
  else
    transpose<double, n, n, n, 1, n, n * n>(A, B);  // "transpose_z" of numba-cpu/operators.py:89
}
}  // namespace

extern "C" {
int ref_contract_f64(int n, int tr, const double* A, const double* B, double* C) {
  switch (n) {
    case 3: contract_n<3>(tr, A, B, C); return 0;
    case 4: contract_n<4>(tr, A, B, C); return 0;
    case 5: contract_n<5>(tr, A, B, C); return 0;
    case 7: contract_n<7>(tr, A, B, C); return 0;
  }
  return -1;
}
int ref_transpose_f64(int n, int which, double* A, double* B) {
  switch (n) {
    case 3: transpose_n<3>(which, A, B); return 0;
    case 4: transpose_n<4>(which, A, B); return 0;
    case 5: transpose_n<5>(which, A, B); return 0;
    case 7: transpose_n<7>(which, A, B); return 0;
  }
  return -1;
}
}
