/*
 * ORACLE (test infrastructure -- NOT product code).
 *
 * Plain-C CPU restatement of the reference's numba-cpu hot path, used
 *   (1) as the checker in tests/ and __graft_entry__.smoke(), and
 *   (2) as bench.py's ``cpu_baseline`` (kind "port"): the reference's numba path
 *       cannot run on the GPU box (numba absent, the reference never travels).
 * Nothing in the product package links, loads or calls this library.
 *
 * Pinned against golden vectors produced by running the reference itself
 * (the .npz fixtures under tests/golden/, see tests/test_oracle_golden.py).
 *
 * Build: see oracle/Makefile (-O3 -ffast-math mirrors numba fastmath=True and the
 * reference's -Ofast, cpp/time_operators/CMakeLists.txt:12-13).
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUF f64
#include "fus_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL float
#define SUF f32
#include "fus_oracle_impl.h"
#undef REAL
#undef SUF

#ifdef _OPENMP
#include <omp.h>
int oracle_max_threads(void) { return omp_get_max_threads(); }
#else
int oracle_max_threads(void) { return 1; }
#endif

/* Test hooks: the two primitives on their own (checked against the reference's C++
 * templates compiled into oracle/_ref/libref_sumfact.so). */
void oracle_contract_f64(int Nk, int Na, int Nb, int Nc, int transposeA, const double* A, const double* B, double* C) {
  contract_f64(Nk, Na, Nb, Nc, transposeA, A, B, C);
}
void oracle_transpose_f64(int Na, int Nb, int Nc, int offa, int offb, int offc, const double* A, double* B) {
  transpose_f64(Na, Nb, Nc, offa, offb, offc, A, B);
}
