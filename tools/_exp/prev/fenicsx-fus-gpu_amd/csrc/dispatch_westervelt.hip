// libfusgpu.so: the fused Westervelt cell passes (general G and in-kernel geometry): validation + dispatch over degree.
// Compiled once per scalar type (-DFUS_INST_T=double|float), see Makefile and fus_dispatch.hpp.
#include "fus_dispatch.hpp"
#include "westervelt.hpp"
#include "westervelt_geom.hpp"

#ifndef FUS_INST_T  // the Makefile builds both; a bare ``hipcc -c`` of this file checks the fp64 instances
#define FUS_INST_T double
#endif

namespace fus_abi {

template <typename T>
int westervelt_cell(const T* u, const T* v, const T* c2, const T* c3, const T* c4, const T* c5, T* b, T* m, const T* G,
                    const T* detJ, const void* ws, const T* dphi, int P, int64_t ncell, void* stream) {
  if (ncell < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (P < FUS_MIN_DEGREE || P > FUS_MAX_DEGREE) return FUS_ERR_UNSUPPORTED_DEGREE;
  if (ncell == 0) return FUS_OK;
  const bool mass = c2 || c5 || m || detJ;  // all four or none: none = the stiffness part alone
  if (!u || !v || !c3 || !c4 || !b || !G || !ws || !dphi) return FUS_ERR_INVALID_ARGUMENT;
  if (mass && (!c2 || !c5 || !m || !detJ)) return FUS_ERR_INVALID_ARGUMENT;
  if (misaligned(G, 2 * sizeof(T)) || misaligned(ws, 256)) return FUS_ERR_INVALID_ARGUMENT;
  bool ord = false, rp = true;
  if (!plan_check(ws, (P + 1) * (P + 1) * (P + 1), cells_per_batch(P), ncell, &ord, nullptr, &rp)) return FUS_ERR_PLAN_MISMATCH;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipErrorInvalidValue;
  switch (P) {
#define FUS_CASE(PP) \
  case PP:           \
    e = mass ? fus::launch_westervelt_cell<T, PP, true>(u, v, c2, c3, c4, c5, b, m, G, detJ, ws, dphi, ncell, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)) \
             : fus::launch_westervelt_cell<T, PP, false>(u, v, c2, c3, c4, c5, b, m, G, detJ, ws, dphi, ncell, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); \
    break;
    FUS_CASE(1) FUS_CASE(2) FUS_CASE(3) FUS_CASE(4) FUS_CASE(5) FUS_CASE(6) FUS_CASE(7) FUS_CASE(8) FUS_CASE(9)
    FUS_CASE(10)
#undef FUS_CASE
  }
  return hip_rc(e);
}

template <typename T>
int westervelt_cell_geom(const T* u, const T* v, const T* c2, const T* c3, const T* c4, const T* c5, T* b, T* m,
                         const T* x_g, const int32_t* x_dofs, const T* pts, const T* wts, const void* ws, const T* dphi,
                         int P, int64_t ncell, void* stream) {
  if (ncell < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (P < FUS_MIN_DEGREE || P > FUS_MAX_DEGREE) return FUS_ERR_UNSUPPORTED_DEGREE;
  if (ncell == 0) return FUS_OK;
  const bool mass = c2 || c5 || m;
  if (!u || !v || !c3 || !c4 || !b || !x_g || !x_dofs || !pts || !wts || !ws || !dphi) return FUS_ERR_INVALID_ARGUMENT;
  if (mass && (!c2 || !c5 || !m)) return FUS_ERR_INVALID_ARGUMENT;
  if (misaligned(ws, 256)) return FUS_ERR_INVALID_ARGUMENT;
  bool ord = false, rp = true;
  if (!plan_check(ws, (P + 1) * (P + 1) * (P + 1), cells_per_batch(P), ncell, &ord, nullptr, &rp)) return FUS_ERR_PLAN_MISMATCH;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipErrorInvalidValue;
  switch (P) {
#define FUS_CASE(PP) \
  case PP:           \
    e = mass ? fus::launch_westervelt_cell_geom<T, PP, true>(u, v, c2, c3, c4, c5, b, m, x_g, x_dofs, pts, wts, ws, dphi, ncell, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)) \
             : fus::launch_westervelt_cell_geom<T, PP, false>(u, v, c2, c3, c4, c5, b, m, x_g, x_dofs, pts, wts, ws, dphi, ncell, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); \
    break;
    FUS_CASE(1) FUS_CASE(2) FUS_CASE(3) FUS_CASE(4) FUS_CASE(5) FUS_CASE(6) FUS_CASE(7) FUS_CASE(8) FUS_CASE(9)
    FUS_CASE(10)
#undef FUS_CASE
  }
  return hip_rc(e);
}

template int westervelt_cell<FUS_INST_T>(const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, FUS_INST_T*, FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const void*, const FUS_INST_T*, int, int64_t, void*);
template int westervelt_cell_geom<FUS_INST_T>(const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, FUS_INST_T*, FUS_INST_T*, const FUS_INST_T*, const int32_t*, const FUS_INST_T*, const FUS_INST_T*, const void*, const FUS_INST_T*, int, int64_t, void*);

}  // namespace fus_abi
