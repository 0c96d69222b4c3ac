// libfusgpu.so: the affine and the in-kernel-geometry planned stiffness applies: validation + dispatch over degree.
// Compiled once per scalar type (-DFUS_INST_T=double|float), see Makefile and fus_dispatch.hpp.
#include "fus_dispatch.hpp"
#include "stiffness_affine.hpp"
#include "stiffness_geom.hpp"

#ifndef FUS_INST_T  // the Makefile builds both; a bare ``hipcc -c`` of this file checks the fp64 instances
#define FUS_INST_T double
#endif

namespace fus_abi {

template <typename T>
int stiffness_apply_planned_affine(const T* x, const T* cc, T* y, const T* G, const T* wratio, const void* ws,
                                   const T* dphi, int P, int64_t ncell, void* stream) {
  if (ncell < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (P < FUS_MIN_DEGREE || P > FUS_MAX_DEGREE) return FUS_ERR_UNSUPPORTED_DEGREE;
  if (ncell == 0) return FUS_OK;
  if (!x || !cc || !y || !G || !wratio || !ws || !dphi) return FUS_ERR_INVALID_ARGUMENT;
  if (misaligned(G, 2 * sizeof(T)) || misaligned(ws, 256)) return FUS_ERR_INVALID_ARGUMENT;
  bool ord = false, rp = true;
  if (!plan_check(ws, (P + 1) * (P + 1) * (P + 1), cells_per_batch(P), ncell, &ord, nullptr, &rp)) return FUS_ERR_PLAN_MISMATCH;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipErrorInvalidValue;
  // P <= 4: unpadded LDS + 5 waves per SIMD (+8 %, profiles/r01f_affine_fast_path.log); above, registers do
  // not allow 5 waves without spilling: padded build, compiler's own allocation
  switch (P) {
#define FUS_CASE(PP) \
  case PP:           \
    e = fus::launch_stiffness_plan_affine<T, PP, true, (PP > 4), (PP <= 4 ? 5 : 1)>(x, cc, y, G, wratio, ws, dphi, ncell, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); \
    break;
    FUS_CASE(1) FUS_CASE(2) FUS_CASE(3) FUS_CASE(4) FUS_CASE(5) FUS_CASE(6) FUS_CASE(7) FUS_CASE(8) FUS_CASE(9)
    FUS_CASE(10)
#undef FUS_CASE
  }
  return hip_rc(e);
}

template <typename T>
int stiffness_apply_planned_geom(const T* x, const T* cc, T* y, const T* x_g, const int32_t* x_dofs, const T* pts,
                                 const T* wts, const void* ws, const T* dphi, int P, int64_t ncell, void* stream) {
  if (ncell < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (P < FUS_MIN_DEGREE || P > FUS_MAX_DEGREE) return FUS_ERR_UNSUPPORTED_DEGREE;
  if (ncell == 0) return FUS_OK;
  if (!x || !cc || !y || !x_g || !x_dofs || !pts || !wts || !ws || !dphi) return FUS_ERR_INVALID_ARGUMENT;
  if (misaligned(ws, 256)) return FUS_ERR_INVALID_ARGUMENT;
  bool ord = false, rp = true;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!plan_check(ws, (P + 1) * (P + 1) * (P + 1), cells_per_batch(P), ncell, &ord, nullptr, &rp)) return FUS_ERR_PLAN_MISMATCH;
  hipError_t e = hipErrorInvalidValue;
  switch (P) {
#define FUS_CASE(PP) \
  case PP:           \
    e = fus::launch_stiffness_plan_geom<T, PP, (PP >= 4), true, fus::geom_min_waves<T, PP>(), fus::geom_factors_in_registers<T, PP>()>(x, cc, y, x_g, x_dofs, pts, wts, ws, dphi, ncell, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); \
    break;
    FUS_CASE(1) FUS_CASE(2) FUS_CASE(3) FUS_CASE(4) FUS_CASE(5) FUS_CASE(6) FUS_CASE(7) FUS_CASE(8) FUS_CASE(9)
    FUS_CASE(10)
#undef FUS_CASE
  }
  return hip_rc(e);
}

template int stiffness_apply_planned_affine<FUS_INST_T>(const FUS_INST_T*, const FUS_INST_T*, FUS_INST_T*, const FUS_INST_T*, const FUS_INST_T*, const void*, const FUS_INST_T*, int, int64_t, void*);
template int stiffness_apply_planned_geom<FUS_INST_T>(const FUS_INST_T*, const FUS_INST_T*, FUS_INST_T*, const FUS_INST_T*, const int32_t*, const FUS_INST_T*, const FUS_INST_T*, const void*, const FUS_INST_T*, int, int64_t, void*);

}  // namespace fus_abi
