// libfusgpu.so: the general-G planned stiffness apply (fus_stiffness_apply_planned_*): validation + dispatch over degree and build.
// Compiled once per scalar type (-DFUS_INST_T=double|float), see Makefile and fus_dispatch.hpp.
#include "fus_dispatch.hpp"
#include "stiffness_plan.hpp"

#ifndef FUS_INST_T  // the Makefile builds both; a bare ``hipcc -c`` of this file checks the fp64 instances
#define FUS_INST_T double
#endif

namespace fus_abi {

// fp32 build with 5 waves per SIMD (only instantiated for float)
template <typename T, int P>
hipError_t launch_plan_f32_5w(const T* x, const T* cc, T* y, const T* G, const void* ws, const T* dphi, int64_t ncell,
                              int remap, hipStream_t s, bool ord, bool runs) {
  if constexpr (sizeof(T) == 4 && P <= 4)
    return fus::launch_stiffness_plan<T, P, false, true, 5>(x, cc, y, G, ws, dphi, ncell, remap, s, ord, runs);
  else
    return fus::launch_stiffness_plan<T, P, false, true, 1>(x, cc, y, G, ws, dphi, ncell, remap, s, ord, runs);
}

template <typename T>
int stiffness_apply_planned(const T* x, const T* cc, T* y, const T* G, const void* ws, const T* dphi, int P,
                            int64_t ncell, void* stream) {
  if (ncell < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (P < FUS_MIN_DEGREE || P > FUS_MAX_DEGREE) return FUS_ERR_UNSUPPORTED_DEGREE;
  if (ncell == 0) return FUS_OK;
  if (!x || !cc || !y || !G || !ws || !dphi) return FUS_ERR_INVALID_ARGUMENT;
  if (misaligned(G, 2 * sizeof(T)) || misaligned(ws, 256)) return FUS_ERR_INVALID_ARGUMENT;
  bool ord = false, rp = true;
  if (!plan_check(ws, (P + 1) * (P + 1) * (P + 1), cells_per_batch(P), ncell, &ord, nullptr, &rp)) return FUS_ERR_PLAN_MISMATCH;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int remap = g_xcd_remap.load(std::memory_order_relaxed);
  hipError_t e = hipErrorInvalidValue;
  // Builds (profiles/r01d_ab_alias_by_degree.log, r02*_ab_*.log; pinned by tests/test_resource_usage.py):
  //   0  three LDS cubes + own x/y buffer           (P <= 3)
  //   1  LDS-aliased, whole G slab issued up front  (P = 4, 5: 4 workgroups per CU at P = 4)
  //   2  LDS-aliased, ring of G slabs               (P >= 6: registers are the binding limit there; P = 8 also
  //                                                  drops the LDS padding to fit a third workgroup per CU)
  //   30 fp32, registers allow 5 waves per SIMD     (fp32, P <= 4)
  int pv = g_plan_variant.load(std::memory_order_relaxed);
  if (pv < 0) {
    if (sizeof(T) == 4)
      pv = (P <= 4) ? 30 : 1;  // fp32: registers are not the limit, the whole G slab up front wins (r02y_ab_fp32.log)
    else {
      // measured per degree at ~10 M dofs (profiles/r02a_ab_builds_and_slp.log, r02b_ab_isolated_and_degrees.log,
      // r02h_ab_degrees_3_9_10.log, r02r_ab_degrees_8_9_10.log): the ring wins where it buys a workgroup per CU
      static const int best[11] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 2};
      pv = best[P];
    }
  }
  switch (P) {
#define FUS_CASE(PP)                                                                                      \
  case PP:                                                                                                \
    switch (pv) {                                                                                         \
      case 1: e = fus::launch_stiffness_plan<T, PP, true, true, 1>(x, cc, y, G, ws, dphi, ncell, remap, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); break;   \
      case 2: e = fus::launch_stiffness_plan<T, PP, true, (PP != 8), fus::plan_ring_min_waves<PP>(), fus::plan_g_ring<PP>()>(x, cc, y, G, ws, dphi, ncell, remap, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); break; \
      case 30: e = launch_plan_f32_5w<T, PP>(x, cc, y, G, ws, dphi, ncell, remap, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); break; \
      default: e = fus::launch_stiffness_plan<T, PP, false, true, 1>(x, cc, y, G, ws, dphi, ncell, remap, s, ord, plan_use_runs<T>((P + 1) * (P + 1) * (P + 1), rp)); break; \
    }                                                                                                     \
    break;
    FUS_CASE(1) FUS_CASE(2) FUS_CASE(3) FUS_CASE(4) FUS_CASE(5) FUS_CASE(6) FUS_CASE(7) FUS_CASE(8) FUS_CASE(9)
    FUS_CASE(10)
#undef FUS_CASE
  }
  return hip_rc(e);
}

template int stiffness_apply_planned<FUS_INST_T>(const FUS_INST_T*, const FUS_INST_T*, FUS_INST_T*, const FUS_INST_T*, const void*, const FUS_INST_T*, int, int64_t, void*);

}  // namespace fus_abi
