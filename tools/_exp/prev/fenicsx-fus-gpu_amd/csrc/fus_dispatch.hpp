// State and helpers shared by the translation units of libfusgpu.so (C ABI: include/fus_gpu.h).  The library is several objects so
// that the planned cell operators -- by far the most kernel instantiations: 10 degrees x builds x (ORDERED, RUNS) x 2 scalar types --
// compile in parallel (Makefile); everything here is ``inline`` (one instance in the linked library).
#pragma once
#include "../../include/fus_gpu.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "plan.hpp"

namespace fus_abi {

inline std::atomic<int> g_stiffness_variant{0};
inline std::atomic<int> g_xcd_remap{0};  // measured slower on MI355X (profiles/r01b_ab_variants.log)
inline std::atomic<int> g_mass_variant{0};
inline std::atomic<int> g_plan_runs{1};  // 0 never, 1 auto, 2 always

// Run-length coded dof lists (8 bytes per run of consecutive dofs instead of 4 per dof; expanded in LDS by
// the apply kernels): the builder decides per batch (a list that does not compress stays raw).
inline int plan_allow_runs(int ndof_per_entity) {
  (void)ndof_per_entity;
  return g_plan_runs.load(std::memory_order_relaxed) != 0;
}
// which encoding of the dof lists a launch reads (the plan holds both).  ``runs_pay``: at least half of the plan's batches carry a
// run table -- a numbering whose lists do not compress (Morton, a graph reordering) makes a run-coded launch read every list one round
// trip late, behind a wasted speculative read of the table (+2..3 %, profiles/r05y_numbering.log)
template <typename T>
inline bool plan_use_runs(int ndof_per_entity, bool runs_pay = true) {
  const int mode = g_plan_runs.load(std::memory_order_relaxed);
  if (mode == 1 && !runs_pay) return false;
  // auto: fp64 always (+4..6 % at every degree); fp32 up to P = 8, i.e. wherever the preamble reads the run words speculatively and
  // issues its loads by every thread (plan.hpp: +7..12 % at P = 2, 4, 5, 6, +7 % at P = 7, +3 % at P = 8; P = 9, 10 keep the raw
  // lists).  Before that the fp32 limit was P = 4 (-12 % at P = 6 then): profiles/r05x_ab_run_tables.log,
  // r05x_ab_run_tables_fp32_p78.log; r02o_ab_run_tables.log, r02y_ab_fp32.log for the earlier kernels
  return mode == 2 || (mode == 1 && (sizeof(T) == 8 || ndof_per_entity <= 729));
}
inline std::atomic<int> g_plan_variant{-1};  // -1 = auto

// Host mirror of the plans built through this library, keyed by workspace address: the apply entry
// points check that a workspace was built, and for the (N, entities per batch, entity count) they are
// called with, before any kernel indexes it (a mismatch would gather / scatter out of bounds), and
// learn from it whether the plan carries a cell order.
struct PlanInfo {
  int N = 0, epb = 0;
  int64_t nent = 0;
  bool ordered = false;
  bool exclusive = false;  // fus_plan_mark_exclusive has run: the plan carries exclusive-dof marks
  bool runs_pay = true;    // at least half of the batches carry a run table (plan_use_runs)
  int64_t nbatch = 0, with_runs = 0;
};
inline std::mutex g_plans_mu;
inline std::unordered_map<const void*, PlanInfo> g_plans;

inline void plan_register(const void* ws, int N, int epb, int64_t nent, bool ordered, int64_t nbatch = 0, int64_t with_runs = 0) {
  std::lock_guard<std::mutex> lk(g_plans_mu);
  g_plans[ws] = PlanInfo{N, epb, nent, ordered, false, 2 * with_runs >= nbatch, nbatch, with_runs};
}
// true if ``ws`` holds a plan for exactly this shape; ``ordered`` out
inline bool plan_check(const void* ws, int N, int epb, int64_t nent, bool* ordered, bool* exclusive = nullptr, bool* runs_pay = nullptr) {
  std::lock_guard<std::mutex> lk(g_plans_mu);
  auto it = g_plans.find(ws);
  if (it == g_plans.end()) return false;
  const PlanInfo& p = it->second;
  if (p.N != N || p.epb != epb || p.nent != nent) return false;
  *ordered = p.ordered;
  if (exclusive) *exclusive = p.exclusive;
  if (runs_pay) *runs_pay = p.runs_pay;
  return true;
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? FUS_OK : FUS_ERR_HIP_BASE - (int)e; }

inline bool misaligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

inline int cells_per_batch(int P) {
  const int n2 = (P + 1) * (P + 1);
  return 256 / n2 > 0 ? 256 / n2 : 1;
}

template <int P>
int64_t plan_bytes_p(int64_t ncell) {
  return fus::plan_view(nullptr, P, fus::plan_cells_per_batch<P>(), ncell).bytes;
}

inline int64_t plan_bytes(int P, int64_t ncell) {
  switch (P) {
#define FUS_CASE(PP) \
  case PP:           \
    return plan_bytes_p<PP>(ncell);
    FUS_CASE(1) FUS_CASE(2) FUS_CASE(3) FUS_CASE(4) FUS_CASE(5) FUS_CASE(6) FUS_CASE(7) FUS_CASE(8) FUS_CASE(9)
    FUS_CASE(10)
#undef FUS_CASE
  }
  return FUS_ERR_UNSUPPORTED_DEGREE;
}


// ---- the planned cell operators: defined and explicitly instantiated (one object per operator family and scalar type, compiled in
// parallel: Makefile) in dispatch_stiffness_plan.hip, dispatch_geometry.hip and dispatch_westervelt.hip
template <typename T>
int stiffness_apply_planned(const T* x, const T* cc, T* y, const T* G, const void* ws, const T* dphi, int P, int64_t ncell, void* stream);
template <typename T>
int stiffness_apply_planned_affine(const T* x, const T* cc, T* y, const T* G, const T* wratio, const void* ws, const T* dphi, int P,
                                   int64_t ncell, void* stream);
template <typename T>
int stiffness_apply_planned_geom(const T* x, const T* cc, T* y, const T* x_g, const int32_t* x_dofs, const T* pts, const T* wts,
                                 const void* ws, const T* dphi, int P, int64_t ncell, void* stream);
template <typename T>
int westervelt_cell(const T* u, const T* v, const T* c2, const T* c3, const T* c4, const T* c5, T* b, T* m, const T* G, const T* detJ,
                    const void* ws, const T* dphi, int P, int64_t ncell, void* stream);
template <typename T>
int westervelt_cell_geom(const T* u, const T* v, const T* c2, const T* c3, const T* c4, const T* c5, T* b, T* m, const T* x_g,
                         const int32_t* x_dofs, const T* pts, const T* wts, const void* ws, const T* dphi, int P, int64_t ncell,
                         void* stream);

}  // namespace fus_abi
