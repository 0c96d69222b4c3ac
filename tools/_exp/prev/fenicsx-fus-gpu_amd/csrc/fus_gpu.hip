// libfusgpu.so -- C ABI (include/fus_gpu.h) over the CDNA4 kernels in this directory.
// Build: see Makefile (hipcc --offload-arch=gfx950).
#include "../../include/fus_gpu.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "fus_dispatch.hpp"
#include "geometry.hpp"
#include "halo.hpp"
#include "halo_comm.hpp"
#include "mass.hpp"
#include "mass_gather.hpp"
#include "plan_build.hpp"
#include "rk4.hpp"
#include "stiffness.hpp"
#include "vecops.hpp"
#include "westervelt.hpp"

using namespace fus_abi;

namespace {

// transposed-dofmap plans of the atomic-free mass apply (csrc/mass_gather.hpp), keyed by workspace address (under g_plans_mu)
std::unordered_map<const void*, fus::GatherHeader> g_gather_plans;

template <typename T, int P>
hipError_t stiffness_dispatch_variant(const T* x, const T* cc, T* y, const T* G, const int32_t* dofmap,
                                      const T* dphi, int64_t ncell, hipStream_t s) {
  const int variant = g_stiffness_variant.load(std::memory_order_relaxed);
  const int remap = g_xcd_remap.load(std::memory_order_relaxed);
  constexpr int CPB256 = fus::default_cells_per_block<P>(256);
  constexpr int CPB128 = fus::default_cells_per_block<P>(128);
  switch (variant) {
    case 1:  // ~128-thread workgroups
      return fus::launch_stiffness_col<T, P, CPB128>(x, cc, y, G, dofmap, dphi, ncell, remap, s);
    default:  // ~256-thread workgroups
      return fus::launch_stiffness_col<T, P, CPB256>(x, cc, y, G, dofmap, dphi, ncell, remap, s);
  }
}

template <typename T>
int stiffness_apply(const T* x, const T* cc, T* y, const T* G, const int32_t* dofmap, const T* dphi, int P,
                    int64_t ncell, void* stream) {
  if (ncell < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (P < FUS_MIN_DEGREE || P > FUS_MAX_DEGREE) return FUS_ERR_UNSUPPORTED_DEGREE;
  if (ncell == 0) return FUS_OK;
  if (!x || !cc || !y || !G || !dofmap || !dphi) return FUS_ERR_INVALID_ARGUMENT;
  if (misaligned(G, 2 * sizeof(T)) || misaligned(x, sizeof(T)) || misaligned(y, sizeof(T)) ||
      misaligned(dofmap, sizeof(int32_t)))
    return FUS_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipErrorInvalidValue;
  switch (P) {
#define FUS_CASE(PP) \
  case PP:           \
    e = stiffness_dispatch_variant<T, PP>(x, cc, y, G, dofmap, dphi, ncell, s); \
    break;
    FUS_CASE(1) FUS_CASE(2) FUS_CASE(3) FUS_CASE(4) FUS_CASE(5) FUS_CASE(6) FUS_CASE(7) FUS_CASE(8) FUS_CASE(9)
    FUS_CASE(10)
#undef FUS_CASE
  }
  return hip_rc(e);
}

template <typename T>
int mass_apply(const T* x, const T* consts, T* y, const T* detJ, const int32_t* dofmap, int N, int64_t nent,
               void* stream) {
  if (nent < 0 || N < 1) return FUS_ERR_INVALID_ARGUMENT;
  if (nent == 0) return FUS_OK;
  if (!x || !consts || !y || !detJ || !dofmap) return FUS_ERR_INVALID_ARGUMENT;
  return hip_rc(fus::launch_mass<T>(x, consts, y, detJ, dofmap, N, nent, static_cast<hipStream_t>(stream)));
}

template <typename T, typename Op, bool UA, bool UB>
int ew(const T* a, const T* b, T* out, int64_t n, Op op, void* stream) {
  if (n < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (n == 0) return FUS_OK;
  if (!out || (UA && !a) || (UB && !b)) return FUS_ERR_INVALID_ARGUMENT;
  return hip_rc(fus::launch_ew<T, Op, UA, UB>(a, b, out, n, op, static_cast<hipStream_t>(stream)));
}

template <typename T, int MODE>
int halo(const T* in, T* out, const int64_t* index, int64_t count, int64_t offset, void* stream) {
  if (count < 0) return FUS_ERR_INVALID_ARGUMENT;
  if (count == 0) return FUS_OK;
  if (!in || !out || !index) return FUS_ERR_INVALID_ARGUMENT;
  return hip_rc(fus::launch_halo<T, MODE>(in, out, index, count, offset, static_cast<hipStream_t>(stream)));
}

template <typename T>
int mass_apply_planned(const T* x, const T* consts, T* y, const T* detJ, const void* ws, int N, int epb,
                              int64_t nent, void* stream) {
  if (nent < 0 || N < 2 || epb < 1 || (int64_t)N * epb > fus::kPlanMaxEntries) return FUS_ERR_INVALID_ARGUMENT;
  if (nent == 0) return FUS_OK;
  if (!x || !consts || !y || !detJ || !ws || misaligned(ws, 256)) return FUS_ERR_INVALID_ARGUMENT;
  bool ord = false, excl = false;
  bool rp = true;
  if (!plan_check(ws, N, epb, nent, &ord, &excl, &rp)) return FUS_ERR_PLAN_MISMATCH;
  return hip_rc(fus::launch_mass_plan<T>(x, consts, y, detJ, ws, N, epb, nent, static_cast<hipStream_t>(stream), ord, plan_use_runs<T>(N, rp), excl));
}

template <typename T>
int mass_apply_gather(const T* x, const T* c, T* y, const T* detJ, const void* ws, int N, int64_t nent, void* stream) {
  if (nent < 0 || N < 1) return FUS_ERR_INVALID_ARGUMENT;
  fus::GatherHeader h{};
  {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    auto it = g_gather_plans.find(ws);
    if (it == g_gather_plans.end() || it->second.N != N || it->second.nent != nent) return FUS_ERR_PLAN_MISMATCH;
    h = it->second;
  }
  if (nent == 0) return FUS_OK;
  if (!x || !c || !y || !detJ) return FUS_ERR_INVALID_ARGUMENT;
  return hip_rc(fus::launch_mass_gather<T>(x, c, y, detJ, ws, h, static_cast<hipStream_t>(stream),
                                           g_mass_variant.load(std::memory_order_relaxed)));
}

// static companions of transposed-dofmap plans (detJ in row order), keyed by their own workspace address
struct GatherStaticInfo {
  const void* plan;
  int elem_bytes;
};
std::unordered_map<const void*, GatherStaticInfo> g_gather_static;

template <typename T>
int mass_gather_static_build(const void* ws, const T* detJ, void* sws, int64_t sws_bytes, void* stream) {
  fus::GatherHeader h{};
  {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    auto it = g_gather_plans.find(ws);
    if (it == g_gather_plans.end()) return FUS_ERR_PLAN_MISMATCH;
    h = it->second;
  }
  if (!sws || misaligned(sws, 256) || (h.nent > 0 && !detJ)) return FUS_ERR_INVALID_ARGUMENT;
  if (sws_bytes < fus::gather_static_bytes(h.nent, (int)h.N, h.nent * h.N, (int)sizeof(T))) return FUS_ERR_INVALID_ARGUMENT;
  int too_wide = 0;
  const hipError_t e = fus::gather_static_build<T>(ws, h, detJ, sws, static_cast<hipStream_t>(stream), &too_wide);
  if (e != hipSuccess) return hip_rc(e);
  if (too_wide) return FUS_ERR_UNSUPPORTED_ENTITY;
  std::lock_guard<std::mutex> lk(g_plans_mu);
  g_gather_static[sws] = GatherStaticInfo{ws, (int)sizeof(T)};
  return FUS_OK;
}

template <typename T>
int mass_apply_gather_static(const T* x, const T* c, T* y, const void* ws, const void* sws, int N, int64_t nent, void* stream) {
  if (nent < 0 || N < 1) return FUS_ERR_INVALID_ARGUMENT;
  fus::GatherHeader h{};
  {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    auto it = g_gather_plans.find(ws);
    if (it == g_gather_plans.end() || it->second.N != N || it->second.nent != nent) return FUS_ERR_PLAN_MISMATCH;
    auto st = g_gather_static.find(sws);
    if (st == g_gather_static.end() || st->second.plan != ws || st->second.elem_bytes != (int)sizeof(T)) return FUS_ERR_PLAN_MISMATCH;
    h = it->second;
  }
  if (nent == 0) return FUS_OK;
  if (!x || !c || !y) return FUS_ERR_INVALID_ARGUMENT;
  return hip_rc(fus::launch_mass_gather_static<T>(x, c, y, ws, h, const_cast<void*>(sws), static_cast<hipStream_t>(stream),
                                                  g_mass_variant.load(std::memory_order_relaxed)));
}

}  // namespace

extern "C" {

int fus_abi_version(void) { return FUS_ABI_VERSION; }

#ifndef FUS_SOURCE_HASH
#define FUS_SOURCE_HASH "unknown"
#endif
const char* fus_source_hash(void) { return FUS_SOURCE_HASH; }

const char* fus_error_string(int code) {
  switch (code) {
    case FUS_OK: return "ok";
    case FUS_ERR_INVALID_ARGUMENT: return "invalid argument (null pointer, negative size or misaligned buffer)";
    case FUS_ERR_UNSUPPORTED_DEGREE: return "unsupported polynomial degree";
    case FUS_ERR_UNSUPPORTED_ENTITY: return "unsupported entity size";
    case FUS_ERR_NO_DEVICE: return "no HIP device";
    case FUS_ERR_PLAN_MISMATCH:
      return "workspace holds no plan built through this library for this (degree / entity size, entity count)";
    case FUS_ERR_COMM: return "communicator / RCCL failure (see fus_comm_last_error)";
    default:
      if (code <= FUS_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(FUS_ERR_HIP_BASE - code));
      return "unknown error";
  }
}

int fus_device_info(int device, char* name, int* compute_units, int64_t* hbm_bytes, int* lds_bytes_per_cu) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return FUS_ERR_NO_DEVICE;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return FUS_ERR_NO_DEVICE;
  if (name) {
    std::snprintf(name, 256, "%s (%s)", p.name, p.gcnArchName);
  }
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
  return FUS_OK;
}

int fus_set_tuning(int key, int value) {
  switch (key) {
    case FUS_TUNE_STIFFNESS_VARIANT: g_stiffness_variant = value; return FUS_OK;
    case FUS_TUNE_XCD_REMAP: g_xcd_remap = value ? 1 : 0; return FUS_OK;
    case FUS_TUNE_MASS_VARIANT: g_mass_variant = value; return FUS_OK;
    case FUS_TUNE_PLAN_VARIANT: g_plan_variant = value; return FUS_OK;
    case FUS_TUNE_PLAN_RUNS: g_plan_runs = value; return FUS_OK;
    case FUS_TUNE_VECTOR_STREAM:
      if (value < 0 || value > 4) return FUS_ERR_INVALID_ARGUMENT;
      fus::vector_stream_mode() = value;
      return FUS_OK;
  }
  return FUS_ERR_INVALID_ARGUMENT;
}

int fus_get_tuning(int key) {
  switch (key) {
    case FUS_TUNE_STIFFNESS_VARIANT: return g_stiffness_variant;
    case FUS_TUNE_XCD_REMAP: return g_xcd_remap;
    case FUS_TUNE_MASS_VARIANT: return g_mass_variant;
    case FUS_TUNE_PLAN_VARIANT: return g_plan_variant;
    case FUS_TUNE_PLAN_RUNS: return g_plan_runs;
    case FUS_TUNE_VECTOR_STREAM: return fus::vector_stream_mode();
  }
  return FUS_ERR_INVALID_ARGUMENT;
}

int fus_stiffness_apply_f64(const double* x, const double* cc, double* y, const double* G, const int32_t* dofmap,
                            const double* dphi, int P, int64_t ncell, void* stream) {
  return stiffness_apply<double>(x, cc, y, G, dofmap, dphi, P, ncell, stream);
}
int fus_stiffness_apply_f32(const float* x, const float* cc, float* y, const float* G, const int32_t* dofmap,
                            const float* dphi, int P, int64_t ncell, void* stream) {
  return stiffness_apply<float>(x, cc, y, G, dofmap, dphi, P, ncell, stream);
}

int64_t fus_stiffness_plan_bytes(int P, int64_t ncell) {
  if (ncell < 0) return FUS_ERR_INVALID_ARGUMENT;
  return plan_bytes(P, ncell);
}

int fus_stiffness_plan_build(const int32_t* dofmap, int P, int64_t ncell, void* workspace, int64_t workspace_bytes,
                             void* stream) {
  if (P < FUS_MIN_DEGREE || P > FUS_MAX_DEGREE) return FUS_ERR_UNSUPPORTED_DEGREE;
  return fus_plan_build_ordered(dofmap, nullptr, (P + 1) * (P + 1) * (P + 1), cells_per_batch(P), ncell, workspace,
                                workspace_bytes, stream);
}

int fus_stiffness_apply_planned_f64(const double* x, const double* cc, double* y, const double* G, const void* ws,
                                    const double* dphi, int P, int64_t ncell, void* stream) {
  return stiffness_apply_planned<double>(x, cc, y, G, ws, dphi, P, ncell, stream);
}
int fus_stiffness_apply_planned_f32(const float* x, const float* cc, float* y, const float* G, const void* ws,
                                    const float* dphi, int P, int64_t ncell, void* stream) {
  return stiffness_apply_planned<float>(x, cc, y, G, ws, dphi, P, ncell, stream);
}

int fus_plan_entities_per_batch(int N) {
  if (N < 1 || N > fus::kPlanMaxEntries) return FUS_ERR_UNSUPPORTED_ENTITY;
  // cells (N = n^3): the stiffness kernel's batch size, so one plan serves both operators
  for (int P = FUS_MIN_DEGREE; P <= FUS_MAX_DEGREE; ++P)
    if ((P + 1) * (P + 1) * (P + 1) == N) {
      return 256 / ((P + 1) * (P + 1)) > 0 ? 256 / ((P + 1) * (P + 1)) : 1;
    }
  const int epb = 1280 / N;  // ~5 entries per thread of a 256-thread workgroup
  return epb > 0 ? epb : 1;
}

int64_t fus_plan_bytes(int N, int entities_per_batch, int64_t nent) {
  if (N < 1 || entities_per_batch < 1 || nent < 0 || (int64_t)N * entities_per_batch > fus::kPlanMaxEntries)
    return FUS_ERR_INVALID_ARGUMENT;
  return fus::plan_view_generic(nullptr, N, entities_per_batch, nent).bytes;
}

int fus_plan_build(const int32_t* dofmap, int N, int entities_per_batch, int64_t nent, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  return fus_plan_build_ordered(dofmap, nullptr, N, entities_per_batch, nent, workspace, workspace_bytes, stream);
}

int fus_plan_release(const void* workspace) {
  std::lock_guard<std::mutex> lk(g_plans_mu);
  g_plans.erase(workspace);
  g_gather_plans.erase(workspace);
  g_gather_static.erase(workspace);
  for (auto it = g_gather_static.begin(); it != g_gather_static.end();)  // companions of a released plan go with it
    it = (it->second.plan == workspace) ? g_gather_static.erase(it) : std::next(it);
  return FUS_OK;
}

int64_t fus_mass_gather_plan_bytes(int N, int64_t nent, int64_t ndofs) {
  if (N < 1 || N > 2048 || nent < 0 || ndofs < 0 || nent * (int64_t)N > (int64_t)INT32_MAX) return FUS_ERR_INVALID_ARGUMENT;
  fus::GatherHeader h{};
  fus::gather_layout(nent, N, ndofs, &h);
  return h.bytes;
}

int fus_mass_gather_plan_build(const int32_t* dofmap, int N, int64_t nent, int64_t ndofs, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  const int64_t need = fus_mass_gather_plan_bytes(N, nent, ndofs);
  if (need < 0) return (int)need;
  if (!workspace || misaligned(workspace, 256) || workspace_bytes < need) return FUS_ERR_INVALID_ARGUMENT;
  if (nent > 0 && !dofmap) return FUS_ERR_INVALID_ARGUMENT;
  fus::GatherHeader h{};
  int bad = 0;
  const hipError_t e = fus::gather_plan_build(dofmap, N, nent, ndofs, workspace, static_cast<hipStream_t>(stream), &h, &bad);
  if (e != hipSuccess) return hip_rc(e);
  if (bad) return FUS_ERR_UNSUPPORTED_ENTITY;
  std::lock_guard<std::mutex> lk(g_plans_mu);
  g_gather_plans[workspace] = h;
  return FUS_OK;
}

int fus_mass_gather_plan_build_rows(const int32_t* dofmap, int N, int64_t nent, int64_t ndofs, const uint8_t* row_set, int which,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  const int64_t need = fus_mass_gather_plan_bytes(N, nent, ndofs);
  if (need < 0) return (int)need;
  if (!workspace || misaligned(workspace, 256) || workspace_bytes < need || !row_set || which < 0 || which > 255)
    return FUS_ERR_INVALID_ARGUMENT;
  if (ndofs >= 0x7fffffffLL) return FUS_ERR_INVALID_ARGUMENT;  // the sentinel key of the dropped rows is ndofs itself
  if (nent > 0 && !dofmap) return FUS_ERR_INVALID_ARGUMENT;
  fus::GatherHeader h{};
  int bad = 0;
  const hipError_t e = fus::gather_plan_build(dofmap, N, nent, ndofs, workspace, static_cast<hipStream_t>(stream), &h, &bad, row_set, which);
  if (e != hipSuccess) return hip_rc(e);
  if (bad) return FUS_ERR_UNSUPPORTED_ENTITY;
  std::lock_guard<std::mutex> lk(g_plans_mu);
  g_gather_plans[workspace] = h;
  return FUS_OK;
}

int fus_mass_gather_plan_info(const void* workspace, int64_t* out4) {
  if (!workspace || !out4) return FUS_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_plans_mu);
  auto it = g_gather_plans.find(workspace);
  if (it == g_gather_plans.end()) return FUS_ERR_PLAN_MISMATCH;
  out4[0] = it->second.nrows;
  out4[1] = it->second.dense;
  out4[2] = it->second.max_len;
  out4[3] = it->second.bytes;
  return FUS_OK;
}


int64_t fus_mass_gather_static_bytes(int N, int64_t nent, int elem_bytes) {
  if (N < 1 || nent < 0 || (elem_bytes != 4 && elem_bytes != 8)) return FUS_ERR_INVALID_ARGUMENT;
  if (nent * (int64_t)N >= (int64_t)1 << 31) return FUS_ERR_INVALID_ARGUMENT;
  return fus::gather_static_bytes(nent, N, nent * (int64_t)N, elem_bytes);
}
int fus_mass_gather_static_build_f64(const void* ws, const double* detJ, void* sws, int64_t sws_bytes, void* stream) {
  return mass_gather_static_build<double>(ws, detJ, sws, sws_bytes, stream);
}
int fus_mass_gather_static_build_f32(const void* ws, const float* detJ, void* sws, int64_t sws_bytes, void* stream) {
  return mass_gather_static_build<float>(ws, detJ, sws, sws_bytes, stream);
}
int fus_mass_apply_gather_static_f64(const double* x, const double* c, double* y, const void* ws, const void* sws, int N, int64_t nent,
                                     void* stream) {
  return mass_apply_gather_static<double>(x, c, y, ws, sws, N, nent, stream);
}
int fus_mass_apply_gather_static_f32(const float* x, const float* c, float* y, const void* ws, const void* sws, int N, int64_t nent,
                                     void* stream) {
  return mass_apply_gather_static<float>(x, c, y, ws, sws, N, nent, stream);
}

int fus_mass_apply_gather_f64(const double* x, const double* c, double* y, const double* detJ, const void* ws, int N,
                              int64_t nent, void* stream) {
  return mass_apply_gather<double>(x, c, y, detJ, ws, N, nent, stream);
}
int fus_mass_apply_gather_f32(const float* x, const float* c, float* y, const float* detJ, const void* ws, int N,
                              int64_t nent, void* stream) {
  return mass_apply_gather<float>(x, c, y, detJ, ws, N, nent, stream);
}

int fus_plan_build_ordered(const int32_t* dofmap, const int32_t* entity_order, int N, int entities_per_batch,
                           int64_t nent, void* workspace, int64_t workspace_bytes, void* stream) {
  const int64_t need = fus_plan_bytes(N, entities_per_batch, nent);
  if (need < 0) return (int)need;
  if (!workspace || misaligned(workspace, 256) || workspace_bytes < need) return FUS_ERR_INVALID_ARGUMENT;
  if (nent > 0 && !dofmap) return FUS_ERR_INVALID_ARGUMENT;
  if (nent > 0) {
    const hipError_t e = fus::launch_plan_build_generic(dofmap, N, entities_per_batch, nent, workspace,
                                                        static_cast<hipStream_t>(stream), plan_allow_runs(N), entity_order);
    if (e != hipSuccess) return hip_rc(e);
  }
  int64_t nbatch = 0, with_runs = 0;
  if (nent > 0) {
    const hipError_t e = fus::plan_run_batches(workspace, static_cast<hipStream_t>(stream), &with_runs);
    if (e != hipSuccess) return hip_rc(e);
    nbatch = (nent + entities_per_batch - 1) / entities_per_batch;
  }
  plan_register(workspace, N, entities_per_batch, nent, entity_order != nullptr, nbatch, with_runs);
  return FUS_OK;
}

int fus_plan_encoding(const void* workspace, int64_t* batches, int64_t* batches_with_runs, int* reads_runs_f64, int* reads_runs_f32) {
  PlanInfo p;
  {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    auto it = g_plans.find(workspace);
    if (it == g_plans.end()) return FUS_ERR_PLAN_MISMATCH;
    p = it->second;
  }
  if (batches) *batches = p.nbatch;
  if (batches_with_runs) *batches_with_runs = p.with_runs;
  if (reads_runs_f64) *reads_runs_f64 = plan_use_runs<double>(p.N, p.runs_pay) ? 1 : 0;
  if (reads_runs_f32) *reads_runs_f32 = plan_use_runs<float>(p.N, p.runs_pay) ? 1 : 0;
  return FUS_OK;
}

int fus_plan_mark_exclusive(void* workspace, int N, int entities_per_batch, int64_t nent, int32_t* dof_use_count,
                            int64_t ndofs, void* stream) {
  if (!workspace || !dof_use_count || ndofs < 0) return FUS_ERR_INVALID_ARGUMENT;
  bool ord = false;
  if (!plan_check(workspace, N, entities_per_batch, nent, &ord)) return FUS_ERR_PLAN_MISMATCH;
  const hipError_t e = fus::launch_plan_mark_exclusive(workspace, N, entities_per_batch, nent, dof_use_count, ndofs,
                                                       static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_rc(e);
  std::lock_guard<std::mutex> lk(g_plans_mu);
  g_plans[workspace].exclusive = true;
  return FUS_OK;
}

int fus_mass_apply_planned_f64(const double* x, const double* c, double* y, const double* detJ, const void* ws, int N,
                               int epb, int64_t nent, void* stream) {
  return mass_apply_planned<double>(x, c, y, detJ, ws, N, epb, nent, stream);
}
int fus_mass_apply_planned_f32(const float* x, const float* c, float* y, const float* detJ, const void* ws, int N,
                               int epb, int64_t nent, void* stream) {
  return mass_apply_planned<float>(x, c, y, detJ, ws, N, epb, nent, stream);
}

int fus_mass_apply_f64(const double* x, const double* c, double* y, const double* detJ, const int32_t* dofmap, int N,
                       int64_t nent, void* stream) {
  return mass_apply<double>(x, c, y, detJ, dofmap, N, nent, stream);
}
int fus_mass_apply_f32(const float* x, const float* c, float* y, const float* detJ, const int32_t* dofmap, int N,
                       int64_t nent, void* stream) {
  return mass_apply<float>(x, c, y, detJ, dofmap, N, nent, stream);
}

#define FUS_FACET(T, SUF)                                                                                        \
  int fus_facet_terms_##SUF(T* y, const T* cA1, T sA1, const T* cA2, T sA2, const T* detJA, const int32_t* dmA,   \
                            int64_t nentA, const T* xB, const T* cB, const T* detJB, const int32_t* dmB,          \
                            int64_t nentB, int N, void* s) {                                                      \
    if (nentA < 0 || nentB < 0 || N < 1 || !y) return FUS_ERR_INVALID_ARGUMENT;                                   \
    if (nentA > 0 && (!cA1 || !detJA || !dmA)) return FUS_ERR_INVALID_ARGUMENT;                                   \
    if (nentB > 0 && (!xB || !cB || !detJB || !dmB)) return FUS_ERR_INVALID_ARGUMENT;                             \
    return hip_rc(fus::launch_facet_terms<T>(y, cA1, sA1, cA2, sA2, detJA, dmA, nentA, xB, cB, detJB, dmB, nentB, \
                                             N, static_cast<hipStream_t>(s)));                                    \
  }                                                                                                               \
  int fus_facet_terms_dev_##SUF(T* y, const T* cA1, const T* cA2, const T* scalars, const T* detJA,               \
                                const int32_t* dmA, int64_t nentA, const T* xB, const T* cB, const T* detJB,      \
                                const int32_t* dmB, int64_t nentB, int N, void* s) {                              \
    if (nentA < 0 || nentB < 0 || N < 1 || !y || !scalars) return FUS_ERR_INVALID_ARGUMENT;                       \
    if (nentA > 0 && (!cA1 || !detJA || !dmA)) return FUS_ERR_INVALID_ARGUMENT;                                   \
    if (nentB > 0 && (!xB || !cB || !detJB || !dmB)) return FUS_ERR_INVALID_ARGUMENT;                             \
    return hip_rc(fus::launch_facet_terms<T>(y, cA1, T(0), cA2, T(0), detJA, dmA, nentA, xB, cB, detJB, dmB,      \
                                             nentB, N, static_cast<hipStream_t>(s), scalars));                    \
  }
FUS_FACET(double, f64)
FUS_FACET(float, f32)
#undef FUS_FACET

#define FUS_VEC(T, SUF)                                                                                       \
  int fus_axpy_##SUF(T alpha, const T* x, T* y, int64_t n, void* s) {                                         \
    return ew<T, fus::OpAxpy<T>, true, true>(x, y, y, n, fus::OpAxpy<T>{alpha}, s);                           \
  }                                                                                                           \
  int fus_scale_##SUF(T alpha, const T* a, T* b, int64_t n, void* s) {                                        \
    return ew<T, fus::OpScale<T>, true, false>(a, nullptr, b, n, fus::OpScale<T>{alpha}, s);                  \
  }                                                                                                           \
  int fus_copy_##SUF(const T* a, T* b, int64_t n, void* s) {                                                  \
    return ew<T, fus::OpCopy<T>, true, false>(a, nullptr, b, n, fus::OpCopy<T>{}, s);                         \
  }                                                                                                           \
  int fus_fill_##SUF(T alpha, T* x, int64_t n, void* s) {                                                     \
    return ew<T, fus::OpFill<T>, false, false>(nullptr, nullptr, x, n, fus::OpFill<T>{alpha}, s);             \
  }                                                                                                           \
  int fus_pointwise_divide_##SUF(const T* a, const T* b, T* c, int64_t n, void* s) {                          \
    return ew<T, fus::OpDiv<T>, true, true>(a, b, c, n, fus::OpDiv<T>{}, s);                                  \
  }                                                                                                           \
  int fus_square_##SUF(const T* a, T* b, int64_t n, void* s) {                                                \
    return ew<T, fus::OpSquare<T>, true, false>(a, nullptr, b, n, fus::OpSquare<T>{}, s);                     \
  }                                                                                                           \
  int fus_muladd_##SUF(const T* w, const T* x, T* y, int64_t n, void* s) {                                    \
    if (n < 0) return FUS_ERR_INVALID_ARGUMENT;                                                               \
    if (n == 0) return FUS_OK;                                                                                \
    if (!w || !x || !y) return FUS_ERR_INVALID_ARGUMENT;                                                      \
    return hip_rc(fus::launch_muladd<T>(w, x, y, n, static_cast<hipStream_t>(s)));                            \
  }                                                                                                           \
  int fus_pack_fwd_##SUF(const T* in, T* out, const int64_t* idx, int64_t cnt, void* s) {                     \
    return halo<T, fus::PACK>(in, out, idx, cnt, 0, s);                                                       \
  }                                                                                                           \
  int fus_unpack_fwd_##SUF(const T* in, T* out, const int64_t* idx, int64_t cnt, int64_t N, void* s) {        \
    return halo<T, fus::UNPACK_SET>(in, out, idx, cnt, N, s);                                                 \
  }                                                                                                           \
  int fus_pack_rev_##SUF(const T* in, T* out, const int64_t* idx, int64_t cnt, int64_t N, void* s) {          \
    return halo<T, fus::PACK>(in, out, idx, cnt, N, s);                                                       \
  }                                                                                                           \
  int fus_unpack_rev_##SUF(const T* in, T* out, const int64_t* idx, int64_t cnt, void* s) {                   \
    return halo<T, fus::UNPACK_ADD>(in, out, idx, cnt, 0, s);                                                 \
  }
FUS_VEC(double, f64)
FUS_VEC(float, f32)

#define FUS_GEOM(T, SUF)                                                                                         \
  int fus_geometry_factors_##SUF(const T* x_g, const int32_t* x_dofs, const T* dphi, const T* weights, int nq,   \
                                 int64_t ncell, T* G, T* detJ, void* s) {                                        \
    if (ncell < 0 || nq < 1) return FUS_ERR_INVALID_ARGUMENT;                                                    \
    if (ncell == 0) return FUS_OK;                                                                               \
    if (!x_g || !x_dofs || !dphi || !weights || (!G && !detJ)) return FUS_ERR_INVALID_ARGUMENT;                  \
    return hip_rc(fus::launch_geometry<T>(x_g, x_dofs, dphi, weights, nq, ncell, G, detJ,                        \
                                          static_cast<hipStream_t>(s)));                                         \
  }                                                                                                              \
  int fus_facet_jacobian_##SUF(const T* x_g, const int32_t* x_dofs, const int32_t* boundary_data, const T* dphi_f, \
                               const T* weights, int nqf, int64_t nfacets, T* detJ_f, void* s) {                 \
    if (nfacets < 0 || nqf < 1) return FUS_ERR_INVALID_ARGUMENT;                                                 \
    if (nfacets == 0) return FUS_OK;                                                                             \
    if (!x_g || !x_dofs || !boundary_data || !dphi_f || !weights || !detJ_f) return FUS_ERR_INVALID_ARGUMENT;    \
    return hip_rc(fus::launch_facet_geometry<T>(x_g, x_dofs, boundary_data, dphi_f, weights, nqf, nfacets,       \
                                                detJ_f, static_cast<hipStream_t>(s)));                           \
  }
FUS_GEOM(double, f64)
FUS_GEOM(float, f32)
#undef FUS_GEOM

#define FUS_AFFINE(T, SUF)                                                                                       \
  int fus_stiffness_apply_planned_affine_##SUF(const T* x, const T* cc, T* y, const T* G, const T* wratio,       \
                                               const void* ws, const T* dphi, int P, int64_t ncell, void* s) {   \
    return stiffness_apply_planned_affine<T>(x, cc, y, G, wratio, ws, dphi, P, ncell, s);                        \
  }
FUS_AFFINE(double, f64)
FUS_AFFINE(float, f32)
#undef FUS_AFFINE

#define FUS_GEOMK(T, SUF)                                                                                        \
  int fus_stiffness_apply_planned_geom_##SUF(const T* x, const T* cc, T* y, const T* x_g, const int32_t* x_dofs, \
                                             const T* pts, const T* wts, const void* ws, const T* dphi, int P,   \
                                             int64_t ncell, void* s) {                                           \
    return stiffness_apply_planned_geom<T>(x, cc, y, x_g, x_dofs, pts, wts, ws, dphi, P, ncell, s);              \
  }
FUS_GEOMK(double, f64)
FUS_GEOMK(float, f32)
#undef FUS_GEOMK

#define FUS_WEST(T, SUF)                                                                                          \
  int fus_westervelt_cell_apply_planned_##SUF(const T* u, const T* v, const T* c2, const T* c3, const T* c4,      \
                                              const T* c5, T* b, T* m, const T* G, const T* detJ, const void* ws, \
                                              const T* dphi, int P, int64_t ncell, void* s) {                     \
    return westervelt_cell<T>(u, v, c2, c3, c4, c5, b, m, G, detJ, ws, dphi, P, ncell, s);                        \
  }                                                                                                               \
  int fus_rk4_stage_nl_##SUF(T bw, T aw, int new_step, const T* m0, T* m, T* b, T* u, T* v, T* u0, T* v0, T* ku,  \
                             T* un, int64_t nlocal, int64_t ntotal, void* s) {                                    \
    if (nlocal < 0 || ntotal < nlocal) return FUS_ERR_INVALID_ARGUMENT;                                           \
    if (ntotal == 0) return FUS_OK;                                                                               \
    if (!m0 || !m || !b || !u || !v || !u0 || !v0 || !ku || !un) return FUS_ERR_INVALID_ARGUMENT;                 \
    return hip_rc(fus::launch_rk4_stage_nl<T>(bw, aw, new_step, m0, m, b, u, v, u0, v0, ku, un, nlocal, ntotal,   \
                                              static_cast<hipStream_t>(s)));                                      \
  }
FUS_WEST(double, f64)
FUS_WEST(float, f32)
#define FUS_NL2(T, SUF)                                                                                            \
  int fus_rk4_stage_nl2_##SUF(T bw, T aw, int new_step, const T* m0, const T* w2, const T* w5, T* b, T* u, T* v,   \
                              T* u0, T* v0, T* ku, T* un, T kappa, T* w, int64_t nlocal, int64_t ntotal, void* s) { \
    if (nlocal < 0 || ntotal < nlocal) return FUS_ERR_INVALID_ARGUMENT;                                            \
    if (ntotal == 0) return FUS_OK;                                                                                \
    if (!m0 || !w2 || !w5 || !b || !u || !v || !u0 || !v0 || !ku || !un) return FUS_ERR_INVALID_ARGUMENT;          \
    if (new_step < 0 || new_step > 7) return FUS_ERR_INVALID_ARGUMENT;                                             \
    return hip_rc(fus::launch_rk4_stage_nl2<T>(bw, aw, new_step, m0, w2, w5, b, u, v, u0, v0, ku, un, kappa, w,    \
                                               nlocal, ntotal, static_cast<hipStream_t>(s)));                      \
  }
FUS_NL2(double, f64)
FUS_NL2(float, f32)
#undef FUS_NL2
#define FUS_WESTG(T, SUF)                                                                                          \
  int fus_westervelt_cell_apply_planned_geom_##SUF(const T* u, const T* v, const T* c2, const T* c3, const T* c4,  \
                                                   const T* c5, T* b, T* m, const T* x_g, const int32_t* x_dofs,   \
                                                   const T* pts, const T* wts, const void* ws, const T* dphi,      \
                                                   int P, int64_t ncell, void* s) {                                \
    return westervelt_cell_geom<T>(u, v, c2, c3, c4, c5, b, m, x_g, x_dofs, pts, wts, ws, dphi, P, ncell, s);      \
  }
FUS_WESTG(double, f64)
FUS_WESTG(float, f32)
#undef FUS_WESTG
#undef FUS_WEST

#define FUS_RK4(T, SUF)                                                                                        \
  int fus_rk4_stage_##SUF(T bw, T aw, int new_step, const T* minv, T* b, T* u, T* v, T* u0, T* v0, T* ku, T* un, \
                          int64_t nlocal, int64_t ntotal, void* s) {                                           \
    if (nlocal < 0 || ntotal < nlocal) return FUS_ERR_INVALID_ARGUMENT;                                        \
    if (ntotal == 0) return FUS_OK;                                                                            \
    if (!minv || !b || !u || !v || !u0 || !v0 || !ku || !un) return FUS_ERR_INVALID_ARGUMENT;                  \
    if (new_step < 0 || new_step > 7) return FUS_ERR_INVALID_ARGUMENT;                                         \
    return hip_rc(fus::launch_rk4_stage<T>(bw, aw, new_step, minv, b, u, v, u0, v0, ku, un, nlocal, ntotal,    \
                                           static_cast<hipStream_t>(s)));                                      \
  }
FUS_RK4(double, f64)
FUS_RK4(float, f32)
#undef FUS_RK4
#undef FUS_VEC

// ------------------------------------------------------------------ communicator + halo exchange
struct fus_comm {
  fus::Comm c;
};
struct fus_halo {
  fus::Halo h;
};

static std::string g_comm_error;  // errors that happen before a communicator exists

int fus_comm_unique_id(void* id) {
  if (!id) return FUS_ERR_INVALID_ARGUMENT;
  fus::RcclApi& api = fus::rccl();
  if (!api.load()) {
    g_comm_error = api.error;
    return FUS_ERR_COMM;
  }
  ncclUniqueId uid;
  static_assert(sizeof(uid) == FUS_UNIQUE_ID_BYTES, "unique id size");
  const ncclResult_t r = api.GetUniqueId(&uid);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclGetUniqueId: ") + api.GetErrorString(r);
    return FUS_ERR_COMM;
  }
  std::memcpy(id, &uid, sizeof(uid));
  return FUS_OK;
}

int fus_comm_create(const void* id, int nranks, int rank, fus_comm_t* out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return FUS_ERR_INVALID_ARGUMENT;
  fus::RcclApi& api = fus::rccl();
  if (!api.load()) {
    g_comm_error = api.error;
    return FUS_ERR_COMM;
  }
  auto* c = new fus_comm;
  c->c.kind = fus::Comm::RCCL;
  c->c.rank = rank;
  c->c.nranks = nranks;
  hipError_t e = fus::comm_make_stream_and_sync(&c->c);
  if (e != hipSuccess) {
    delete c;
    return hip_rc(e);
  }
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = api.CommInitRank(&c->c.nccl, nranks, uid, rank);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclCommInitRank: ") + api.GetErrorString(r);
    (void)hipStreamDestroy(c->c.stream);
    delete c;
    return FUS_ERR_COMM;
  }
  *out = c;
  return FUS_OK;
}

int fus_comm_create_peer(int nranks, int rank, fus_comm_t* out) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return FUS_ERR_INVALID_ARGUMENT;
  auto* c = new fus_comm;
  c->c.kind = fus::Comm::PEER;
  c->c.rank = rank;
  c->c.nranks = nranks;
  hipError_t e = fus::comm_make_stream_and_sync(&c->c);
  if (e != hipSuccess) {
    if (c->c.stream) (void)hipStreamDestroy(c->c.stream);
    delete c;
    return hip_rc(e);
  }
  *out = c;
  return FUS_OK;
}

int fus_comm_create_local(int world_id, int nranks, int rank, fus_comm_t* out) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return FUS_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(fus::local_worlds_mutex());
  auto& worlds = fus::local_worlds();
  std::shared_ptr<fus::LocalWorld> w = worlds[world_id].lock();
  if (!w) {
    w = std::make_shared<fus::LocalWorld>();
    w->nranks = nranks;
    w->halos.resize(nranks);
    worlds[world_id] = w;
  }
  if (w->nranks != nranks) return FUS_ERR_INVALID_ARGUMENT;
  auto* c = new fus_comm;
  c->c.kind = fus::Comm::LOCAL;
  c->c.rank = rank;
  c->c.nranks = nranks;
  c->c.world = w;
  hipError_t e = fus::comm_make_stream_and_sync(&c->c);
  if (e != hipSuccess) {
    delete c;
    return hip_rc(e);
  }
  *out = c;
  return FUS_OK;
}

int fus_comm_rank(fus_comm_t comm) { return comm ? comm->c.rank : FUS_ERR_INVALID_ARGUMENT; }
int fus_comm_size(fus_comm_t comm) { return comm ? comm->c.nranks : FUS_ERR_INVALID_ARGUMENT; }
void* fus_comm_stream(fus_comm_t comm) { return comm ? comm->c.stream : nullptr; }

static int comm_fork_join_rc(fus_comm_t comm, void* stream, int which, bool lazy, bool attach = false) {
  if (!comm) return FUS_ERR_INVALID_ARGUMENT;
  bool misuse = false;
  const hipError_t e = fus::comm_fork_join(&comm->c, static_cast<hipStream_t>(stream), which, lazy, &misuse, attach);
  return misuse ? FUS_ERR_INVALID_ARGUMENT : hip_rc(e);
}
int fus_comm_fork(fus_comm_t comm, void* stream) { return comm_fork_join_rc(comm, stream, 0, false); }
int fus_comm_fork_lazy(fus_comm_t comm, void* stream) { return comm_fork_join_rc(comm, stream, 0, true); }
int fus_comm_fork_ex(fus_comm_t comm, void* stream, int flags) {
  if (flags & ~(FUS_FORK_LAZY | FUS_FORK_ATTACH)) return FUS_ERR_INVALID_ARGUMENT;
  return comm_fork_join_rc(comm, stream, 0, (flags & FUS_FORK_LAZY) != 0, (flags & FUS_FORK_ATTACH) != 0);
}
int fus_comm_fork_flush(fus_comm_t comm) {
  if (!comm) return FUS_ERR_INVALID_ARGUMENT;
  return hip_rc(fus::comm_flush_attached(&comm->c));
}
int fus_comm_join(fus_comm_t comm, void* stream) { return comm_fork_join_rc(comm, stream, 1, false); }
int fus_comm_arm_join(fus_comm_t comm) {
  if (!comm) return FUS_ERR_INVALID_ARGUMENT;
  return hip_rc(fus::comm_arm_join(&comm->c));
}
int fus_comm_health(fus_comm_t comm, int64_t* failures) {
  if (!comm || !failures) return FUS_ERR_INVALID_ARGUMENT;
  return fus::comm_health(&comm->c, failures) == 0 ? FUS_OK : FUS_ERR_COMM;
}
int fus_comm_health_detail(fus_comm_t comm, int64_t* out3) {
  if (!comm || !out3) return FUS_ERR_INVALID_ARGUMENT;
  int64_t total = 0;
  return fus::comm_health(&comm->c, &total, out3) == 0 ? FUS_OK : FUS_ERR_COMM;
}
int fus_comm_sync_timeouts(fus_comm_t comm, int64_t* out) {
  if (!comm || !out) return FUS_ERR_INVALID_ARGUMENT;
  *out = 0;
  if (!comm->c.sync_words) return FUS_OK;
  uint64_t w = 0;
  hipError_t e = hipStreamSynchronize(comm->c.stream);
  if (e == hipSuccess) e = hipMemcpy(&w, comm->c.sync_words + 2 + fus::ST_TIMEOUTS, sizeof w, hipMemcpyDeviceToHost);
  *out = (int64_t)w;
  return hip_rc(e);
}

const char* fus_comm_last_error(fus_comm_t comm) {
  return comm ? comm->c.last_error.c_str() : g_comm_error.c_str();
}

int fus_comm_destroy(fus_comm_t comm) {
  if (!comm) return FUS_OK;
  if (comm->c.nhalos > 0) {  // a halo holds a pointer to its communicator: destroy the halos first
    comm->c.last_error = "fus_comm_destroy: " + std::to_string(comm->c.nhalos) + " halo object(s) of this communicator are still alive";
    return FUS_ERR_COMM;
  }
  (void)fus::comm_flush_attached(&comm->c);  // a fork signal still waiting for a launch to carry it must not outlive its flag
  if (comm->c.stream) (void)hipStreamSynchronize(comm->c.stream);
  if (comm->c.stream2) (void)hipStreamSynchronize(comm->c.stream2);
  if (comm->c.nccl) (void)fus::rccl().CommDestroy(comm->c.nccl);
  if (comm->c.stream2 && comm->c.stream2 != comm->c.stream) (void)hipStreamDestroy(comm->c.stream2);
  if (comm->c.stream) (void)hipStreamDestroy(comm->c.stream);
  if (comm->c.sync_words) (void)hipFree(comm->c.sync_words);
  delete comm;
  return FUS_OK;
}

int fus_halo_create(fus_comm_t comm, int elem_bytes, int64_t nlocal, int64_t nghost, int n_owner_ranks,
                    const int32_t* owner_ranks, const int64_t* owner_sizes, const int64_t* owners_idx,
                    int n_ghost_ranks, const int32_t* ghost_ranks, const int64_t* ghost_sizes,
                    const int64_t* ghosts_idx, fus_halo_t* out) {
  if (!comm || !out || (elem_bytes != 4 && elem_bytes != 8) || nlocal < 0 || nghost < 0 || n_owner_ranks < 0 ||
      n_ghost_ranks < 0)
    return FUS_ERR_INVALID_ARGUMENT;
  if ((n_owner_ranks > 0 && (!owner_ranks || !owner_sizes)) || (n_ghost_ranks > 0 && (!ghost_ranks || !ghost_sizes)))
    return FUS_ERR_INVALID_ARGUMENT;
  int64_t no = 0, ng = 0;
  for (int i = 0; i < n_owner_ranks; ++i) {
    if (owner_sizes[i] < 0 || owner_ranks[i] < 0 || owner_ranks[i] >= comm->c.nranks) return FUS_ERR_INVALID_ARGUMENT;
    no += owner_sizes[i];
  }
  for (int i = 0; i < n_ghost_ranks; ++i) {
    if (ghost_sizes[i] < 0 || ghost_ranks[i] < 0 || ghost_ranks[i] >= comm->c.nranks) return FUS_ERR_INVALID_ARGUMENT;
    ng += ghost_sizes[i];
  }
  if (no > nghost || (no > 0 && !owners_idx) || (ng > 0 && !ghosts_idx)) return FUS_ERR_INVALID_ARGUMENT;
  // out-of-range indices would fault in the pack / unpack kernels: check them here, once
  bool direct = no > 0;
  for (int64_t i = 0; i < no; ++i) {
    if (owners_idx[i] < 0 || owners_idx[i] >= nghost) return FUS_ERR_INVALID_ARGUMENT;
    if (owners_idx[i] != i) direct = false;
  }
  for (int64_t i = 0; i < ng; ++i)
    if (ghosts_idx[i] < 0 || ghosts_idx[i] >= nlocal) return FUS_ERR_INVALID_ARGUMENT;
  auto* hh = new fus_halo;
  fus::Halo& h = hh->h;
  h.comm = &comm->c;
  h.eb = elem_bytes;
  h.nlocal = nlocal;
  h.nghost = nghost;
  h.direct = direct;
  ++comm->c.nhalos;
  comm->c.halos.push_back(&h);
  const bool peer = comm->c.kind == fus::Comm::PEER;
  hipError_t e = fus::side_init(h.owners, n_owner_ranks, owner_ranks, owner_sizes, owners_idx, comm->c.stream);
  if (e == hipSuccess) e = fus::side_init(h.ghosts, n_ghost_ranks, ghost_ranks, ghost_sizes, ghosts_idx, comm->c.stream);
  if (e == hipSuccess && no > 0 && !peer) e = hipMalloc(&h.buf_owner, no * elem_bytes);
  if (e == hipSuccess && ng > 0 && !peer) e = hipMalloc(&h.buf_ghost, ng * elem_bytes);
  if (e == hipSuccess && peer) e = fus::halo_ipc_create(&h);
  for (hipEvent_t* ev : {&h.ev_ready, &h.ev_done, &h.ev_packed, &h.ev_pulled})
    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipStreamSynchronize(comm->c.stream);  // the index lists came from host arrays the caller may free
  if (e != hipSuccess) {
    fus_halo_destroy(hh);
    return hip_rc(e);
  }
  if (comm->c.kind == fus::Comm::LOCAL) {
    std::lock_guard<std::mutex> lock(fus::local_worlds_mutex());
    auto& mine = comm->c.world->halos[comm->c.rank];
    h.index = (int)mine.size();
    mine.push_back(&h);
  }
  *out = hh;
  return FUS_OK;
}

int fus_halo_destroy(fus_halo_t halo) {
  if (!halo) return FUS_OK;
  fus::Halo& h = halo->h;
  if (h.comm && h.comm->stream) (void)hipStreamSynchronize(h.comm->stream);
  if (h.comm && h.comm->stream2) (void)hipStreamSynchronize(h.comm->stream2);
  if (h.comm && h.comm->kind == fus::Comm::LOCAL && h.comm->world) {
    std::lock_guard<std::mutex> lock(fus::local_worlds_mutex());
    auto& mine = h.comm->world->halos[h.comm->rank];
    if (h.index < (int)mine.size() && mine[h.index] == &h) mine[h.index] = nullptr;
  }
  if (h.comm) {
    --h.comm->nhalos;
    auto& hv = h.comm->halos;
    hv.erase(std::remove(hv.begin(), hv.end(), &h), hv.end());
    if (h.comm->join_halo == &h) {
      h.comm->join_halo = nullptr;
      h.comm->join_armed = false;
    }
  }
  fus::halo_ipc_free(&h);
  fus::side_free(h.owners);
  fus::side_free(h.ghosts);
  if (h.buf_owner) (void)hipFree(h.buf_owner);
  if (h.buf_ghost) (void)hipFree(h.buf_ghost);
  for (hipEvent_t ev : {h.ev_ready, h.ev_done, h.ev_packed, h.ev_pulled})
    if (ev) (void)hipEventDestroy(ev);
  delete halo;
  return FUS_OK;
}

int fus_halo_is_direct(fus_halo_t halo) { return halo ? (halo->h.direct ? 1 : 0) : FUS_ERR_INVALID_ARGUMENT; }

int64_t fus_halo_ipc_blob_bytes(fus_halo_t halo) {
  if (!halo || halo->h.comm->kind != fus::Comm::PEER) return FUS_ERR_INVALID_ARGUMENT;
  return fus::halo_ipc_blob_bytes(&halo->h);
}
int fus_halo_ipc_export(fus_halo_t halo, void* blob) {
  if (!halo || !blob || halo->h.comm->kind != fus::Comm::PEER) return FUS_ERR_INVALID_ARGUMENT;
  return fus::halo_ipc_export(&halo->h, blob) == 0 ? FUS_OK : FUS_ERR_COMM;
}
int fus_halo_ipc_connect(fus_halo_t halo, int nblobs, const void* const* blobs) {
  if (!halo || nblobs < 0 || (nblobs > 0 && !blobs) || halo->h.comm->kind != fus::Comm::PEER) return FUS_ERR_INVALID_ARGUMENT;
  return fus::halo_ipc_connect(&halo->h, nblobs, blobs) == 0 ? FUS_OK : FUS_ERR_COMM;
}
int fus_halo_ipc_status(fus_halo_t halo, int64_t* out8) {
  if (!halo || !out8 || halo->h.comm->kind != fus::Comm::PEER) return FUS_ERR_INVALID_ARGUMENT;
  return fus::halo_ipc_status(&halo->h, out8) == 0 ? FUS_OK : FUS_ERR_COMM;
}

#define FUS_HALO_OP(NAME, FN, DIR)                                                  \
  int NAME(fus_halo_t halo, void* buffer, void* stream) {                           \
    if (!halo || !buffer) return FUS_ERR_INVALID_ARGUMENT;                          \
    return fus::FN(&halo->h, buffer, static_cast<hipStream_t>(stream), DIR) == 0 ? FUS_OK : FUS_ERR_COMM; \
  }
FUS_HALO_OP(fus_halo_forward_begin, halo_begin, 0)
FUS_HALO_OP(fus_halo_forward_end, halo_end, 0)
FUS_HALO_OP(fus_halo_reverse_begin, halo_begin, 1)
FUS_HALO_OP(fus_halo_reverse_end, halo_end, 1)
#undef FUS_HALO_OP

#define FUS_HALO_GROUP(NAME, DIR)                                                                          \
  int NAME(const fus_halo_t* halos, void* const* buffers, int n, void* stream) {                           \
    if (n < 0 || n > 8 || (n > 0 && (!halos || !buffers))) return FUS_ERR_INVALID_ARGUMENT;                \
    fus::Halo* hs[8];                                                                                      \
    for (int k = 0; k < n; ++k) {                                                                          \
      if (!halos[k] || !buffers[k]) return FUS_ERR_INVALID_ARGUMENT;                                       \
      hs[k] = &halos[k]->h;                                                                                \
    }                                                                                                      \
    return fus::halo_begin_group(hs, buffers, n, static_cast<hipStream_t>(stream), DIR) == 0 ? FUS_OK : FUS_ERR_COMM; \
  }
FUS_HALO_GROUP(fus_halo_forward_begin_group, 0)
FUS_HALO_GROUP(fus_halo_reverse_begin_group, 1)
#undef FUS_HALO_GROUP

int fus_halo_forward(fus_halo_t halo, void* buffer, void* stream) {
  if (!halo || !buffer) return FUS_ERR_INVALID_ARGUMENT;
  const int r = fus::halo_exchange_inline(&halo->h, buffer, static_cast<hipStream_t>(stream), 0);  // PEER: on the caller's stream
  if (r <= 0) return r == 0 ? FUS_OK : FUS_ERR_COMM;
  const int rc = fus_halo_forward_begin(halo, buffer, stream);
  return rc != FUS_OK ? rc : fus_halo_forward_end(halo, buffer, stream);
}
int fus_halo_reverse(fus_halo_t halo, void* buffer, void* stream) {
  if (!halo || !buffer) return FUS_ERR_INVALID_ARGUMENT;
  const int r = fus::halo_exchange_inline(&halo->h, buffer, static_cast<hipStream_t>(stream), 1);
  if (r <= 0) return r == 0 ? FUS_OK : FUS_ERR_COMM;
  const int rc = fus_halo_reverse_begin(halo, buffer, stream);
  return rc != FUS_OK ? rc : fus_halo_reverse_end(halo, buffer, stream);
}

}  // extern "C"
