// Mass operator apply (GLL collocation => diagonal per entity):
//   y[dofmap[e][i]] += x[dofmap[e][i]] * detJ[e][i] * entity_constants[e]
// replaces numba-cpu/operators.py:50-66 and cuda/operators.py:18-70.
// One thread per (entity, local dof); detJ / dofmap are read fully coalesced, x is gathered,
// the scatter-add is the hardware FP atomic.  Used for cells (N = n^3) and boundary facets (N = n^2).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"
#include "stiffness_plan.hpp"

namespace fus {

template <typename T, typename I>
__global__ void __launch_bounds__(256)
    mass_kernel(const T* __restrict__ x, const T* __restrict__ entity_constants, T* __restrict__ y,
                const T* __restrict__ detJ, const int32_t* __restrict__ dofmap, I N, I total) {
  const I stride = (I)gridDim.x * 256;
  for (I idx = (I)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const I e = idx / N;
    const int32_t dof = dofmap[idx];
    const T v = x[dof] * (detJ[idx] * entity_constants[e]);
    unsafeAtomicAdd(y + dof, v);
  }
}

template <typename T>
inline hipError_t launch_mass(const T* x, const T* consts, T* y, const T* detJ, const int32_t* dofmap, int N,
                              int64_t nent, hipStream_t stream) {
  const int64_t total = nent * (int64_t)N;
  if (total <= 0) return hipSuccess;
  int64_t nblocks = (total + 255) / 256;
  const int64_t cap = 256LL * 64;  // grid-stride beyond 64 workgroups per CU
  if (nblocks > cap) nblocks = cap;
  if (total < 0x7fffffffLL) {
    hipLaunchKernelGGL((mass_kernel<T, uint32_t>), dim3((unsigned)nblocks), dim3(256), 0, stream, x, consts, y, detJ,
                       dofmap, (uint32_t)N, (uint32_t)total);
  } else {
    hipLaunchKernelGGL((mass_kernel<T, int64_t>), dim3((unsigned)nblocks), dim3(256), 0, stream, x, consts, y, detJ,
                       dofmap, (int64_t)N, total);
  }
  return hipGetLastError();
}

// The boundary-facet terms of one RK4 stage in ONE launch.  The reference launches, per stage,
//   mass(g, facet_coeff1) on the source facets (+ mass(dg, facet_coeff2_1) in the Westervelt solver) and
//   mass(v_n, facet_coeff2) on the absorbing facets          cuda/demo_linear_box.py:546-549,
//                                                            cuda/demo_nonlinear_bowl.py:633-641
// where g / dg are the source value and its derivative filled into whole vectors.  Here:
//   set A (x = 1):  y[dmA[e][i]] += (sA1 cA1[e] + sA2 cA2[e]) detJA[e][i]          (sA* = g(t), dg(t))
//   set B:          y[dmB[e][i]] += xB[dmB[e][i]] cB[e] detJB[e][i]
// a few thousand facets: launch-latency-bound, so one launch instead of three to five matters.
// ``sdev`` != nullptr: (sA1, sA2) are read from device memory instead of the launch arguments, so that a
// time loop captured in a hipGraph can be replayed with new source values (fus_facet_terms_dev_*).
template <typename T>
__global__ void __launch_bounds__(256)
    facet_terms_kernel(T* __restrict__ y, const T* __restrict__ cA1, T sA1, const T* __restrict__ cA2, T sA2,
                       const T* __restrict__ detJA, const int32_t* __restrict__ dmA, int64_t totalA,
                       const T* __restrict__ xB, const T* __restrict__ cB, const T* __restrict__ detJB,
                       const int32_t* __restrict__ dmB, int64_t totalB, int N, const T* __restrict__ sdev) {
  if (sdev != nullptr) {
    sA1 = sdev[0];
    sA2 = sdev[1];
  }
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < totalA + totalB; idx += stride) {
    if (idx < totalA) {
      const int64_t e = idx / N;
      T c = sA1 * cA1[e];
      if (cA2 != nullptr) c += sA2 * cA2[e];
      unsafeAtomicAdd(y + dmA[idx], c * detJA[idx]);
    } else {
      const int64_t j = idx - totalA;
      const int64_t e = j / N;
      const int32_t dof = dmB[j];
      unsafeAtomicAdd(y + dof, xB[dof] * (cB[e] * detJB[j]));
    }
  }
}

template <typename T>
inline hipError_t launch_facet_terms(T* y, const T* cA1, T sA1, const T* cA2, T sA2, const T* detJA, const int32_t* dmA,
                                     int64_t nentA, const T* xB, const T* cB, const T* detJB, const int32_t* dmB,
                                     int64_t nentB, int N, hipStream_t stream, const T* sdev = nullptr) {
  const int64_t total = (nentA + nentB) * (int64_t)N;
  if (total <= 0) return hipSuccess;
  int64_t nblocks = (total + 255) / 256;
  if (nblocks > 4096) nblocks = 4096;
  hipLaunchKernelGGL((facet_terms_kernel<T>), dim3((unsigned)nblocks), dim3(256), 0, stream, y, cA1, sA1, cA2, sA2, detJA,
                     dmA, nentA * (int64_t)N, xB, cB, detJB, dmB, nentB * (int64_t)N, N, sdev);
  return hipGetLastError();
}

// Planned mass apply (batch plan of csrc/plan.hpp built for the same entity dofmap, N dofs per
// entity, epb entities per batch).  Per batch: gather x once per distinct dof into LDS, every
// (entity, local dof) entry multiplies and pre-reduces into LDS, one global atomic per distinct
// dof with consecutive lanes on ascending addresses.  EPT = entries per thread (upper bound).
// EXCL: the plan carries exclusive-dof marks (plan.hpp): a marked dof is touched by this workgroup alone, so its sum is
// finished with a plain load (issued with the x gather) + store instead of a memory-side atomic.
template <typename T, int EPT, bool EXCL>
__global__ void __launch_bounds__(256)
    mass_plan_kernel(const T* __restrict__ x, const T* __restrict__ entity_constants, T* __restrict__ y,
                     const T* __restrict__ detJ, const int32_t* __restrict__ nu, const int32_t* __restrict__ udofs,
                     const uint16_t* __restrict__ slot, int N, int epb, int64_t nent, uint32_t inv_n,
                     const int32_t* __restrict__ order, const int32_t* __restrict__ runs, LaunchSignal sig,
                     const uint32_t* __restrict__ excl, int excl_words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  launch_signal_publish(sig);
  const int M = N * epb;
  PlanAcc* sy = reinterpret_cast<PlanAcc*>(smem_raw);  // partial sums in double also for fp32 (stiffness_plan.hpp)
  T* sx = reinterpret_cast<T*>(sy + M);

  const int tid = threadIdx.x;
  const int64_t batch = blockIdx.x;
  const int64_t ent0 = batch * epb;
  const int64_t left = nent - ent0;
  const int valid = (int)((left < epb ? left : epb) * N);
  const int64_t base = batch * (int64_t)M;
  const int32_t* ud = udofs + base;
  const bool use_runs = runs != nullptr;  // launch-uniform, tested at run time here (this kernel already has 18 shapes per type)
  const int32_t* rn = use_runs ? runs + batch * (int64_t)(2 * kPlanMaxRuns) : nullptr;

  int32_t mydof[EPT];
  const RunWords rt = use_runs ? batch_dofs_issue<true, EPT, 256>(ud, rn, M, tid, mydof)
                               : batch_dofs_issue<false, EPT, 256>(ud, rn, M, tid, mydof);
  uint16_t sl[EPT];
  T w[EPT];
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int i = tid + r * 256;
    const int ic = i < valid ? i : 0;
    sl[r] = slot[base + ic];
    const uint32_t e = __umulhi((uint32_t)ic, inv_n);  // ic / N
    if (order != nullptr) {  // entity at batch position e is order[ent0 + e]
      const int64_t ent = order[ent0 + e];
      w[r] = detJ[ent * N + (ic - (int)e * N)] * entity_constants[ent];
    } else {
      w[r] = detJ[base + ic] * entity_constants[ent0 + e];
    }
  }
  const int packed = nu[batch];
  const int nu_b = packed & 0xffff, nr_b = use_runs ? plan_runs_of<true>(packed) : 0;
  if (use_runs) batch_dofs_resolve<true, EPT, 256, true>(rt, ud, M, nu_b, nr_b, tid, reinterpret_cast<int32_t*>(sy), mydof);
  T xv[EPT];
#pragma unroll
  for (int r = 0; r < EPT; ++r) xv[r] = x[mydof[r]];
  [[maybe_unused]] T yv[EXCL ? EPT : 1];
  [[maybe_unused]] bool mine[EXCL ? EPT : 1];
  if constexpr (EXCL) {
    const uint32_t* ex = excl + batch * (int64_t)excl_words;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int s = tid + r * 256;
      mine[r] = s < nu_b && ((ex[s >> 5] >> (s & 31)) & 1u);
      yv[r] = mine[r] ? y[mydof[r]] : T(0);
    }
  }
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int s = tid + r * 256;
    if (s < nu_b) {
      sx[s] = xv[r];
      sy[s] = PlanAcc(0);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int i = tid + r * 256;
    if (i < valid) lds_atomic_add(&sy[sl[r]], (PlanAcc)(sx[sl[r]] * w[r]));
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int s = tid + r * 256;
    if (s < nu_b) {
      if constexpr (EXCL) {
        if (mine[r])
          y[mydof[r]] = yv[r] + (T)sy[s];
        else
          unsafeAtomicAdd(y + mydof[r], (T)sy[s]);
      } else {
        unsafeAtomicAdd(y + mydof[r], (T)sy[s]);
      }
    }
  }
}

template <typename T>
inline hipError_t launch_mass_plan(const T* x, const T* consts, T* y, const T* detJ, const void* workspace, int N,
                                   int epb, int64_t nent, hipStream_t stream, bool ordered = false, bool use_runs = false,
                                   bool exclusive = false) {
  if (nent <= 0) return hipSuccess;
  const int M = N * epb;
  if (M < 1 || M > kPlanMaxEntries) return hipErrorInvalidValue;
  PlanView v = plan_view_generic(const_cast<void*>(workspace), N, epb, nent);
  const uint32_t inv_n = (uint32_t)((0x100000000ull + (uint64_t)N - 1) / (uint64_t)N);  // ceil(2^32 / N), N >= 2
  const size_t lds = (size_t)M * (sizeof(PlanAcc) + sizeof(T));
  const dim3 grid((unsigned)v.nbatch), block(256);
  const LaunchSignal sig = take_launch_signal(stream);
#define FUS_MASS_LAUNCH(E)                                                                                               \
  if (exclusive)                                                                                                         \
    hipLaunchKernelGGL((mass_plan_kernel<T, E, true>), grid, block, lds, stream, x, consts, y, detJ, v.nu, v.udofs, v.slot, \
                       N, epb, nent, inv_n, ordered ? v.order : nullptr, use_runs ? v.runs : nullptr, sig, v.excl,      \
                       (int)v.excl_words);                                                                               \
  else                                                                                                                   \
    hipLaunchKernelGGL((mass_plan_kernel<T, E, false>), grid, block, lds, stream, x, consts, y, detJ, v.nu, v.udofs, v.slot, \
                       N, epb, nent, inv_n, ordered ? v.order : nullptr, use_runs ? v.runs : nullptr, sig, nullptr, 0)
  const int ept = (M + 255) / 256;
  if (ept <= 1) FUS_MASS_LAUNCH(1);
  else if (ept <= 2) FUS_MASS_LAUNCH(2);
  else if (ept <= 3) FUS_MASS_LAUNCH(3);
  else if (ept <= 4) FUS_MASS_LAUNCH(4);
  else if (ept <= 5) FUS_MASS_LAUNCH(5);
  else if (ept <= 6) FUS_MASS_LAUNCH(6);
  else if (ept <= 8) FUS_MASS_LAUNCH(8);
  else if (ept <= 11) FUS_MASS_LAUNCH(11);
  else FUS_MASS_LAUNCH(16);
#undef FUS_MASS_LAUNCH
  return settle_launch_signal(stream, sig, hipGetLastError());
}

}  // namespace fus
