// Planned stiffness apply for AFFINE cells (opt-in; never the headline path, whose
// bytes contract is the general per-quadrature-point G).  The geometric factor of an affine cell is
// one symmetric 3x3 matrix times the quadrature weight, G[c][q] = G[c][0] * (w_q / w_0), so only the
// first record of the cell (48 B instead of 48 n^3 B) is read; the per-plane factors are formed where
// they are used.  Own __global__ template (see stiffness_plan.hpp for why).
#pragma once

#include "stiffness_plan.hpp"

namespace fus {

template <typename T, int P, int CPB, bool ALIAS, bool PADLDS, int MINW, bool ORDERED, bool RUNS>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()), MINW)
    stiffness_plan_affine_kernel(const T* __restrict__ x, const T* __restrict__ cell_constants, T* __restrict__ y,
                                 const T* __restrict__ G, const int32_t* __restrict__ nu,
                                 const int32_t* __restrict__ udofs, const uint16_t* __restrict__ slot,
                                 const T* __restrict__ dphi, int64_t ncell, const T* __restrict__ wratio,
                                 const int32_t* __restrict__ order, const int32_t* __restrict__ runs, LaunchSignal sig) {
  using Sh = PlanShape<T, P, CPB, PADLDS>;
  constexpr int n = Sh::n, n2 = Sh::n2, Nd = Sh::Nd, S = Sh::S, BLOCK = Sh::BLOCK, M = Sh::M, SPT = Sh::SPT;
  launch_signal_publish(sig);

  __shared__ T sD[n2 + 1];  // + 1: plan_table_store
  __shared__ T su[CPB * S];
  __shared__ T sfy[CPB * S];
  __shared__ T sfz[CPB * S];
  __shared__ PlanAcc sacc[PlanOwnAcc<T, ALIAS>::value ? M : 1];
  T* const sx = ALIAS ? sfy : reinterpret_cast<T*>(sacc);  // x values of the batch's distinct dofs
  PlanAcc* const sy = PlanOwnAcc<T, ALIAS>::value ? sacc : reinterpret_cast<PlanAcc*>(su);  // their y partial sums

  const int tid = threadIdx.x;
  const unsigned batch = blockIdx.x;
  const int lc = tid / n2;
  const int t = tid - lc * n2;
  const int ty = t / n, tz = t - ty * n;
  const int64_t pos = (int64_t)batch * CPB + lc;  // position in the plan's cell order
  const bool active = (lc < CPB) && (pos < ncell);
  const int32_t* ud = udofs + (int64_t)batch * M;
  const int32_t* rn = runs + (int64_t)batch * (2 * kPlanMaxRuns);  // read only when RUNS

  // ---- issue every HBM load of the batch up front (the rules: plan.hpp, "the preamble every planned kernel shares")
  const int64_t pos_ld = plan_load_pos<CPB>((int64_t)batch * CPB, lc, ncell);
  const uint32_t row = plan_row_issue<ORDERED>(order, pos_ld);
  const T dval = dphi[tid < n2 ? tid : 0];
  int32_t mydof[SPT];
  const RunWords rt = batch_dofs_issue<RUNS, SPT, BLOCK>(ud, rn, M, tid, mydof);
  uint16_t sl[n];
  T g0[6];
  T wr[n];
  const int64_t cell = plan_row<ORDERED>(row, pos_ld);  // row of the per-cell arrays
  PlanSlotWord<n> sraw[n];  // narrowed once the gather is on its way (plan.hpp, PlanSlotWord)
  T coeff = T(0);
  if (plan_loads_by_all<n>() || active) {
    const uint16_t* sp = slot + pos_ld * Nd + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sraw[ix] = sp[ix * n2];
#pragma unroll
    for (int ix = 0; ix < n; ++ix) wr[ix] = wratio[ix * n2 + t];
    load_g6<T>(G + cell * Nd * 6, g0);
    coeff = cell_constants[cell];
  }
  const int packed = nu[batch];
  const int nu_b = packed & 0xffff, nr_b = plan_runs_of<RUNS>(packed);
  plan_table_store<n, n2>(sD, tid, dval);
  batch_dofs_resolve<RUNS, SPT, BLOCK>(rt, ud, M, nu_b, nr_b, tid, reinterpret_cast<int32_t*>(su), mydof);

  T u[n];
  plan_gather_x<T, n, n2, SPT, BLOCK>(x, mydof, nu_b, tid, active, sraw, sl, sx, su + lc * S + t, u);
  if constexpr (!ALIAS) plan_zero<T, SPT, BLOCK>(sy, nu_b, tid);

  T fx[n];
  if (active) {
    T dy[n], dz[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      dy[i] = sD[ty * n + i];
      dz[i] = sD[tz * n + i];
    }
    const T* cu_y = su + lc * S + tz;
    const T* cu_z = su + lc * S + ty * n;
    T* cfy = sfy + lc * S + t;
    T* cfz = sfz + lc * S + t;
#pragma unroll
    for (int qx = 0; qx < n; ++qx) {
      T vx, vy, vz;
      plan_grad_at<T, n, n2>(qx, dphi, u, dy, dz, cu_y, cu_z, vx, vy, vz);
      const T cw = coeff * wr[qx];
      fx[qx] = cw * (g0[0] * vx + g0[1] * vy + g0[2] * vz);
      cfy[qx * n2] = cw * (g0[1] * vx + g0[3] * vy + g0[4] * vz);
      cfz[qx * n2] = cw * (g0[2] * vx + g0[4] * vy + g0[5] * vz);
    }
  }
  __syncthreads();
  if constexpr (ALIAS) {
    plan_zero<T, SPT, BLOCK>(sy, nu_b, tid);
    __syncthreads();
  }

  plan_backward<T, n, n2>(dphi, sD, ty, tz, active, fx, sfy + lc * S + tz, sfz + lc * S + ty * n, sl, sy);
  plan_flush<T, SPT, BLOCK>(y, mydof, nu_b, tid, sy);
}

template <typename T, int P, bool ALIAS, bool PADLDS, int MINW>
inline hipError_t launch_stiffness_plan_affine(const T* x, const T* cc, T* y, const T* G, const T* wratio,
                                               const void* workspace, const T* dphi, int64_t ncell,
                                               hipStream_t stream, bool ordered = false, bool use_runs = false) {
  constexpr int CPB = plan_cells_per_batch<P>();
  if (ncell <= 0) return hipSuccess;
  PlanView v = plan_view(const_cast<void*>(workspace), P, CPB, ncell);
  constexpr int threads = col_block_threads<P, CPB>();
  const LaunchSignal sig = take_launch_signal(stream);
  plan_dispatch(ordered, use_runs, [&](auto o, auto r) {
    hipLaunchKernelGGL((stiffness_plan_affine_kernel<T, P, CPB, ALIAS, PADLDS, MINW, decltype(o)::value, decltype(r)::value>),
                       dim3((unsigned)v.nbatch), dim3(threads), 0, stream, x, cc, y, G, v.nu, v.udofs, v.slot, dphi, ncell, wratio,
                       v.order, v.runs, sig);
  });
  return settle_launch_signal(stream, sig, hipGetLastError());
}

}  // namespace fus
