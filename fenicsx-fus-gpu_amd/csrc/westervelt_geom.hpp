// Fused Westervelt cell pass with the geometry formed in the kernel (SURVEY 8f ranks 3 + 4 combined):
// as westervelt_cell_kernel (westervelt.hpp), but neither G (48 n^3 bytes per cell) nor detJ (8 n^3) is
// read -- both come from the 8 vertices of the (trilinear) cell, with the formulas of
// numba-cpu/precompute.py:76-163 specialised as in stiffness_geom.hpp.  BASELINE config 5 (P = 6, bowl
// mesh): 31.9 -> 12.7 kB per cell.  Own bytes contract, own bench line; never the headline.
#pragma once

#include "stiffness_geom.hpp"
#include "westervelt.hpp"

namespace fus {

template <typename T, int P, int CPB, int MINW, bool MASS, bool ORDERED, bool RUNS>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()), MINW)
    westervelt_cell_geom_kernel(const T* __restrict__ u_in, const T* __restrict__ v_in, const T* __restrict__ c2,
                                const T* __restrict__ c3, const T* __restrict__ c4, const T* __restrict__ c5,
                                T* __restrict__ b, T* __restrict__ m, const T* __restrict__ x_g,
                                const int32_t* __restrict__ x_dofs, const T* __restrict__ pts,
                                const T* __restrict__ wts, const int32_t* __restrict__ nu,
                                const int32_t* __restrict__ udofs, const uint16_t* __restrict__ slot,
                                const T* __restrict__ dphi, int64_t ncell, const int32_t* __restrict__ order,
                                const int32_t* __restrict__ runs, LaunchSignal sig) {
  constexpr int n = P + 1, n2 = n * n, Nd = n2 * n;
  launch_signal_publish(sig);
  constexpr int S = lds_cell_stride<T, P>();
  constexpr int BLOCK = col_block_threads<P, CPB>();
  constexpr int M = CPB * Nd;
  constexpr int SPT = (M + BLOCK - 1) / BLOCK;
  constexpr int VPT = (CPB * 24 + BLOCK - 1) / BLOCK;

  __shared__ T sD[n2 + 1];  // + 1: plan_table_store
  __shared__ T sP[n + 1], sW[n + 1];
  __shared__ T sX[CPB * 24];
  __shared__ T su[CPB * S];
  __shared__ T sfy[CPB * S];
  __shared__ T sfz[CPB * S];
  T* const sxu = sfy;
  T* const sxv = sfz;
  // partial sums are accumulated in double (PlanAcc, stiffness_plan.hpp): fp64 kernels alias them onto dead
  // cubes, fp32 kernels get arrays of their own
  constexpr bool OWN_ACC = sizeof(T) != sizeof(PlanAcc);
  __shared__ PlanAcc sacc_b[OWN_ACC ? M : 1];
  __shared__ PlanAcc sacc_m[(OWN_ACC && MASS) ? M : 1];
  PlanAcc* const sm = OWN_ACC ? sacc_m : reinterpret_cast<PlanAcc*>(sfy);
  PlanAcc* const sb = OWN_ACC ? sacc_b : reinterpret_cast<PlanAcc*>(su);

  const int tid = threadIdx.x;
  const unsigned batch = blockIdx.x;
  const int lc = tid / n2;
  const int t = tid - lc * n2;
  const int ty = t / n, tz = t - ty * n;
  const int64_t cell0 = (int64_t)batch * CPB;
  const int64_t pos = cell0 + lc;
  const bool active = (lc < CPB) && (pos < ncell);
  const int32_t* ud = udofs + (int64_t)batch * M;
  const int32_t* rn = runs + (int64_t)batch * (2 * kPlanMaxRuns);  // read only when RUNS

  // ---- round trip 1: everything that depends on the kernel arguments alone (the rules: plan.hpp, "the preamble every planned
  // kernel shares")
  const int64_t pos_ld = plan_load_pos<CPB>(cell0, lc, ncell);
  const uint32_t row = plan_row_issue<ORDERED>(order, pos_ld);
  const T dval = dphi[tid < n2 ? tid : 0];
  const T pval = pts[tid < n ? tid : 0];
  const T wval = wts[tid < n ? tid : 0];
  int32_t mydof[SPT];
  const RunWords rt = batch_dofs_issue<RUNS, SPT, BLOCK>(ud, rn, M, tid, mydof);
  int32_t vid[VPT];
  stage_vertex_ids<ORDERED, VPT, BLOCK, CPB>(x_dofs, order, cell0, ncell, tid, vid);
  uint16_t sl[n];
  if (plan_loads_by_all<n>() || active) {
    const uint16_t* sp = slot + pos_ld * Nd + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sl[ix] = sp[ix * n2];
  }
  // ---- round trip 2: what those point to -- (ORDERED: vertex ids and the cell's constants;) u, v and the vertex coordinates
  stage_vertex_ids_of_rows<ORDERED, VPT, BLOCK, CPB>(x_dofs, tid, vid);
  T k2 = T(0), k3 = T(0), k4 = T(0), k5 = T(0);
  if (plan_loads_by_all<n>() || active) {
    const int64_t cell = plan_row<ORDERED>(row, pos_ld);
    if constexpr (MASS) {
      k2 = c2[cell];
      k5 = c5[cell];
    }
    k3 = c3[cell];
    k4 = c4[cell];
  }
  const int packed = nu[batch];
  const int nu_b = packed & 0xffff, nr_b = plan_runs_of<RUNS>(packed);
  plan_table_store<n, n2>(sD, tid, dval);
  plan_table_store<n, n>(sP, tid, pval);
  plan_table_store<n, n>(sW, tid, wval);
  batch_dofs_resolve<RUNS, SPT, BLOCK>(rt, ud, M, nu_b, nr_b, tid, reinterpret_cast<int32_t*>(su), mydof);
  {
    T xu[SPT], xv[SPT];
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      xu[r] = u_in[mydof[r]];
      xv[r] = v_in[mydof[r]];
    }
    T cv[VPT];
    stage_vertex_coords_issue<T, VPT, BLOCK, CPB>(x_g, vid, tid, cv);
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      if (s < nu_b) {
        sxu[s] = xu[r];
        sxv[s] = xv[r];
      }
    }
    stage_vertex_coords_store<T, VPT, BLOCK, CPB>(cv, tid, sX);
  }
  __syncthreads();  // B1: u / v values and vertex coordinates are in LDS

  T J0[3], Ja[3], Jba[3], Jc[3], Jdc[3];
  T wyz = T(0);
  T w[n];       // combined stiffness input  c3 u + c4 v
  T bextra[n];  // detJ c5 v^2
  T madd[n];    // detJ c2 u
  if (active) {
    column_jacobian_rows<T>(sX + lc * 24, sP[ty], sP[tz], J0, Ja, Jba, Jc, Jdc);
    wyz = sW[ty] * sW[tz];
    T* cu = su + lc * S + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) {
      const T uu = sxu[sl[ix]], vv = sxv[sl[ix]];
      w[ix] = k3 * uu + k4 * vv;
      if constexpr (MASS) {
        const T dj = column_absdet_at<T>(pts[ix], J0, Ja, Jba, Jc, Jdc) * (wts[ix] * wyz);  // scaled Jacobian determinant
        bextra[ix] = dj * k5 * vv * vv;
        madd[ix] = dj * k2 * uu;
      }
      cu[ix * n2] = w[ix];
    }
  }
  __syncthreads();  // B2
  if constexpr (MASS) {
    plan_zero<T, SPT, BLOCK>(sm, nu_b, tid);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int ix = 0; ix < n; ++ix) lds_atomic_add(&sm[sl[ix]], (PlanAcc)madd[ix]);
    }
    __syncthreads();
    plan_flush<T, SPT, BLOCK>(m, mydof, nu_b, tid, sm);
    __syncthreads();
  }

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
      plan_grad_at<T, n, n2>(qx, dphi, w, dy, dz, cu_y, cu_z, vx, vy, vz);
      if constexpr (sizeof(T) == 8 && P == 6) {  // the flux without forming G (stiffness_geom.hpp column_flux_at: 12 fp64 operations less per
        T fy, fz;                                 // quadrature point); where measured: config 5's step -1.5 % (profiles/r06k_ab_flux_form_westervelt.log)
        column_flux_at<T>(pts[qx], wts[qx] * wyz, J0, Ja, Jba, Jc, Jdc, vx, vy, vz, fx[qx], fy, fz);
        cfy[qx * n2] = fy;
        cfz[qx * n2] = fz;
      } else {
        T gq[6];
        column_g_at<T>(pts[qx], wts[qx] * wyz, J0, Ja, Jba, Jc, Jdc, gq);
        fx[qx] = gq[0] * vx + gq[1] * vy + gq[2] * vz;
        cfy[qx * n2] = gq[1] * vx + gq[3] * vy + gq[4] * vz;
        cfz[qx * n2] = gq[2] * vx + gq[4] * vy + gq[5] * vz;
      }
    }
  }
  __syncthreads();  // B3
  plan_zero<T, SPT, BLOCK>(sb, nu_b, tid);
  __syncthreads();

  if (active) {
    T dyT[n], dzT[n];
#pragma unroll
    for (int q = 0; q < n; ++q) {
      dyT[q] = sD[q * n + ty];
      dzT[q] = sD[q * n + tz];
    }
    const T* cf_y = sfy + lc * S + tz;
    const T* cf_z = sfz + lc * S + ty * n;
#pragma unroll
    for (int jx = 0; jx < n; ++jx) {
      T acc = MASS ? bextra[jx] : T(0);
#pragma unroll
      for (int qx = 0; qx < n; ++qx) acc += dphi[qx * n + jx] * fx[qx];
#pragma unroll
      for (int q = 0; q < n; ++q) {
        acc += dyT[q] * cf_y[jx * n2 + q * n];
        acc += dzT[q] * cf_z[jx * n2 + q];
      }
      lds_atomic_add(&sb[sl[jx]], (PlanAcc)acc);
    }
  }
  __syncthreads();  // B4
  plan_flush<T, SPT, BLOCK>(b, mydof, nu_b, tid, sb);
}

template <typename T, int P, bool MASS = true>
inline hipError_t launch_westervelt_cell_geom(const T* u, const T* v, const T* c2, const T* c3, const T* c4,
                                              const T* c5, T* b, T* m, const T* x_g, const int32_t* x_dofs,
                                              const T* pts, const T* wts, const void* workspace, const T* dphi,
                                              int64_t ncell, hipStream_t stream, bool ordered = false, bool use_runs = false) {
  constexpr int CPB = plan_cells_per_batch<P>();
  if (ncell <= 0) return hipSuccess;
  PlanView pv = plan_view(const_cast<void*>(workspace), P, CPB, ncell);
  constexpr int threads = col_block_threads<P, CPB>();
  const LaunchSignal sig = take_launch_signal(stream);
  plan_dispatch(ordered, use_runs, [&](auto o, auto r) {
    hipLaunchKernelGGL((westervelt_cell_geom_kernel<T, P, CPB, 1, MASS, decltype(o)::value, decltype(r)::value>),
                       dim3((unsigned)pv.nbatch), dim3(threads), 0, stream, u, v, c2, c3, c4, c5, b, m, x_g, x_dofs, pts, wts, pv.nu,
                       pv.udofs, pv.slot, dphi, ncell, pv.order, pv.runs, sig);
  });
  return settle_launch_signal(stream, sig, hipGetLastError());
}

}  // namespace fus
