// Planned stiffness apply with the geometric factor formed IN THE KERNEL from the cell's 8 vertices
// (second half): no G stream at all.  Reported as its own line with its own bytes
// contract (bench.py --mode stiffness_geom), never mixed into the headline, whose contract is the
// general G array.
//
// Same formulas and conventions as the reference's host precompute
//   numba-cpu/precompute.py:115-163  compute_scaled_geometrical_factor
// (J_[a][d] = sum_v dphi[a][q][v] X[v][d], G = w |det J_| inv(J_)^T inv(J_) upper triangle, indexed
// by reference directions), specialised to what its callers pass: P1 (trilinear, 8-vertex)
// hexahedra, vertex v = vx + 2 vy + 4 vz, and the tensor GLL rule q = qx n^2 + qy n + qz
// (numba-cpu/test_operators.py:98-107).  For a trilinear map the three rows of J_ along a column
// (qy, qz fixed -- exactly what one thread owns) are
//   J_[0] = bilinear in (xi_y, xi_z) of the x-edge vectors            -- constant along the column
//   J_[1] = (1 - xi_x) A + xi_x B,   J_[2] = (1 - xi_x) C + xi_x D    -- linear in xi_x
// so a thread keeps 15 values instead of the 6 n of the G slab and forms adj(J_), det and the six
// entries of G per quadrature point in registers (~60 flops on top of the ~80 of the contractions;
// the kernel stays far below the fp64 vector peak).  The 24 vertex coordinates of each cell are
// gathered once per cell into LDS (x_dofs -> x_g, one value per thread) alongside the x gather.
#pragma once

#include "stiffness_plan.hpp"

namespace fus {

// Occupancy hint of the shipped builds (waves per SIMD the register allocation must allow).
template <typename T, int P>
__host__ __device__ constexpr int geom_min_waves() {
  return 1;
}

// Geometric factor of the column at quadrature plane qx, scaled by s0 = cell constant * w_y * w_z:
// rows of J_ (J0 constant, J1 / J2 linear in xi_x), adj(J_), det, G = w |det| adj^T adj / det^2.
template <typename T>
__device__ __forceinline__ void column_g_at(T ex, T wx_s0, const T (&J0)[3], const T (&Ja)[3], const T (&Jba)[3],
                                            const T (&Jc)[3], const T (&Jdc)[3], T (&gq)[6]) {
  T J1[3], J2[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    J1[d] = Ja[d] + ex * Jba[d];
    J2[d] = Jc[d] + ex * Jdc[d];
  }
  // adj(J_): A[d][a], inv(J_) = A / det  (same expressions as geometry_kernel, geometry.hpp)
  T A[3][3];
  A[0][0] = J1[1] * J2[2] - J1[2] * J2[1];
  A[0][1] = J0[2] * J2[1] - J0[1] * J2[2];
  A[0][2] = J0[1] * J1[2] - J0[2] * J1[1];
  A[1][0] = J1[2] * J2[0] - J1[0] * J2[2];
  A[1][1] = J0[0] * J2[2] - J0[2] * J2[0];
  A[1][2] = J0[2] * J1[0] - J0[0] * J1[2];
  A[2][0] = J1[0] * J2[1] - J1[1] * J2[0];
  A[2][1] = J0[1] * J2[0] - J0[0] * J2[1];
  A[2][2] = J0[0] * J1[1] - J0[1] * J1[0];
  const T det = J0[0] * A[0][0] + J0[1] * A[1][0] + J0[2] * A[2][0];
  // c w_q |det| / det^2 = c w_q / |det|.  fp64: the reciprocal as v_rcp_f64 + two Newton steps instead of the IEEE division sequence (8
  // instructions less per quadrature point; same result to the last digits: profiles/r05y_ab_rcp_division.log, -1.5 % at P = 4)
  const T ad = det < T(0) ? -det : det;
  T s;
  if constexpr (sizeof(T) == 8) {
    double r = __builtin_amdgcn_rcp((double)ad);
    r = r * (2.0 - (double)ad * r);
    r = r * (2.0 - (double)ad * r);
    s = wx_s0 * (T)r;
  } else {
    s = wx_s0 / ad;
  }
  gq[0] = s * (A[0][0] * A[0][0] + A[1][0] * A[1][0] + A[2][0] * A[2][0]);
  gq[1] = s * (A[0][0] * A[0][1] + A[1][0] * A[1][1] + A[2][0] * A[2][1]);
  gq[2] = s * (A[0][0] * A[0][2] + A[1][0] * A[1][2] + A[2][0] * A[2][2]);
  gq[3] = s * (A[0][1] * A[0][1] + A[1][1] * A[1][1] + A[2][1] * A[2][1]);
  gq[4] = s * (A[0][1] * A[0][2] + A[1][1] * A[1][2] + A[2][1] * A[2][2]);
  gq[5] = s * (A[0][2] * A[0][2] + A[1][2] * A[1][2] + A[2][2] * A[2][2]);
}

// The flux G (vx, vy, vz) at quadrature plane qx WITHOUT forming G (round 6): with the columns of adj(J_)
//   a = J1 x J2,   b = J2 x J0,   c = J0 x J1        (A[d][0], A[d][1], A[d][2] of column_g_at)
// G = s [col_alpha . col_beta], so  G v = s (a . w, b . w, c . w)  with  w = a vx + b vy + c vz:  9 + 12 operations instead of the 24 of
// the six scaled dot products + 9 of the symmetric product -- 55 instead of 67 fp64 operations per quadrature point; these kernels are bound
// by instruction issue.  Same conventions and the same reciprocal as column_g_at; the result differs from G v in rounding only.
// Used WHERE MEASURED FASTER (geom_flux_operator_form; profiles/r06j_*, r06k_*, r06l_*: interleaved with the build before it): the in-loop
// builds at fp64 P = 6, 8, 9 (-2.2, -2.7, -0.5 %) and the Westervelt cell pass at fp64 P = 6 (step -1.5 %); NOT at P = 7 (+8 % in the steady
// state: the first 100-launch burst equal, every later one slower), P = 10 (+1.4 %), fp32 from P = 5 (+4 ... +5 %).  Below degree 6 the kernel
// holds the n x 6 factors in registers (PREG) unless geom_factors_in_registers says otherwise.
template <typename T, int P>
__host__ __device__ constexpr bool geom_flux_operator_form() {
  return sizeof(T) == 8 ? (P == 2 || P == 3 || P == 6 || P == 8 || P == 9) : (P == 2 || P == 3 || P == 4);
}
// PREG of stiffness_plan_geom_kernel per degree and scalar type: the n x 6 factors of the column held in registers (formed between the gather's
// barriers) up to degree 5 -- except where the in-loop flux form is measured faster at the SAME occupancy without scratch (fp64 P = 2, 3: -2.7, -3.4 %;
// fp32 P = 2, 3, 4: -4.4, -2.9, -2.6 %; profiles/r06j_*, r06l_*).  fp64 P = 4, 5 would gain 1.8 / 2.8 % only with 12 / 36 bytes of scratch: they stay.
template <typename T, int P>
__host__ __device__ constexpr bool geom_factors_in_registers() {
  return P <= 5 && !(P >= 2 && geom_flux_operator_form<T, P>());
}
template <typename T>
__device__ __forceinline__ void column_flux_at(T ex, T wx_s0, const T (&J0)[3], const T (&Ja)[3], const T (&Jba)[3], const T (&Jc)[3],
                                               const T (&Jdc)[3], T vx, T vy, T vz, T& fx, T& fy, T& fz) {
  T J1[3], J2[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    J1[d] = Ja[d] + ex * Jba[d];
    J2[d] = Jc[d] + ex * Jdc[d];
  }
  T a[3], b[3], c[3];
  a[0] = J1[1] * J2[2] - J1[2] * J2[1];
  a[1] = J1[2] * J2[0] - J1[0] * J2[2];
  a[2] = J1[0] * J2[1] - J1[1] * J2[0];
  b[0] = J0[2] * J2[1] - J0[1] * J2[2];
  b[1] = J0[0] * J2[2] - J0[2] * J2[0];
  b[2] = J0[1] * J2[0] - J0[0] * J2[1];
  c[0] = J0[1] * J1[2] - J0[2] * J1[1];
  c[1] = J0[2] * J1[0] - J0[0] * J1[2];
  c[2] = J0[0] * J1[1] - J0[1] * J1[0];
  const T det = J0[0] * a[0] + J0[1] * a[1] + J0[2] * a[2];
  const T ad = det < T(0) ? -det : det;
  T s;
  if constexpr (sizeof(T) == 8) {
    double r = __builtin_amdgcn_rcp((double)ad);
    r = r * (2.0 - (double)ad * r);
    r = r * (2.0 - (double)ad * r);
    s = wx_s0 * (T)r;
  } else {
    s = wx_s0 / ad;
  }
  T w[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) w[d] = s * (a[d] * vx + b[d] * vy + c[d] * vz);
  fx = a[0] * w[0] + a[1] * w[1] + a[2] * w[2];
  fy = b[0] * w[0] + b[1] * w[1] + b[2] * w[2];
  fz = c[0] * w[0] + c[1] * w[1] + c[2] * w[2];
}

// Rows of J_ along the column (xi_y, xi_z) = (ey, ez) of a trilinear cell with vertex coordinates
// X[(vx + 2 vy + 4 vz) * 3 + d]:  J_[0] = J0,  J_[1] = Ja + xi_x Jba,  J_[2] = Jc + xi_x Jdc.
template <typename T>
__device__ __forceinline__ void column_jacobian_rows(const T* __restrict__ X, T ey, T ez, T (&J0)[3], T (&Ja)[3],
                                                     T (&Jba)[3], T (&Jc)[3], T (&Jdc)[3]) {
  const T fy0 = T(1) - ey, fz0 = T(1) - ez;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const T x000 = X[0 * 3 + d], x100 = X[1 * 3 + d], x010 = X[2 * 3 + d], x110 = X[3 * 3 + d];
    const T x001 = X[4 * 3 + d], x101 = X[5 * 3 + d], x011 = X[6 * 3 + d], x111 = X[7 * 3 + d];
    // d/dxi_x: bilinear in (xi_y, xi_z) of the four x-edges
    J0[d] = fz0 * (fy0 * (x100 - x000) + ey * (x110 - x010)) + ez * (fy0 * (x101 - x001) + ey * (x111 - x011));
    // d/dxi_y at xi_x = 0 (A) and 1 (B): linear in xi_z of the y-edges
    const T A = fz0 * (x010 - x000) + ez * (x011 - x001);
    const T B = fz0 * (x110 - x100) + ez * (x111 - x101);
    // d/dxi_z at xi_x = 0 (C) and 1 (D): linear in xi_y of the z-edges
    const T C = fy0 * (x001 - x000) + ey * (x011 - x010);
    const T D = fy0 * (x101 - x100) + ey * (x111 - x110);
    Ja[d] = A;
    Jba[d] = B - A;
    Jc[d] = C;
    Jdc[d] = D - C;
  }
}

// |det J_| at plane xi_x = ex of the column (the scaled Jacobian determinant is w_q times this:
// numba-cpu/precompute.py:76-112).
template <typename T>
__device__ __forceinline__ T column_absdet_at(T ex, const T (&J0)[3], const T (&Ja)[3], const T (&Jba)[3],
                                              const T (&Jc)[3], const T (&Jdc)[3]) {
  T J1[3], J2[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    J1[d] = Ja[d] + ex * Jba[d];
    J2[d] = Jc[d] + ex * Jdc[d];
  }
  const T det = J0[0] * (J1[1] * J2[2] - J1[2] * J2[1]) + J0[1] * (J1[2] * J2[0] - J1[0] * J2[2]) +
                J0[2] * (J1[0] * J2[1] - J1[1] * J2[0]);
  return det < T(0) ? -det : det;
}

// Stage the 24 vertex coordinates of each cell of the batch in LDS (sX[cell in batch][vertex][axis]), entry e = tid + r * BLOCK
// = (cell in batch) * 24 + vertex * 3 + axis.  Three steps, so that the loads travel with the other loads of the preamble
// (plan.hpp): (A) with the plan's lists: the vertex id itself, or, for an ORDERED plan, the row of the cell; (B) ORDERED only:
// row -> vertex id; (C) with the x gather: the coordinate (stage_vertex_coords_issue), stored after the gather was issued.
template <bool ORDERED, int VPT, int BLOCK, int CPB>
__device__ __forceinline__ void stage_vertex_ids(const int32_t* __restrict__ x_dofs, const int32_t* __restrict__ order,
                                                 int64_t cell0, int64_t ncell, int tid, int32_t (&vid)[VPT]) {
#pragma unroll
  for (int r = 0; r < VPT; ++r) {
    const int e = tid + r * BLOCK;
    const int c = e / 24, v = (e - c * 24) / 3;
    const bool ok = (e < CPB * 24) && (cell0 + c < ncell);
    const int64_t pc = ok ? cell0 + c : 0;
    vid[r] = ORDERED ? order[pc] : x_dofs[pc * 8 + (ok ? v : 0)];
  }
}
template <bool ORDERED, int VPT, int BLOCK, int CPB>
__device__ __forceinline__ void stage_vertex_ids_of_rows(const int32_t* __restrict__ x_dofs, int tid, int32_t (&vid)[VPT]) {
  if constexpr (ORDERED) {
#pragma unroll
    for (int r = 0; r < VPT; ++r) {
      const int e = tid + r * BLOCK;
      const int c = e / 24, v = (e - c * 24) / 3;
      vid[r] = x_dofs[(int64_t)(uint32_t)vid[r] * 8 + (e < CPB * 24 ? v : 0)];
    }
  }
}
template <typename T, int VPT, int BLOCK, int CPB>
__device__ __forceinline__ void stage_vertex_coords_issue(const T* __restrict__ x_g, const int32_t (&vid)[VPT], int tid, T (&cv)[VPT]) {
#pragma unroll
  for (int r = 0; r < VPT; ++r) {
    const int e = tid + r * BLOCK;
    cv[r] = x_g[(int64_t)vid[r] * 3 + (e < CPB * 24 ? e % 3 : 0)];
  }
}
template <typename T, int VPT, int BLOCK, int CPB>
__device__ __forceinline__ void stage_vertex_coords_store(const T (&cv)[VPT], int tid, T* __restrict__ sX) {
#pragma unroll
  for (int r = 0; r < VPT; ++r) {
    const int e = tid + r * BLOCK;
    if (e < CPB * 24) sX[e] = cv[r];
  }
}

// PREG: form the n x 6 factors of the column BEFORE the contraction phases (between the two barriers
// of the gather, while the u values are not yet in registers): the main loop then has the register
// profile of the general kernel (P = 4 fp64: 4 workgroups per CU) and the geometry arithmetic runs in
// the shadow of the gather.  Without it the factors are formed plane by plane inside the loop (fewest
// registers: the build for P >= 6).
template <typename T, int P, int CPB, bool ALIAS, bool PADLDS, int MINW, bool PREG, bool ORDERED, bool RUNS>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()), MINW)
    stiffness_plan_geom_kernel(const T* __restrict__ x, const T* __restrict__ cell_constants, T* __restrict__ y,
                               const T* __restrict__ x_g, const int32_t* __restrict__ x_dofs,
                               const T* __restrict__ pts, const T* __restrict__ wts,
                               const int32_t* __restrict__ nu, const int32_t* __restrict__ udofs,
                               const uint16_t* __restrict__ slot, const T* __restrict__ dphi, int64_t ncell,
                               const int32_t* __restrict__ order, const int32_t* __restrict__ runs, LaunchSignal sig) {
  using Sh = PlanShape<T, P, CPB, PADLDS>;
  constexpr int n = Sh::n, n2 = Sh::n2, Nd = Sh::Nd, S = Sh::S, BLOCK = Sh::BLOCK, M = Sh::M, SPT = Sh::SPT;
  launch_signal_publish(sig);
  constexpr int VPT = (CPB * 24 + BLOCK - 1) / BLOCK;  // vertex coordinates staged per thread (1 for P >= 4)

  __shared__ T sD[n2 + 1];  // + 1: plan_table_store
  __shared__ T sP[n + 1], sW[n + 1];
  __shared__ T sX[CPB * 24];
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
  const int64_t cell0 = (int64_t)batch * CPB;
  const int64_t pos = cell0 + lc;  // position in the plan's cell order
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
  // ---- round trip 2: what those point to -- (ORDERED: vertex ids and the cell's constant;) x and the vertex coordinates
  stage_vertex_ids_of_rows<ORDERED, VPT, BLOCK, CPB>(x_dofs, tid, vid);
  T coeff = T(0);
  if (plan_loads_by_all<n>() || active) coeff = cell_constants[plan_row<ORDERED>(row, pos_ld)];
  const int packed = nu[batch];
  const int nu_b = packed & 0xffff, nr_b = plan_runs_of<RUNS>(packed);
  plan_table_store<n, n2>(sD, tid, dval);
  plan_table_store<n, n>(sP, tid, pval);
  plan_table_store<n, n>(sW, tid, wval);
  batch_dofs_resolve<RUNS, SPT, BLOCK>(rt, ud, M, nu_b, nr_b, tid, reinterpret_cast<int32_t*>(su), mydof);

  // ---- gather x (as plan_gather_x) with the column geometry formed between its two barriers
  {
    T xv[SPT];
#pragma unroll
    for (int r = 0; r < SPT; ++r) xv[r] = x[mydof[r]];
    T cv[VPT];
    stage_vertex_coords_issue<T, VPT, BLOCK, CPB>(x_g, vid, tid, cv);
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      if (s < nu_b) sx[s] = xv[r];
    }
    stage_vertex_coords_store<T, VPT, BLOCK, CPB>(cv, tid, sX);
  }
  __syncthreads();  // x values and vertex coordinates are in LDS

  T J0[3], Ja[3], Jba[3], Jc[3], Jdc[3];
  T s0 = T(0);
  T g[PREG ? n : 1][6];
  if (active) {
    column_jacobian_rows<T>(sX + lc * 24, sP[ty], sP[tz], J0, Ja, Jba, Jc, Jdc);
    s0 = coeff * sW[ty] * sW[tz];
    if constexpr (PREG) {
#pragma unroll
      for (int qx = 0; qx < n; ++qx) column_g_at<T>(pts[qx], wts[qx] * s0, J0, Ja, Jba, Jc, Jdc, g[qx]);
    }
  }

  T u[n];
  if (active) {
    T* cu = su + lc * S + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) {
      u[ix] = sx[sl[ix]];
      cu[ix * n2] = u[ix];
    }
  }
  __syncthreads();
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
      if constexpr (!PREG && geom_flux_operator_form<T, P>()) {  // the flux without forming G; pts / wts with compile-time indices: scalar loads
        T fy, fz;
        column_flux_at<T>(pts[qx], wts[qx] * s0, J0, Ja, Jba, Jc, Jdc, vx, vy, vz, fx[qx], fy, fz);
        cfy[qx * n2] = fy;
        cfz[qx * n2] = fz;
      } else {
        T gl[6];
        if constexpr (!PREG) column_g_at<T>(pts[qx], wts[qx] * s0, J0, Ja, Jba, Jc, Jdc, gl);
        const T* gq = PREG ? g[PREG ? qx : 0] : gl;
        fx[qx] = gq[0] * vx + gq[1] * vy + gq[2] * vz;
        cfy[qx * n2] = gq[1] * vx + gq[3] * vy + gq[4] * vz;
        cfz[qx * n2] = gq[2] * vx + gq[4] * vy + gq[5] * vz;
      }
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

// CPB: cells per batch = the plan's entities per batch (default: the plan builder's own choice, ~256 threads per workgroup).
template <typename T, int P, bool ALIAS, bool PADLDS, int MINW, bool PREG, int CPB = plan_cells_per_batch<P>()>
inline hipError_t launch_stiffness_plan_geom(const T* x, const T* cc, T* y, const T* x_g, const int32_t* x_dofs,
                                             const T* pts, const T* wts, const void* workspace, const T* dphi,
                                             int64_t ncell, hipStream_t stream, bool ordered = false, bool use_runs = false) {
  if (ncell <= 0) return hipSuccess;
  PlanView v = plan_view(const_cast<void*>(workspace), P, CPB, ncell);
  constexpr int threads = col_block_threads<P, CPB>();
  const LaunchSignal sig = take_launch_signal(stream);
  plan_dispatch(ordered, use_runs, [&](auto o, auto r) {
    hipLaunchKernelGGL((stiffness_plan_geom_kernel<T, P, CPB, ALIAS, PADLDS, MINW, PREG, decltype(o)::value, decltype(r)::value>),
                       dim3((unsigned)v.nbatch), dim3(threads), 0, stream, x, cc, y, x_g, x_dofs, pts, wts, v.nu, v.udofs, v.slot,
                       dphi, ncell, v.order, v.runs, sig);
  });
  return settle_launch_signal(stream, sig, hipGetLastError());
}

}  // namespace fus
