// Planned stiffness apply (the shipped default): same column-per-thread contraction structure as
// stiffness_col_kernel (stiffness.hpp), gather / scatter through the batch plan (plan.hpp).
//
// This header holds ONLY product kernels.  Three kernels share the gather / contraction / flush
// phases below as force-inlined device functions and differ in where the geometric factor comes from:
//   stiffness_plan_kernel         general per-quadrature-point G[ncell][n^3][6] (the headline path)
//   stiffness_plan_affine_kernel  affine cells: one 6-value record per cell (stiffness_affine.hpp)
//   stiffness_plan_geom_kernel    G formed in registers from the cell's 8 vertices (stiffness_geom.hpp)
// Each is its own __global__ template so that an edit to one cannot change the register allocation
// of another; tests/test_resource_usage.py pins VGPR / occupancy / scratch of the default builds.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"
#include "stiffness.hpp"

namespace fus {

template <typename T, int P, int CPB, bool PADLDS>
struct PlanShape {
  static constexpr int n = P + 1, n2 = n * n, Nd = n2 * n;
  static constexpr int S = PADLDS ? lds_cell_stride<T, P>() : Nd;  // LDS stride between cell cubes
  static constexpr int BLOCK = col_block_threads<P, CPB>();
  static constexpr int M = CPB * Nd;                       // plan entries per batch
  static constexpr int SPT = (M + BLOCK - 1) / BLOCK;      // distinct-dof slots per thread (upper bound)
};

// Type of the per-batch partial sums in LDS.  ds_add_f32 is several times slower than ds_add_f64 on
// gfx950 (the fp32 planned kernel spent 28 % of its wave cycles stalled on LDS issue and ran 20 % faster
// with the LDS atomics removed, the fp64 kernel not at all: profiles/r02x_lds_atomic_f32.log), so fp32
// kernels also accumulate in double (which costs them nothing else: the sums are rounded to float once,
// at the flush).
typedef double PlanAcc;

// Slabs of G a thread keeps resident in the ring build of stiffness_plan_kernel (GPRE below): the
// largest ring that still reaches the next occupancy step of the register file (<= 128 VGPRs: 4
// waves per SIMD, <= 168: 3; tools/resource_usage.py), e.g. P = 6: 4 of 7 slabs, 156 VGPRs.
template <int P>
__host__ __device__ constexpr int plan_g_ring() {
  constexpr int ring[11] = {1, 2, 3, 4, 3, 3, 4, 3, 2, 1, 6};
  return ring[P];
}
// occupancy the ring build asks of the register allocator (P = 5: 130 VGPRs unforced, 2 over the step)
template <int P>
__host__ __device__ constexpr int plan_ring_min_waves() {
  return 1;
}

// Phase A (after every HBM load of the batch has been issued): gather x once per distinct dof,
// stage the values in LDS (sx), then every column reads its n values into registers (u) and writes
// them to the cell's u cube.  Two barriers; on return su is readable by the whole cell.
template <typename T, int n, int n2, int SPT, int BLOCK>
__device__ __forceinline__ void plan_gather_x(const T* __restrict__ x, const int32_t (&mydof)[SPT], int nu_b, int tid,
                                              bool active, const PlanSlotWord<n> (&sraw)[n], uint16_t (&sl)[n], T* __restrict__ sx,
                                              T* __restrict__ cu, T (&u)[n], T scale = T(1)) {
  T xv[SPT];
#pragma unroll
  for (int r = 0; r < SPT; ++r) xv[r] = x[mydof[r]];
  // the slots were loaded as 32-bit words; narrowing them is their first use, and a wait covers every OLDER load as well (the whole G
  // slab of the general kernel): it must come AFTER the gather is on its way (unpinned, the scheduler hoists it: it frees two registers)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ix = 0; ix < n; ++ix) sl[ix] = (uint16_t)sraw[ix];
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    if (s < nu_b) sx[s] = xv[r];
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int ix = 0; ix < n; ++ix) {
      u[ix] = scale * sx[sl[ix]];
      cu[ix * n2] = u[ix];
    }
  }
  __syncthreads();
}

// Gradient of the cell's field at quadrature point (qx, ty, tz): tx direction in registers against
// the derivative table in SGPRs (compile-time indices), ty / tz directions from the u cube in LDS.
template <typename T, int n, int n2>
__device__ __forceinline__ void plan_grad_at(int qx, const T* __restrict__ dphi, const T (&u)[n], const T (&dy)[n],
                                             const T (&dz)[n], const T* __restrict__ cu_y,
                                             const T* __restrict__ cu_z, T& vx, T& vy, T& vz) {
  vx = T(0);
#pragma unroll
  for (int ix = 0; ix < n; ++ix) vx += dphi[qx * n + ix] * u[ix];
  vy = T(0);
  vz = T(0);
#pragma unroll
  for (int i = 0; i < n; ++i) {
    vy += dy[i] * cu_y[qx * n2 + i * n];
    vz += dz[i] * cu_z[qx * n2 + i];
  }
}

// Phase C: transposed contractions of the fluxes (fx in registers, f_y / f_z cubes in LDS) and
// LDS pre-reduction of the batch's contributions into sy; ends with a barrier.
template <typename T, int n, int n2>
__device__ __forceinline__ void plan_backward(const T* __restrict__ dphi, const T* __restrict__ sD, int ty, int tz,
                                              bool active, const T (&fx)[n], const T* __restrict__ cf_y,
                                              const T* __restrict__ cf_z, const uint16_t (&sl)[n],
                                              PlanAcc* __restrict__ sy) {
  if (active) {
    T dyT[n], dzT[n];
#pragma unroll
    for (int q = 0; q < n; ++q) {
      dyT[q] = sD[q * n + ty];
      dzT[q] = sD[q * n + tz];
    }
#pragma unroll
    for (int jx = 0; jx < n; ++jx) {
      T acc = T(0);
#pragma unroll
      for (int qx = 0; qx < n; ++qx) acc += dphi[qx * n + jx] * fx[qx];
#pragma unroll
      for (int q = 0; q < n; ++q) {
        acc += dyT[q] * cf_y[jx * n2 + q * n];
        acc += dzT[q] * cf_z[jx * n2 + q];
      }
      lds_atomic_add(&sy[sl[jx]], (PlanAcc)acc);
    }
  }
  __syncthreads();
}

// Phase D: one global atomic per distinct dof; consecutive lanes -> ascending, mostly contiguous addresses.
template <typename T, int SPT, int BLOCK>
__device__ __forceinline__ void plan_flush(T* __restrict__ y, const int32_t (&mydof)[SPT], int nu_b, int tid,
                                           const PlanAcc* __restrict__ sy) {
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    if (s < nu_b) unsafeAtomicAdd(y + mydof[r], (T)sy[s]);
  }
}

template <typename T, int SPT, int BLOCK>
__device__ __forceinline__ void plan_zero(PlanAcc* __restrict__ sy, int nu_b, int tid) {
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    if (s < nu_b) sy[s] = PlanAcc(0);
  }
}

// LDS regions of the x values (sx) and of the partial sums (sy).  fp64: ALIAS puts sx in the f_y cube and sy
// in the u cube (no array of their own), otherwise both share one array of M values.  fp32: sy is double
// and gets an array of its own (M doubles; sx shares it when not ALIAS).
template <typename T, bool ALIAS>
struct PlanOwnAcc {
  static constexpr bool value = !ALIAS || sizeof(T) != sizeof(PlanAcc);
};

// General geometry.  GPRE = number of qx slabs of G a thread holds in registers: GPRE == n issues the
// whole 48 n^3-byte slab of the cell up front (most bytes in flight; best while registers allow >= 4
// waves per SIMD: P <= 5); GPRE < n keeps a ring of GPRE slabs and refills the slot a quadrature
// plane has just consumed (fewer live registers: one more workgroup per CU from P = 6 up).
//
// LDS: three cubes per cell (u, f_y, f_z) + the batch's distinct-dof values.  Lifetimes:
//   x values [load, B2)   u cube [B1, B3)   f_y/f_z [B2, B4)   y partial sums [B3, end)
// ALIAS: x values live in the f_y region and the y sums in the u region (one more barrier).
template <typename T, int P, int CPB, bool ALIAS, bool PADLDS, int MINW, int GPRE, bool ORDERED, bool RUNS>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()), MINW)
    stiffness_plan_kernel(const T* __restrict__ x, const T* __restrict__ cell_constants, T* __restrict__ y,
                          const T* __restrict__ G, const int32_t* __restrict__ nu,
                          const int32_t* __restrict__ udofs, const uint16_t* __restrict__ slot,
                          const T* __restrict__ dphi, int64_t ncell, int xcd_remap,
                          const int32_t* __restrict__ order, const int32_t* __restrict__ runs, LaunchSignal sig) {
  using Sh = PlanShape<T, P, CPB, PADLDS>;
  constexpr int n = Sh::n, n2 = Sh::n2, Nd = Sh::Nd, S = Sh::S, BLOCK = Sh::BLOCK, M = Sh::M, SPT = Sh::SPT;
  static_assert(GPRE >= 1 && GPRE <= n, "GPRE: slabs of G held in registers");
  launch_signal_publish(sig);

  __shared__ T sD[n2 + 1];  // + 1: plan_table_store
  __shared__ T su[CPB * S];
  __shared__ T sfy[CPB * S];
  __shared__ T sfz[CPB * S];
  __shared__ PlanAcc sacc[PlanOwnAcc<T, ALIAS>::value ? M : 1];
  T* const sx = ALIAS ? sfy : reinterpret_cast<T*>(sacc);  // x values of the batch's distinct dofs
  PlanAcc* const sy = PlanOwnAcc<T, ALIAS>::value ? sacc : reinterpret_cast<PlanAcc*>(su);  // their y partial sums

  const int tid = threadIdx.x;
  const unsigned batch = remap_block(blockIdx.x, gridDim.x, xcd_remap);
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
  T g[GPRE][6];
  const int64_t cell = plan_row<ORDERED>(row, pos_ld);  // row of the per-cell arrays
  const T* Gc = G + (cell * Nd + t) * 6;
  PlanSlotWord<n> sraw[n];  // narrowed once the gather is on its way (plan.hpp, PlanSlotWord)
  if (plan_loads_by_all<n>() || active) {
    const uint16_t* sp = slot + pos_ld * Nd + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sraw[ix] = sp[ix * n2];
#pragma unroll
    for (int ix = 0; ix < GPRE; ++ix) load_g6<T>(Gc + (int64_t)ix * n2 * 6, g[ix]);
  }
  const int packed = nu[batch];
  const int nu_b = packed & 0xffff, nr_b = plan_runs_of<RUNS>(packed);
  plan_table_store<n, n2>(sD, tid, dval);
  batch_dofs_resolve<RUNS, SPT, BLOCK>(rt, ud, M, nu_b, nr_b, tid, reinterpret_cast<int32_t*>(su), mydof);

  const T coeff = cell_constants[cell];  // with the gather: it scales u there (c K u = K (c u)), so the main loop holds no constant
  T u[n];
  plan_gather_x<T, n, n2, SPT, BLOCK>(x, mydof, nu_b, tid, active, sraw, sl, sx, su + lc * S + t, u, coeff);

  if constexpr (!ALIAS) plan_zero<T, SPT, BLOCK>(sy, nu_b, tid);  // x values are dead: the buffer becomes the y accumulator

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
      const T* gq = g[qx % GPRE];
      fx[qx] = gq[0] * vx + gq[1] * vy + gq[2] * vz;
      cfy[qx * n2] = gq[1] * vx + gq[3] * vy + gq[4] * vz;
      cfz[qx * n2] = gq[2] * vx + gq[4] * vy + gq[5] * vz;
      if constexpr (GPRE < n) {
        if (qx + GPRE < n) load_g6<T>(Gc + (int64_t)(qx + GPRE) * n2 * 6, g[qx % GPRE]);
      }
    }
  }
  __syncthreads();
  if constexpr (ALIAS) {  // the u cube is dead: zero it as the y accumulator
    plan_zero<T, SPT, BLOCK>(sy, nu_b, tid);
    __syncthreads();
  }

  plan_backward<T, n, n2>(dphi, sD, ty, tz, active, fx, sfy + lc * S + tz, sfz + lc * S + ty * n, sl, sy);
  plan_flush<T, SPT, BLOCK>(y, mydof, nu_b, tid, sy);
}

template <typename T, int P, bool ALIAS, bool PADLDS, int MINW, int GPRE = P + 1>
inline hipError_t launch_stiffness_plan(const T* x, const T* cc, T* y, const T* G, const void* workspace,
                                        const T* dphi, int64_t ncell, int xcd_remap, hipStream_t stream,
                                        bool ordered = false, bool use_runs = false) {
  constexpr int CPB = plan_cells_per_batch<P>();
  if (ncell <= 0) return hipSuccess;
  PlanView v = plan_view(const_cast<void*>(workspace), P, CPB, ncell);
  constexpr int threads = col_block_threads<P, CPB>();
  const LaunchSignal sig = take_launch_signal(stream);
  plan_dispatch(ordered, use_runs, [&](auto o, auto r) {
    hipLaunchKernelGGL((stiffness_plan_kernel<T, P, CPB, ALIAS, PADLDS, MINW, GPRE, decltype(o)::value, decltype(r)::value>),
                       dim3((unsigned)v.nbatch), dim3(threads), 0, stream, x, cc, y, G, v.nu, v.udofs, v.slot, dphi, ncell,
                       xcd_remap, v.order, v.runs, sig);
  });
  return settle_launch_signal(stream, sig, hipGetLastError());
}

}  // namespace fus
