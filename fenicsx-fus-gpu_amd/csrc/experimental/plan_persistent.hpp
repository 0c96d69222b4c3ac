// Persistent, software-pipelined form of the planned stiffness apply (raw plans).
//
// Why (profiles/r01d_ablation.log): with the G, x and atomic traffic removed the one-batch-per-
// workgroup kernel still takes 0.13 ms of its 0.23 ms -- each workgroup pays the chain
// "index load -> dependent x gather -> barrier -> ..." once per batch with only 3 workgroups per
// CU to hide it.  Here a workgroup walks batches b, b + grid, b + 2 grid, ... and loads the NEXT
// batch's dof list and slots while it computes the current one, and gathers the next batch's x
// values before it flushes the current one, so only the G stream (issued at the top of an
// iteration, needed two barriers later) is left on the per-batch critical path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../plan.hpp"

namespace fus {

template <typename T, int P, int CPB, bool ALIAS, int MINW>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()), MINW)
    stiffness_plan_persistent_kernel(const T* __restrict__ x, const T* __restrict__ cell_constants,
                                     T* __restrict__ y, const T* __restrict__ G, const int32_t* __restrict__ nu,
                                     const int32_t* __restrict__ udofs, const uint16_t* __restrict__ slot,
                                     const T* __restrict__ dphi, int64_t ncell, int nbatch) {
  constexpr int n = P + 1, n2 = n * n, Nd = n2 * n;
  constexpr int S = lds_cell_stride<T, P>();
  constexpr int BLOCK = col_block_threads<P, CPB>();
  constexpr int M = CPB * Nd;
  constexpr int SPT = (M + BLOCK - 1) / BLOCK;

  __shared__ T sD[n2];
  __shared__ T su[CPB * S];
  __shared__ T sfy[CPB * S];
  __shared__ T sfz[CPB * S];
  __shared__ T sxy_own[ALIAS ? 1 : M];
  T* const sx = ALIAS ? sfy : sxy_own;
  T* const sy = ALIAS ? su : sxy_own;

  const int tid = threadIdx.x;
  const int lc = tid / n2;
  const int t = tid - lc * n2;
  const int ty = t / n, tz = t - ty * n;
  const int stride = gridDim.x;

  if (tid < n2) sD[tid] = dphi[tid];

  // ---- prologue: indices and x values of the first batch ---------------------------------------
  int batch = blockIdx.x;
  int nu_b = 0;
  int32_t mydof[SPT];
  uint16_t sl[n];
  T xv[SPT];
#pragma unroll
  for (int ix = 0; ix < n; ++ix) sl[ix] = 0;
  if (batch < nbatch) {
    nu_b = nu[batch] & 0xffff;
    const int32_t* ud = udofs + (int64_t)batch * M;
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      mydof[r] = ud[s < M ? s : 0];
    }
    const int64_t cell = (int64_t)batch * CPB + lc;
    if (lc < CPB && cell < ncell) {
      const uint16_t* sp = slot + cell * Nd + t;
#pragma unroll
      for (int ix = 0; ix < n; ++ix) sl[ix] = sp[ix * n2];
    }
#pragma unroll
    for (int r = 0; r < SPT; ++r) xv[r] = x[mydof[r]];
  }

  for (; batch < nbatch; batch += stride) {
    const int64_t cell = (int64_t)batch * CPB + lc;
    const bool active = (lc < CPB) && (cell < ncell);
    const int nbatch_next = batch + stride;
    const bool has_next = nbatch_next < nbatch;

    // ---- this batch's G stream, then the next batch's indices ----------------------------------
    T g[n][6];
    T coeff = T(0);
    if (active) {
      const T* Gc = G + (cell * Nd + t) * 6;
#pragma unroll
      for (int ix = 0; ix < n; ++ix) load_g6<T>(Gc + (int64_t)ix * n2 * 6, g[ix]);
      coeff = cell_constants[cell];
    }
    int nu_n = 0;
    int32_t mydof_n[SPT];
    uint16_t sl_n[n];
#pragma unroll
    for (int r = 0; r < SPT; ++r) mydof_n[r] = 0;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sl_n[ix] = 0;
    if (has_next) {
      nu_n = nu[nbatch_next] & 0xffff;
      const int32_t* udn = udofs + (int64_t)nbatch_next * M;
#pragma unroll
      for (int r = 0; r < SPT; ++r) {
        const int s = tid + r * BLOCK;
        mydof_n[r] = udn[s < M ? s : 0];
      }
      const int64_t celln = (int64_t)nbatch_next * CPB + lc;
      if (lc < CPB && celln < ncell) {
        const uint16_t* sp = slot + celln * Nd + t;
#pragma unroll
        for (int ix = 0; ix < n; ++ix) sl_n[ix] = sp[ix * n2];
      }
    }

#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      if (s < nu_b) sx[s] = xv[r];
    }
    __syncthreads();  // B1

    T u[n];
    if (active) {
      T* cu = su + lc * S + t;
#pragma unroll
      for (int ix = 0; ix < n; ++ix) {
        u[ix] = sx[sl[ix]];
        cu[ix * n2] = u[ix];
      }
    }
    __syncthreads();  // B2
    if constexpr (!ALIAS) {
#pragma unroll
      for (int r = 0; r < SPT; ++r) {
        const int s = tid + r * BLOCK;
        if (s < nu_b) sy[s] = T(0);
      }
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
        T vx = T(0);
#pragma unroll
        for (int ix = 0; ix < n; ++ix) vx += dphi[qx * n + ix] * u[ix];
        T vy = T(0), vz = T(0);
#pragma unroll
        for (int i = 0; i < n; ++i) {
          vy += dy[i] * cu_y[qx * n2 + i * n];
          vz += dz[i] * cu_z[qx * n2 + i];
        }
        const T* gq = g[qx];
        fx[qx] = coeff * (gq[0] * vx + gq[1] * vy + gq[2] * vz);
        cfy[qx * n2] = coeff * (gq[1] * vx + gq[3] * vy + gq[4] * vz);
        cfz[qx * n2] = coeff * (gq[2] * vx + gq[4] * vy + gq[5] * vz);
      }
    }
    __syncthreads();  // B3
    if constexpr (ALIAS) {
#pragma unroll
      for (int r = 0; r < SPT; ++r) {
        const int s = tid + r * BLOCK;
        if (s < nu_b) sy[s] = T(0);
      }
      __syncthreads();
    }

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
        T acc = T(0);
#pragma unroll
        for (int qx = 0; qx < n; ++qx) acc += dphi[qx * n + jx] * fx[qx];
#pragma unroll
        for (int q = 0; q < n; ++q) {
          acc += dyT[q] * cf_y[jx * n2 + q * n];
          acc += dzT[q] * cf_z[jx * n2 + q];
        }
        lds_atomic_add(&sy[sl[jx]], acc);
      }
    }
    // the next batch's x gather: its dof list arrived during the phases above
    T xv_n[SPT];
#pragma unroll
    for (int r = 0; r < SPT; ++r) xv_n[r] = has_next ? x[mydof_n[r]] : T(0);
    __syncthreads();  // B4

#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      if (s < nu_b) unsafeAtomicAdd(y + mydof[r], sy[s]);
    }
    __syncthreads();  // B5: LDS is reused by the next batch

    nu_b = nu_n;
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      mydof[r] = mydof_n[r];
      xv[r] = xv_n[r];
    }
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sl[ix] = sl_n[ix];
  }
}

template <typename T, int P, bool ALIAS, int MINW>
inline hipError_t launch_stiffness_plan_persistent(const T* x, const T* cc, T* y, const T* G, const void* workspace,
                                                   const T* dphi, int64_t ncell, int blocks_per_cu,
                                                   hipStream_t stream) {
  constexpr int CPB = plan_cells_per_batch<P>();
  if (ncell <= 0) return hipSuccess;
  PlanView v = plan_view(const_cast<void*>(workspace), P, CPB, ncell);
  constexpr int threads = col_block_threads<P, CPB>();
  int64_t grid = 256LL * blocks_per_cu;
  if (grid > v.nbatch) grid = v.nbatch;
  hipLaunchKernelGGL((stiffness_plan_persistent_kernel<T, P, CPB, ALIAS, MINW>), dim3((unsigned)grid), dim3(threads), 0,
                     stream, x, cc, y, G, v.nu, v.udofs, v.slot, dphi, ncell, (int)v.nbatch);
  return hipGetLastError();
}

}  // namespace fus
