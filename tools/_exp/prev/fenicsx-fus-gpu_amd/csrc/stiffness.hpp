// Sum-factorised stiffness operator apply for GLL spectral hexahedra -- CDNA4 (gfx950) kernels.
//
// Replaces the reference's per-cell operator
//   numba-cpu/operators.py:121-225  (gather, 3 forward 1-D contractions, symmetric-G transform x
//                                    cell constant, 3 transposed contractions, scatter-add)
//   cuda/operators.py:87-190        (one 125-thread block per cell, 4 LDS cubes, AoS G loads)
// Written from the maths, not from either implementation.
//
// Kernel "col" (column-per-thread).  With n = P + 1 and local dof l = tx n^2 + ty n + tz:
//   * one thread owns the (ty, tz) column of a cell and keeps its n values along tx in
//     registers; n^2 threads per cell, CPB cells per workgroup (P = 4: 25 threads/cell, 10 cells
//     per 256-thread workgroup, 250 of 256 lanes busy; 64-lane waves span cells freely because a
//     thread's identity is (cell, ty, tz), not a 3-D block index);
//   * the tx-direction contractions run entirely in registers with the derivative table held in
//     SGPRs (its index is compile-time, so the loads are scalar); only the ty / tz directions go
//     through LDS (one cube for u, two for the transformed fluxes; the third flux stays in
//     registers) -- 24 LDS reads per dof instead of the 60 of a thread-per-dof mapping;
//   * for register slot tx the threads of a cell touch local dofs tx n^2 + (0..n^2-1): the
//     dofmap read is contiguous, the G read is the contiguous 48 n^2-byte slab G[cell][tx n^2 ..]
//     (each lane takes its 48 bytes as three 16-byte loads issued back to back), so the dominant
//     G stream is read exactly once in full lines straight from the reference's AoS layout;
//   * all HBM loads of a cell (dofmap -> x gather, G) are issued before the first barrier.
// Scatter-add uses the hardware floating-point atomic (global_atomic_add_f64 / _f32).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fus {

typedef double fus_double2 __attribute__((ext_vector_type(2)));
typedef float fus_float2 __attribute__((ext_vector_type(2)));
template <typename T>
struct vec2_of;
template <>
struct vec2_of<double> {
  using type = fus_double2;
};
template <>
struct vec2_of<float> {
  using type = fus_float2;
};

// 6 consecutive T (one quadrature point's symmetric G) as three 2-wide vector loads.
// Requires G to be aligned to 2*sizeof(T) (checked on the host).
// NT: non-temporal (streaming) loads -- G is read exactly once per apply.
template <typename T, bool NT = false>
__device__ __forceinline__ void load_g6(const T* __restrict__ p, T (&g)[6]) {
  using V = typename vec2_of<T>::type;
  const V* v = reinterpret_cast<const V*>(p);
  V a, b, c;
  if constexpr (NT) {
    a = __builtin_nontemporal_load(v);
    b = __builtin_nontemporal_load(v + 1);
    c = __builtin_nontemporal_load(v + 2);
  } else {
    a = v[0];
    b = v[1];
    c = v[2];
  }
  g[0] = a.x;
  g[1] = a.y;
  g[2] = b.x;
  g[3] = b.y;
  g[4] = c.x;
  g[5] = c.y;
}

__host__ __device__ constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

// LDS stride between the cubes of consecutive cells: padded so that the few distinct addresses a
// 32-lane group reads in one instruction (n per cell, 1-4 cells per group) fall on distinct banks.
template <typename T, int P>
__host__ __device__ constexpr int lds_cell_stride() {
  constexpr int n = P + 1, Nd = n * n * n;
  constexpr int want = (P < 8) ? 8 : 16;  // stride mod 32
  int s = Nd;
  while (s % 32 != want) ++s;
  return s;
}

template <int P>
__host__ __device__ constexpr int default_cells_per_block(int target_threads) {
  constexpr int n2 = (P + 1) * (P + 1);
  return target_threads / n2 > 0 ? target_threads / n2 : 1;
}

template <int P, int CPB>
__host__ __device__ constexpr int col_block_threads() {
  return round_up(CPB * (P + 1) * (P + 1), 64);
}

// Block index -> batch index.  Workgroups are dealt round-robin over the 8 XCDs (observed, not
// contractual: used for speed only); with the remap each XCD walks a contiguous range of cell
// batches so that neighbouring cells, which share x / y dofs, meet in one L2.
__device__ __forceinline__ unsigned remap_block(unsigned bid, unsigned nblocks, int xcd_remap) {
  if (!xcd_remap) return bid;
  const unsigned per = nblocks >> 3, rem = nblocks & 7u;
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

template <typename T, int P, int CPB>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()))
    stiffness_col_kernel(const T* __restrict__ x, const T* __restrict__ cell_constants, T* __restrict__ y,
                         const T* __restrict__ G, const int32_t* __restrict__ dofmap,
                         const T* __restrict__ dphi, int64_t ncell, int xcd_remap) {
  constexpr int n = P + 1, n2 = n * n, Nd = n2 * n;
  constexpr int S = lds_cell_stride<T, P>();

  __shared__ T sD[n2];
  __shared__ T su[CPB * S];
  __shared__ T sfy[CPB * S];
  __shared__ T sfz[CPB * S];

  const int tid = threadIdx.x;
  const unsigned batch = remap_block(blockIdx.x, gridDim.x, xcd_remap);
  const int lc = tid / n2;       // cell within the batch
  const int t = tid - lc * n2;   // column id = ty * n + tz
  const int ty = t / n, tz = t - ty * n;
  const int64_t cell = (int64_t)batch * CPB + lc;
  const bool active = (lc < CPB) && (cell < ncell);

  if (tid < n2) sD[tid] = dphi[tid];

  int32_t dof[n];
  T u[n];
  T g[n][6];
  T coeff = T(0);
  if (active) {
    const int32_t* dm = dofmap + cell * Nd + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) dof[ix] = dm[ix * n2];
    const T* Gc = G + (cell * Nd + t) * 6;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) load_g6<T>(Gc + (int64_t)ix * n2 * 6, g[ix]);
    coeff = cell_constants[cell];
#pragma unroll
    for (int ix = 0; ix < n; ++ix) u[ix] = x[dof[ix]];
    T* cu = su + lc * S + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) cu[ix * n2] = u[ix];
  }
  __syncthreads();

  T fx[n];
  if (active) {
    // thread-dependent rows of D for the ty / tz directions
    T dy[n], dz[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      dy[i] = sD[ty * n + i];
      dz[i] = sD[tz * n + i];
    }
    const T* cu_y = su + lc * S + tz;       // + tx n^2 + iy n
    const T* cu_z = su + lc * S + ty * n;   // + tx n^2 + iz
    T* cfy = sfy + lc * S + t;
    T* cfz = sfz + lc * S + t;
#pragma unroll
    for (int qx = 0; qx < n; ++qx) {
      // tx-direction: registers x SGPR table (dphi index is compile-time => scalar loads)
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
      T acc = T(0);
#pragma unroll
      for (int qx = 0; qx < n; ++qx) acc += dphi[qx * n + jx] * fx[qx];
#pragma unroll
      for (int q = 0; q < n; ++q) {
        acc += dyT[q] * cf_y[jx * n2 + q * n];
        acc += dzT[q] * cf_z[jx * n2 + q];
      }
      unsafeAtomicAdd(y + dof[jx], acc);
    }
  }
}

template <typename T, int P, int CPB>
inline hipError_t launch_stiffness_col(const T* x, const T* cc, T* y, const T* G, const int32_t* dofmap,
                                       const T* dphi, int64_t ncell, int xcd_remap, hipStream_t stream) {
  if (ncell <= 0) return hipSuccess;
  const int64_t nblocks = (ncell + CPB - 1) / CPB;
  if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
  constexpr int threads = col_block_threads<P, CPB>();
  hipLaunchKernelGGL((stiffness_col_kernel<T, P, CPB>), dim3((unsigned)nblocks), dim3(threads), 0, stream, x, cc, y,
                     G, dofmap, dphi, ncell, xcd_remap);
  return hipGetLastError();
}

}  // namespace fus
