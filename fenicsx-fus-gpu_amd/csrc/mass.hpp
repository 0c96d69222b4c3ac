// Mass operator apply (GLL collocation => diagonal per entity):
//   y[dofmap[e][i]] += x[dofmap[e][i]] * detJ[e][i] * entity_constants[e]
// replaces numba-cpu/operators.py:50-66 and cuda/operators.py:18-70.
// One thread per (entity, local dof); detJ / dofmap are read fully coalesced, x is gathered,
// the scatter-add is the hardware FP atomic.  Used for cells (N = n^3) and boundary facets (N = n^2).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

}  // namespace fus
