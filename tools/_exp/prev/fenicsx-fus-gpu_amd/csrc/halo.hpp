// Halo pack / unpack kernels: all neighbours in one launch (the reference launches one tiny
// kernel per neighbour, cuda/scatterer.py:155-160,178-183,242-247,265-272).
//   pack_fwd   out[i] = in[index[i]]            cuda/scatterer.py:18-35
//   unpack_fwd out[index[i] + N] = in[i]        cuda/scatterer.py:38-57
//   pack_rev   out[i] = in[index[i] + N]        cuda/scatterer.py:60-79
//   unpack_rev out[index[i]] += in[i] (atomic)  cuda/scatterer.py:82-101
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fus {

enum HaloMode { PACK = 0, UNPACK_SET = 1, UNPACK_ADD = 2 };

template <typename T, int MODE>
__global__ void __launch_bounds__(256)
    halo_kernel(const T* __restrict__ in, T* __restrict__ out, const int64_t* __restrict__ index, int64_t count,
                int64_t offset) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const int64_t j = index[i] + offset;
    if constexpr (MODE == PACK)
      out[i] = in[j];
    else if constexpr (MODE == UNPACK_SET)
      out[j] = in[i];
    else
      unsafeAtomicAdd(out + j, in[i]);
  }
}

template <typename T, int MODE>
inline hipError_t launch_halo(const T* in, T* out, const int64_t* index, int64_t count, int64_t offset,
                              hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  int64_t nblocks = (count + 255) / 256;
  if (nblocks > 2048) nblocks = 2048;
  hipLaunchKernelGGL((halo_kernel<T, MODE>), dim3((unsigned)nblocks), dim3(256), 0, stream, in, out, index, count,
                     offset);
  return hipGetLastError();
}

}  // namespace fus
