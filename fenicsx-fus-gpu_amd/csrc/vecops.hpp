// Streaming vector kernels of the RK4 stage: axpy, copy, fill, pointwise_divide, square
// (cuda/operators.py:195-274, numba-cpu/operators.py:230-300).  HBM-bound: 16-byte accesses per
// lane when the operands allow it, grid capped at 8 workgroups per CU and grid-strided.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fus {

template <typename T>
struct vec16;
template <>
struct vec16<double> {
  using type = double2;
  static constexpr int W = 2;
};
template <>
struct vec16<float> {
  using type = float4;
  static constexpr int W = 4;
};

template <typename T>
struct OpAxpy {  // y = alpha x + y
  T alpha;
  __device__ __forceinline__ T operator()(T a, T b) const { return alpha * a + b; }
};
template <typename T>
struct OpScale {  // out = alpha a
  T alpha;
  __device__ __forceinline__ T operator()(T a, T) const { return alpha * a; }
};
template <typename T>
struct OpCopy {  // out = a
  __device__ __forceinline__ T operator()(T a, T) const { return a; }
};
template <typename T>
struct OpFill {  // out = alpha
  T alpha;
  __device__ __forceinline__ T operator()(T, T) const { return alpha; }
};
template <typename T>
struct OpDiv {  // out = a / b
  __device__ __forceinline__ T operator()(T a, T b) const { return a / b; }
};
template <typename T>
struct OpSquare {  // out = a * a
  __device__ __forceinline__ T operator()(T a, T) const { return a * a; }
};

template <typename Op>
__device__ __forceinline__ double2 apply2(const double2& a, const double2& b, const Op& op) {
  return double2{op(a.x, b.x), op(a.y, b.y)};
}
template <typename Op>
__device__ __forceinline__ float4 apply2(const float4& a, const float4& b, const Op& op) {
  return float4{op(a.x, b.x), op(a.y, b.y), op(a.z, b.z), op(a.w, b.w)};
}

// out[i] = op(a[i], b[i]);  USE_A / USE_B say which inputs are actually read.
template <typename T, typename Op, bool USE_A, bool USE_B, bool VEC>
__global__ void __launch_bounds__(256)
    ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n, Op op) {
  using V = typename vec16<T>::type;
  constexpr int W = vec16<T>::W;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if constexpr (VEC) {
    const int64_t nv = n / W;
    const V* av = reinterpret_cast<const V*>(a);
    const V* bv = reinterpret_cast<const V*>(b);
    V* ov = reinterpret_cast<V*>(out);
    for (int64_t i = gid; i < nv; i += stride) {
      V va{}, vb{};
      if constexpr (USE_A) va = av[i];
      if constexpr (USE_B) vb = bv[i];
      ov[i] = apply2(va, vb, op);
    }
    const int64_t i = nv * W + gid;  // tail
    if (i < n) {
      T sa = T(0), sb = T(0);
      if constexpr (USE_A) sa = a[i];
      if constexpr (USE_B) sb = b[i];
      out[i] = op(sa, sb);
    }
  } else {
    for (int64_t i = gid; i < n; i += stride) {
      T sa = T(0), sb = T(0);
      if constexpr (USE_A) sa = a[i];
      if constexpr (USE_B) sb = b[i];
      out[i] = op(sa, sb);
    }
  }
}

// y[i] += w[i] * x[i]: the cell mass apply in CACHED-DIAGONAL form.  With GLL collocation the mass operator of
// numba-cpu/operators.py:19-68 is diagonal, M(c) x = (M(c) 1) (.) x, so a driver that applies the same M(c) many times
// assembles w = M(c) 1 once (one gather-scale-scatter apply) and applies 3 vector touches per dof afterwards instead of
// 47.6 B/dof of gather / scatter.  Opt-in (operators.diagonal_mass_operator), its own bytes contract.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) muladd_kernel(const T* __restrict__ w, const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  using V = typename vec16<T>::type;
  constexpr int W = vec16<T>::W;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if constexpr (VEC) {
    const int64_t nv = n / W;
    const V* wv = reinterpret_cast<const V*>(w);
    const V* xv = reinterpret_cast<const V*>(x);
    V* yv = reinterpret_cast<V*>(y);
    for (int64_t i = gid; i < nv; i += stride) {
      const V a = wv[i], b = xv[i];
      V c = yv[i];
      if constexpr (W == 2) {
        c.x += a.x * b.x;
        c.y += a.y * b.y;
      } else {
        c.x += a.x * b.x;
        c.y += a.y * b.y;
        c.z += a.z * b.z;
        c.w += a.w * b.w;
      }
      yv[i] = c;
    }
    const int64_t i = nv * W + gid;
    if (i < n) y[i] += w[i] * x[i];
  } else {
    for (int64_t i = gid; i < n; i += stride) y[i] += w[i] * x[i];
  }
}

template <typename T>
inline hipError_t launch_muladd(const T* w, const T* x, T* y, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  constexpr int W = vec16<T>::W;
  const bool aligned = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
  const int64_t work = aligned ? (n + W - 1) / W : n;
  int64_t nblocks = (work + 255) / 256;
  if (nblocks > 2048) nblocks = 2048;
  if (aligned)
    hipLaunchKernelGGL((muladd_kernel<T, true>), dim3((unsigned)nblocks), dim3(256), 0, stream, w, x, y, n);
  else
    hipLaunchKernelGGL((muladd_kernel<T, false>), dim3((unsigned)nblocks), dim3(256), 0, stream, w, x, y, n);
  return hipGetLastError();
}

template <typename T, typename Op, bool USE_A, bool USE_B>
inline hipError_t launch_ew(const T* a, const T* b, T* out, int64_t n, Op op, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  constexpr int W = vec16<T>::W;
  const bool aligned = ((reinterpret_cast<uintptr_t>(out) | (USE_A ? reinterpret_cast<uintptr_t>(a) : 0) |
                         (USE_B ? reinterpret_cast<uintptr_t>(b) : 0)) & 15u) == 0;
  const int64_t work = aligned ? (n + W - 1) / W : n;
  int64_t nblocks = (work + 255) / 256;
  if (nblocks > 2048) nblocks = 2048;
  if (aligned)
    hipLaunchKernelGGL((ew_kernel<T, Op, USE_A, USE_B, true>), dim3((unsigned)nblocks), dim3(256), 0, stream, a, b,
                       out, n, op);
  else
    hipLaunchKernelGGL((ew_kernel<T, Op, USE_A, USE_B, false>), dim3((unsigned)nblocks), dim3(256), 0, stream, a, b,
                       out, n, op);
  return hipGetLastError();
}

}  // namespace fus
