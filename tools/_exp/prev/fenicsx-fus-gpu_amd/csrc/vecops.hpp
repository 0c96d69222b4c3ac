// Streaming vector kernels of the RK4 stage: axpy, copy, fill, pointwise_divide, square
// (cuda/operators.py:195-274, numba-cpu/operators.py:230-300).  HBM-bound: 16-byte accesses per
// lane when the operands allow it, grid capped at 8 workgroups per CU and grid-strided.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fus {

// Streaming accesses.  A vector kernel on operands far larger than the caches re-reads nothing it touches, and -- what
// matters -- every line it WRITES with a plain store stays dirty in the 256 MB memory-side Infinity Cache and is written
// back while the NEXT kernel runs: the headline apply takes 219-222 us after nothing / a busy wait / a 1 GiB read-only
// stream, 287 us after a 1 GiB fill, 263 us after a 1 GiB copy, 251-272 us after the RK4 vector pass
// (tools/interleave_probe.py, profiles/r04i_interleave_probe.log).  Non-temporal stores go to memory without lingering.
// Loads too: most of these kernels update in place (u += ..., y += w x), and a non-temporal store to a line the kernel's own
// plain load has just brought into the cache hits there and leaves it dirty all the same -- fused RK4 step on one box: cached
// 1.621 ms, non-temporal stores only 1.569, loads and stores 1.484 (profiles/r04k_ab_vector_stream_policy.log).
// Kernel template parameter NT: 0 cached accesses, 1 non-temporal loads AND stores, 2 non-temporal stores only.
// mode (fus_set_tuning FUS_TUNE_VECTOR_STREAM): 0 never; 1 auto = loads and stores for operands > kStreamBytes (default);
// 2 always loads and stores; 3 auto, stores only; 4 always, stores only
constexpr int64_t kStreamBytes = 24ll << 20;
inline int& vector_stream_mode() {
  static int m = 1;
  return m;
}
inline int vector_stream(int64_t operand_bytes) {
  const int m = vector_stream_mode();
  const bool on = m == 2 || m == 4 || ((m == 1 || m == 3) && operand_bytes > kStreamBytes);
  return !on ? 0 : (m >= 3 ? 2 : 1);
}
template <int NT, typename V>
__device__ __forceinline__ V ld_stream(const V* p) {
  if constexpr (NT == 1)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}
template <int NT, typename V>
__device__ __forceinline__ void st_stream(V* p, V v) {
  if constexpr (NT != 0)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

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

// out[i] = op(a[i], b[i]);  USE_A / USE_B say which inputs are actually read.  NT: streaming (non-temporal) accesses.
template <typename T, typename Op, bool USE_A, bool USE_B, bool VEC, int NT>
__global__ void __launch_bounds__(256)
    ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n, Op op) {
  using V = typename vec16<T>::type;
  constexpr int W = vec16<T>::W;
  typedef T VN __attribute__((ext_vector_type(W)));  // native vector: what the non-temporal builtins take
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if constexpr (VEC) {
    const int64_t nv = n / W;
    const VN* av = reinterpret_cast<const VN*>(a);
    const VN* bv = reinterpret_cast<const VN*>(b);
    VN* ov = reinterpret_cast<VN*>(out);
    for (int64_t i = gid; i < nv; i += stride) {
      V va{}, vb{};
      if constexpr (USE_A) {
        const VN t = ld_stream<NT>(av + i);
        __builtin_memcpy(&va, &t, sizeof(V));
      }
      if constexpr (USE_B) {
        const VN t = ld_stream<NT>(bv + i);
        __builtin_memcpy(&vb, &t, sizeof(V));
      }
      const V r = apply2(va, vb, op);
      VN t;
      __builtin_memcpy(&t, &r, sizeof(V));
      st_stream<NT>(ov + i, t);
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
      if constexpr (USE_A) sa = ld_stream<NT>(a + i);
      if constexpr (USE_B) sb = ld_stream<NT>(b + i);
      st_stream<NT>(out + i, op(sa, sb));
    }
  }
}

// y[i] += w[i] * x[i]: the cell mass apply in CACHED-DIAGONAL form.  With GLL collocation the mass operator of
// numba-cpu/operators.py:19-68 is diagonal, M(c) x = (M(c) 1) (.) x, so a driver that applies the same M(c) many times
// assembles w = M(c) 1 once (one gather-scale-scatter apply) and applies 3 vector touches per dof afterwards instead of
// 47.6 B/dof of gather / scatter.  Opt-in (operators.diagonal_mass_operator), its own bytes contract.
template <typename T, bool VEC, int NT>
__global__ void __launch_bounds__(256) muladd_kernel(const T* __restrict__ w, const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  constexpr int W = vec16<T>::W;
  typedef T VN __attribute__((ext_vector_type(W)));
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if constexpr (VEC) {
    const int64_t nv = n / W;
    const VN* wv = reinterpret_cast<const VN*>(w);
    const VN* xv = reinterpret_cast<const VN*>(x);
    VN* yv = reinterpret_cast<VN*>(y);
    for (int64_t i = gid; i < nv; i += stride) {
      const VN a = ld_stream<NT>(wv + i), b = ld_stream<NT>(xv + i);
      VN c = ld_stream<NT>(yv + i);
      c += a * b;  // element-wise
      st_stream<NT>(yv + i, c);
    }
    const int64_t i = nv * W + gid;
    if (i < n) y[i] += w[i] * x[i];
  } else {
    for (int64_t i = gid; i < n; i += stride) st_stream<NT>(y + i, ld_stream<NT>(y + i) + ld_stream<NT>(w + i) * ld_stream<NT>(x + i));
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
  const int nt = vector_stream(n * (int64_t)sizeof(T));
#define FUS_MA(VEC_, NT_) hipLaunchKernelGGL((muladd_kernel<T, VEC_, NT_>), dim3((unsigned)nblocks), dim3(256), 0, stream, w, x, y, n)
  if (aligned) {
    if (nt == 1) FUS_MA(true, 1); else if (nt == 2) FUS_MA(true, 2); else FUS_MA(true, 0);
  } else {
    if (nt == 1) FUS_MA(false, 1); else if (nt == 2) FUS_MA(false, 2); else FUS_MA(false, 0);
  }
#undef FUS_MA
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
  const int nt = vector_stream(n * (int64_t)sizeof(T));
#define FUS_EW(VEC_, NT_) \
  hipLaunchKernelGGL((ew_kernel<T, Op, USE_A, USE_B, VEC_, NT_>), dim3((unsigned)nblocks), dim3(256), 0, stream, a, b, out, n, op)
#define FUS_EW3(VEC_) \
  if (nt == 1)        \
    FUS_EW(VEC_, 1);  \
  else if (nt == 2)   \
    FUS_EW(VEC_, 2);  \
  else                \
    FUS_EW(VEC_, 0)
  if (aligned) {
    FUS_EW3(true);
  } else {
    FUS_EW3(false);
  }
#undef FUS_EW3
#undef FUS_EW
  return hipGetLastError();
}

}  // namespace fus
