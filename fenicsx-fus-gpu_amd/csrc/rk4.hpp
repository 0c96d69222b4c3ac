// Fused RK4 stage vector kernel.
//
// Between two operator applications the reference launches, per stage, 12 streaming kernels
// (cuda/demo_linear_box.py:491-563: 5 copy, 4 axpy, 2 fill, 1 pointwise_divide = 216 B/dof).
// Everything after scatter_rev(b) of stage i and before scatter_fwd of stage i+1 is elementwise,
// so it is ONE kernel here (88-104 B/dof):
//     kv  = b * minv                       pointwise_divide(b, m, kv)   :556   (1/m precomputed, cf. the
//                                                                       "store 1/m" TODO cpp/common/Linear.hpp:216-218)
//     u  += bw * ku ;  v += bw * kv        axpy x2                      :562-563   bw = b_runge[i] dt
//     [new step: u0 = u ; v0 = v]          copy x2                      :491-492
//     un  = u0 + aw * ku                   copy + axpy                  :496,499    aw = a_runge[i+1] dt
//     vn  = v0 + aw * kv                   copy + axpy                  :497,500
//     ku  = vn                             copy (f0)                    :508  (ku doubles as v_n: same values)
//     b   = 0                              fill                         :541
// kv is never stored.  Updates run over the owned dofs [0, nlocal); b is zeroed over
// [0, ntotal) (owned + ghosts), ghost values of un / ku are refreshed by the forward scatter.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fus {

// ``kind`` (the ABI's ``new_step`` argument):
//   0 MIDDLE   stages 2, 3:  reads b minv ku u v u0 v0, writes u v un ku b            (12 vector touches)
//   1 legacy   last stage that also materialises the next step's stage inputs
//              (u0 = u, v0 = v, un = u, ku = v): reads 5, writes 7
//   2 FIRST    stage 1 of a step whose inputs ARE (u0, v0) -- the driver hands u0 / v0 to the operator
//              instead of copies of them: reads b minv u0 v0, writes u v un ku b        (9 touches)
//   3 LAST     stage 4: the new solution goes straight to u0 / v0, nothing else is materialised:
//              reads b minv ku u v, writes u0 v0 b                                      (8 touches)
// A step run as FIRST, MIDDLE, MIDDLE, LAST moves 41 vector touches instead of 48 with the same
// arithmetic in the same order (bitwise the same u, v).
template <typename T>
__global__ void __launch_bounds__(256)
    rk4_stage_kernel(T bw, T aw, int kind, const T* __restrict__ minv, T* __restrict__ b, T* __restrict__ u,
                     T* __restrict__ v, T* __restrict__ u0, T* __restrict__ v0, T* __restrict__ ku,
                     T* __restrict__ un, int64_t nlocal, int64_t ntotal) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ntotal; i += stride) {
    if (i < nlocal) {
      const T kv = b[i] * minv[i];
      if (kind == 2) {  // FIRST: u == u0, v == v0, ku == v0
        const T u0i = u0[i], v0i = v0[i];
        u[i] = u0i + bw * v0i;
        v[i] = v0i + bw * kv;
        un[i] = u0i + aw * v0i;
        ku[i] = v0i + aw * kv;
      } else if (kind == 3) {  // LAST
        u0[i] = u[i] + bw * ku[i];
        v0[i] = v[i] + bw * kv;
      } else {
        const T kui = ku[i];
        const T ui = u[i] + bw * kui;
        const T vi = v[i] + bw * kv;
        u[i] = ui;
        v[i] = vi;
        T u0i, v0i;
        if (kind == 1) {
          u0i = ui;
          v0i = vi;
          u0[i] = ui;
          v0[i] = vi;
        } else {
          u0i = u0[i];
          v0i = v0[i];
        }
        un[i] = u0i + aw * kui;
        ku[i] = v0i + aw * kv;
      }
    }
    b[i] = T(0);
  }
}

template <typename T>
inline hipError_t launch_rk4_stage(T bw, T aw, int new_step, const T* minv, T* b, T* u, T* v, T* u0, T* v0, T* ku,
                                   T* un, int64_t nlocal, int64_t ntotal, hipStream_t stream) {
  if (ntotal <= 0) return hipSuccess;
  int64_t nblocks = (ntotal + 255) / 256;
  if (nblocks > 4096) nblocks = 4096;
  hipLaunchKernelGGL((rk4_stage_kernel<T>), dim3((unsigned)nblocks), dim3(256), 0, stream, bw, aw, new_step, minv, b,
                     u, v, u0, v0, ku, un, nlocal, ntotal);
  return hipGetLastError();
}

}  // namespace fus
