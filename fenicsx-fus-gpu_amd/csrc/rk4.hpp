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

#include <type_traits>

#include "vecops.hpp"

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
//
// LEAN kinds 4, 5, 6, 7 (round 6; one per stage of a step, used as a set; bw = b_runge[0] dt = dt / 6, aw = a_runge[1] dt = dt / 2, the
// other coefficients are their exact doubles): 34 touches.  The floor argument: b must be complete before kv = b / m, so
// every pass reads b, minv and re-zeroes b (3); passes 1-3 must write the next stage's (un, vn) (2) and need (u0, v0) for them (2);
// passes 2, 3 read vn of their own stage (1): 26.  What is left is the traffic of the two accumulators (15 of the 41), and it shrinks because
//   * u's increments are the vn's, each KNOWN ONE PASS EARLY (vn_{i+1} is formed in pass i): pass 2 writes u0 + b1 v0 + b2 vn2 + b3 vn3,
//     pass 3 adds b4 vn4 and writes the NEW u straight into u0 (u0 is dead once un4 has been formed) -- pass 4 does not touch u at all;
//   * pass 1 writes neither accumulator: pass 2 reads u0, v0, vn2 anyway and re-derives u0 + b1 v0 and b1 kv1 = (vn2 - v0) b1 / a2 (a
//     first-order difference scaled by b1 / a2 = 1 / 3: absolute error eps |v0| / 3, the size of v's own rounding).
//   4 FIRST'   reads b minv u0 v0,            writes un ku b            (7)
//   5 SECOND'  reads b minv u0 v0 ku,         writes u v un ku b        (10)   u = accumulator INCLUDING b3 vn3
//   6 THIRD'   reads b minv u0 v0 ku u v,     writes u0 v un ku b       (12)   u0 = new u
//   7 LAST'    reads b minv v,                writes v0 b               (5)    v0 = new v
// u is formed with the reference's operations in the reference's order (bitwise the same); v differs from the sequence above in the
// rounding of b1 kv1 only.  Deriving more (v's accumulator in pass 3 from un and u0) would divide a difference of u's by dt twice:
// not done.
// one dof of the stage (all operands in registers): the arithmetic of the table above
template <typename T>
struct Rk4In {
  T b, minv, u, v, u0, v0, ku;
};
template <typename T>
struct Rk4Out {
  T u, v, u0, v0, un, ku;
};
// which vectors a stage kind reads / writes (b and minv are always read, b is always re-zeroed)
struct Rk4Access {
  bool rd_u, rd_v, rd_0, rd_ku, wr_u, wr_v, wr_n, wr_u0, wr_v0;
};
__host__ __device__ __forceinline__ Rk4Access rk4_access(int kind) {
  switch (kind) {
    case 1: return {true, true, false, true, true, true, true, true, true};
    case 2: return {false, false, true, false, true, true, true, false, false};
    case 3: return {true, true, false, true, false, false, false, true, true};
    case 4: return {false, false, true, false, false, false, true, false, false};
    case 5: return {false, false, true, true, true, true, true, false, false};
    case 6: return {true, true, true, true, false, true, true, true, false};
    case 7: return {false, true, false, false, false, false, false, false, true};
    default: return {true, true, true, true, true, true, true, false, false};  // 0 MIDDLE
  }
}
// vector touches of one pass of kind ``kind`` (b read + zeroed, minv read, + the table above)
__host__ __device__ constexpr int rk4_touches(int kind) {
  constexpr int t[8] = {12, 12, 9, 8, 7, 10, 12, 5};
  return t[kind & 7];
}
template <typename T>
__device__ __forceinline__ Rk4Out<T> rk4_update(int kind, T bw, T aw, const Rk4In<T>& in) {
  Rk4Out<T> o{};
  const T kv = in.b * in.minv;
  if (kind >= 4) {  // LEAN set: bw = dt / 6, aw = dt / 2
    const T b2 = bw + bw, a4 = aw + aw;
    if (kind == 4) {
      o.un = in.u0 + aw * in.v0;
      o.ku = in.v0 + aw * kv;
    } else if (kind == 5) {
      const T vn2 = in.ku, vn3 = in.v0 + aw * kv;
      o.v = (in.v0 + (vn2 - in.v0) * (bw / aw)) + b2 * kv;
      o.u = ((in.u0 + bw * in.v0) + b2 * vn2) + b2 * vn3;
      o.un = in.u0 + aw * vn2;
      o.ku = vn3;
    } else if (kind == 6) {
      const T vn3 = in.ku, vn4 = in.v0 + a4 * kv;
      o.v = in.v + b2 * kv;
      o.u0 = in.u + bw * vn4;
      o.un = in.u0 + a4 * vn3;
      o.ku = vn4;
    } else {
      o.v0 = in.v + bw * kv;
    }
    return o;
  }
  if (kind == 2) {  // FIRST: u == u0, v == v0, ku == v0
    o.u = in.u0 + bw * in.v0;
    o.v = in.v0 + bw * kv;
    o.un = in.u0 + aw * in.v0;
    o.ku = in.v0 + aw * kv;
  } else if (kind == 3) {  // LAST
    o.u0 = in.u + bw * in.ku;
    o.v0 = in.v + bw * kv;
  } else {
    o.u = in.u + bw * in.ku;
    o.v = in.v + bw * kv;
    const T u0i = kind == 1 ? o.u : in.u0, v0i = kind == 1 ? o.v : in.v0;
    o.u0 = u0i;
    o.v0 = v0i;
    o.un = u0i + aw * in.ku;
    o.ku = v0i + aw * kv;
  }
  return o;
}

// W dofs per thread as ONE 16-byte access per array where the arrays are 16-byte aligned (W = 2 doubles / 4 floats), scalar
// otherwise.  NT (vectors far larger than the caches: vecops.hpp vector_stream): EVERY access non-temporal.  The loads: the
// pass re-reads nothing before ~1 GB of other data has gone through the caches.  The stores -- un, ku and b included, although
// the next kernel reads them: a line written with a plain store stays dirty in the memory-side Infinity Cache and is written
// back WHILE THE OPERATOR RUNS (its launch takes 251-272 us after a plain-store vector pass against 219-222 us after a busy
// wait or a read-only stream: profiles/r04i_interleave_probe.log); re-reading 82 MB of un from HBM is the cheaper side.
template <typename T, int W, int NT>
__global__ void __launch_bounds__(256)
    rk4_stage_kernel(T bw, T aw, int kind, const T* __restrict__ minv, T* __restrict__ b, T* __restrict__ u,
                     T* __restrict__ v, T* __restrict__ u0, T* __restrict__ v0, T* __restrict__ ku,
                     T* __restrict__ un, int64_t nlocal, int64_t ntotal) {
  typedef T VW __attribute__((ext_vector_type(W)));  // native vector: what the non-temporal builtins take
  using V = typename std::conditional<W == 1, T, VW>::type;
  const int64_t stride = (int64_t)gridDim.x * 256 * W;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * W; i < ntotal; i += stride) {
    if (i + W <= nlocal) {
      T rb[W], rm[W], ru[W], rv[W], ru0[W], rv0[W], rku[W];
      auto ld = [&](const T* p, T(&r)[W], bool nt) {
        V t = (NT == 1 && nt) ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p + i)) : *reinterpret_cast<const V*>(p + i);
        __builtin_memcpy(r, &t, sizeof(V));
      };
      ld(b, rb, true);
      ld(minv, rm, true);
      const Rk4Access a = rk4_access(kind);
      if (a.rd_u) ld(u, ru, true);
      if (a.rd_v) ld(v, rv, true);
      if (a.rd_0) {
        ld(u0, ru0, true);
        ld(v0, rv0, true);
      }
      if (a.rd_ku) ld(ku, rku, true);
      T ou[W], ov[W], ou0[W], ov0[W], oun[W], oku[W];
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const Rk4Out<T> o = rk4_update<T>(kind, bw, aw, Rk4In<T>{rb[k], rm[k], ru[k], rv[k], ru0[k], rv0[k], rku[k]});
        ou[k] = o.u, ov[k] = o.v, ou0[k] = o.u0, ov0[k] = o.v0, oun[k] = o.un, oku[k] = o.ku;
      }
      auto st = [&](T* p, const T(&r)[W], bool nt) {
        V t;
        __builtin_memcpy(&t, r, sizeof(V));
        if (NT != 0 && nt)
          __builtin_nontemporal_store(t, reinterpret_cast<V*>(p + i));
        else
          *reinterpret_cast<V*>(p + i) = t;
      };
      if (a.wr_u) st(u, ou, true);
      if (a.wr_v) st(v, ov, true);
      if (a.wr_n) {
        st(un, oun, true);
        st(ku, oku, true);
      }
      if (a.wr_u0) st(u0, ou0, true);
      if (a.wr_v0) st(v0, ov0, true);
      T z[W];
#pragma unroll
      for (int k = 0; k < W; ++k) z[k] = T(0);
      st(b, z, true);
    } else {  // the last owned dofs (nlocal not a multiple of W) and the ghost block of b
      for (int64_t j = i; j < i + W && j < ntotal; ++j) {
        if (j < nlocal) {
          const Rk4Access a = rk4_access(kind);
          const Rk4Out<T> o = rk4_update<T>(kind, bw, aw, Rk4In<T>{b[j], minv[j], a.rd_u ? u[j] : T(0), a.rd_v ? v[j] : T(0),
                                                                     a.rd_0 ? u0[j] : T(0), a.rd_0 ? v0[j] : T(0), a.rd_ku ? ku[j] : T(0)});
          if (a.wr_u) u[j] = o.u;
          if (a.wr_v) v[j] = o.v;
          if (a.wr_n) un[j] = o.un, ku[j] = o.ku;
          if (a.wr_u0) u0[j] = o.u0;
          if (a.wr_v0) v0[j] = o.v0;
        }
        b[j] = T(0);
      }
    }
  }
}

template <typename T>
inline hipError_t launch_rk4_stage(T bw, T aw, int new_step, const T* minv, T* b, T* u, T* v, T* u0, T* v0, T* ku,
                                   T* un, int64_t nlocal, int64_t ntotal, hipStream_t stream) {
  if (ntotal <= 0) return hipSuccess;
  constexpr int W = 16 / (int)sizeof(T);
  uintptr_t bits = 0;
  for (const void* p : {(const void*)minv, (const void*)b, (const void*)u, (const void*)v, (const void*)u0, (const void*)v0, (const void*)ku,
                        (const void*)un})
    bits |= reinterpret_cast<uintptr_t>(p);
  const bool aligned = (bits & 15u) == 0;
  const int64_t work = aligned ? (ntotal + W - 1) / W : ntotal;
  int64_t nblocks = (work + 255) / 256;
  if (nblocks > 4096) nblocks = 4096;
  const int nt = vector_stream(ntotal * (int64_t)sizeof(T));
#define FUS_RK4(W_, NT_) \
  hipLaunchKernelGGL((rk4_stage_kernel<T, W_, NT_>), dim3((unsigned)nblocks), dim3(256), 0, stream, bw, aw, new_step, minv, b, u, v, \
                     u0, v0, ku, un, nlocal, ntotal)
  if (aligned) {
    if (nt == 1) FUS_RK4(W, 1); else if (nt == 2) FUS_RK4(W, 2); else FUS_RK4(W, 0);
  } else {
    if (nt == 1) FUS_RK4(1, 1); else if (nt == 2) FUS_RK4(1, 2); else FUS_RK4(1, 0);
  }
#undef FUS_RK4
  return hipGetLastError();
}

}  // namespace fus
