// Fused Westervelt cell kernel.
//
// Per RK4 stage the reference launches, over the SAME cells,
//   mass(u_n, c2) -> m          cuda/demo_nonlinear_bowl.py:612-616   (solution-dependent lumped mass)
//   stiffness(u_n, c3) -> b     :624-626
//   stiffness(v_n, c4) -> b     :627-629
//   mass(w_n = v_n^2, c5) -> b  :630-632  (+ square :603)
// i.e. G is streamed twice, detJ twice, the dofmap four times, b scattered three times.
// K is linear and the constants are per cell, so  c3 K u + c4 K v = K (c3 u + c4 v)  cell by cell,
// and with GLL collocation the mass terms are pointwise in the cell; one pass does all four:
//   b[dof] += [D^T G D (c3 u + c4 v)]_cell + detJ c5 v^2 ;   m[dof] += detJ c2 u
// G and detJ are read once, u and v are gathered once (through the batch plan), b and m are
// pre-reduced in LDS and flushed with one atomic per distinct dof each.
// Same column-per-thread contraction structure as stiffness_plan_kernel (plan.hpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stiffness_plan.hpp"
#include "vecops.hpp"

namespace fus {

// GPRE: slabs of G held in registers (ring, as stiffness_plan_kernel): n = whole slab up front.
// LDS: three cubes only.  The lumped-mass sums are accumulated EARLY, in the region that held the u
// values, and flushed before the flux cubes are written -- while the G loads are still in flight --
// instead of living in a fourth array to the end of the kernel (P = 6: 58 -> 45 KB, 3 workgroups
// per CU instead of 2 once the registers allow it).
// MASS = false: the stiffness part alone, b += K(c3) u + K(c4) v (detJ, c2, c5, m unused).  With GLL
// collocation the mass operator is diagonal, M(c) x = diag(M(c) 1) x, so a driver can precompute the two
// diagonals once and apply the mass terms pointwise in its vector kernel (fus_rk4_stage_nl2_*): no
// detJ stream, no second atomic flush and no extra barriers in the cell pass.
template <typename T, int P, int CPB, int MINW, int GPRE, bool MASS, bool ORDERED, bool RUNS>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()), MINW)
    westervelt_cell_kernel(const T* __restrict__ u_in, const T* __restrict__ v_in, const T* __restrict__ c2,
                           const T* __restrict__ c3, const T* __restrict__ c4, const T* __restrict__ c5,
                           T* __restrict__ b, T* __restrict__ m, const T* __restrict__ G, const T* __restrict__ detJ,
                           const int32_t* __restrict__ nu, const int32_t* __restrict__ udofs,
                           const uint16_t* __restrict__ slot, const T* __restrict__ dphi, int64_t ncell,
                           const int32_t* __restrict__ order, const int32_t* __restrict__ runs, LaunchSignal sig) {
  constexpr int n = P + 1, n2 = n * n, Nd = n2 * n;
  launch_signal_publish(sig);
  constexpr int S = lds_cell_stride<T, P>();
  constexpr int BLOCK = col_block_threads<P, CPB>();
  constexpr int M = CPB * Nd;
  constexpr int SPT = (M + BLOCK - 1) / BLOCK;
  static_assert(GPRE >= 1 && GPRE <= n, "GPRE: slabs of G held in registers");

  // regions: su (combined input cube, later the b accumulator), sfy (u values, then the m
  // accumulator, later flux y), sfz (v values, later flux z)
  __shared__ T sD[n2 + 1];  // + 1: plan_table_store
  __shared__ T su[CPB * S];
  __shared__ T sfy[CPB * S];
  __shared__ T sfz[CPB * S];
  T* const sxu = sfy;
  T* const sxv = sfz;
  // partial sums are accumulated in double (PlanAcc, stiffness_plan.hpp): fp64 kernels alias them onto dead
  // cubes, fp32 kernels get arrays of their own
  constexpr bool OWN_ACC = sizeof(T) != sizeof(PlanAcc);
  __shared__ PlanAcc sacc_b[OWN_ACC ? M : 1];
  __shared__ PlanAcc sacc_m[(OWN_ACC && MASS) ? M : 1];
  PlanAcc* const sm = OWN_ACC ? sacc_m : reinterpret_cast<PlanAcc*>(sfy);
  PlanAcc* const sb = OWN_ACC ? sacc_b : reinterpret_cast<PlanAcc*>(su);

  const int tid = threadIdx.x;
  const unsigned batch = blockIdx.x;
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
  T dj[n];
  T k2 = T(0), k5 = T(0);
  const int64_t cell = plan_row<ORDERED>(row, pos_ld);  // row of the per-cell arrays
  const T* Gc = G + (cell * Nd + t) * 6;
  PlanSlotWord<n> sraw[n];  // narrowed once the gather is on its way (plan.hpp, PlanSlotWord)
  T k3 = T(0), k4 = T(0);
  if (plan_loads_by_all<n>() || active) {
    const uint16_t* sp = slot + pos_ld * Nd + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sraw[ix] = sp[ix * n2];
    if constexpr (MASS) {
      const T* dc = detJ + cell * Nd + t;
#pragma unroll
      for (int ix = 0; ix < n; ++ix) dj[ix] = dc[ix * n2];
      k2 = c2[cell];
      k5 = c5[cell];
    }
    k3 = c3[cell];
    k4 = c4[cell];
#pragma unroll
    for (int ix = 0; ix < GPRE; ++ix) load_g6<T>(Gc + (int64_t)ix * n2 * 6, g[ix]);
  }
  const int packed = nu[batch];
  const int nu_b = packed & 0xffff, nr_b = plan_runs_of<RUNS>(packed);
  plan_table_store<n, n2>(sD, tid, dval);
  batch_dofs_resolve<RUNS, SPT, BLOCK>(rt, ud, M, nu_b, nr_b, tid, reinterpret_cast<int32_t*>(su), mydof);
  {
    T xu[SPT], xv[SPT];
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      xu[r] = u_in[mydof[r]];
      xv[r] = v_in[mydof[r]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sl[ix] = (uint16_t)sraw[ix];
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      if (s < nu_b) {
        sxu[s] = xu[r];
        sxv[s] = xv[r];
      }
    }
  }
  __syncthreads();  // B1

  T w[n];       // combined stiffness input  c3 u + c4 v
  T bextra[n];  // detJ c5 v^2
  T madd[n];    // detJ c2 u   (lumped-mass contribution; dead after the early flush below)
  if (active) {
    T* cu = su + lc * S + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) {
      const T uu = sxu[sl[ix]], vv = sxv[sl[ix]];
      w[ix] = k3 * uu + k4 * vv;
      if constexpr (MASS) {
        bextra[ix] = dj[ix] * k5 * vv * vv;
        madd[ix] = dj[ix] * k2 * uu;
      }
      cu[ix * n2] = w[ix];
    }
  }
  __syncthreads();  // B2: u / v values are dead; the input cube is complete
  if constexpr (MASS) {
    // ---- lumped mass: pre-reduce in the dead u-value region and flush, under the shadow of the G loads
    plan_zero<T, SPT, BLOCK>(sm, nu_b, tid);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int ix = 0; ix < n; ++ix) lds_atomic_add(&sm[sl[ix]], (PlanAcc)madd[ix]);
    }
    __syncthreads();
    plan_flush<T, SPT, BLOCK>(m, mydof, nu_b, tid, sm);
    __syncthreads();  // the m sums have been read: the region becomes the flux-y cube
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
      T vx, vy, vz;
      plan_grad_at<T, n, n2>(qx, dphi, w, dy, dz, cu_y, cu_z, vx, vy, vz);
      const T* gq = g[qx % GPRE];
      fx[qx] = gq[0] * vx + gq[1] * vy + gq[2] * vz;
      cfy[qx * n2] = gq[1] * vx + gq[3] * vy + gq[4] * vz;
      cfz[qx * n2] = gq[2] * vx + gq[4] * vy + gq[5] * vz;
      if constexpr (GPRE < n) {
        if (qx + GPRE < n) load_g6<T>(Gc + (int64_t)(qx + GPRE) * n2 * 6, g[qx % GPRE]);
      }
    }
  }
  __syncthreads();  // B3: the input cube is dead: it becomes the b accumulator
  plan_zero<T, SPT, BLOCK>(sb, nu_b, tid);
  __syncthreads();  // B3.5

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
      T acc = MASS ? bextra[jx] : T(0);
#pragma unroll
      for (int qx = 0; qx < n; ++qx) acc += dphi[qx * n + jx] * fx[qx];
#pragma unroll
      for (int q = 0; q < n; ++q) {
        acc += dyT[q] * cf_y[jx * n2 + q * n];
        acc += dzT[q] * cf_z[jx * n2 + q];
      }
      lds_atomic_add(&sb[sl[jx]], (PlanAcc)acc);
    }
  }
  __syncthreads();  // B4
  plan_flush<T, SPT, BLOCK>(b, mydof, nu_b, tid, sb);
}

// Slabs of G resident per thread in the shipped build: whole slab up to P = 5, ring above.
template <int P>
__host__ __device__ constexpr int westervelt_g_ring() {
  return P <= 3 ? P + 1 : plan_g_ring<P>();
}

template <typename T, int P, bool MASS = true>
inline hipError_t launch_westervelt_cell(const T* u, const T* v, const T* c2, const T* c3, const T* c4, const T* c5,
                                         T* b, T* m, const T* G, const T* detJ, const void* workspace, const T* dphi,
                                         int64_t ncell, hipStream_t stream, bool ordered = false, bool use_runs = false) {
  constexpr int CPB = plan_cells_per_batch<P>();
  if (ncell <= 0) return hipSuccess;
  PlanView pv = plan_view(const_cast<void*>(workspace), P, CPB, ncell);
  constexpr int threads = col_block_threads<P, CPB>();
  constexpr int MINW = 1;
  // stiffness-only: the ring sizes of the stiffness kernel (its register profile + one more gather)
  constexpr int RING = MASS ? westervelt_g_ring<P>() : (P >= 6 ? plan_g_ring<P>() : P + 1);
  const LaunchSignal sig = take_launch_signal(stream);
  plan_dispatch(ordered, use_runs, [&](auto o, auto r) {
    hipLaunchKernelGGL((westervelt_cell_kernel<T, P, CPB, MINW, RING, MASS, decltype(o)::value, decltype(r)::value>),
                       dim3((unsigned)pv.nbatch), dim3(threads), 0, stream, u, v, c2, c3, c4, c5, b, m, G, detJ, pv.nu, pv.udofs,
                       pv.slot, dphi, ncell, pv.order, pv.runs, sig);
  });
  return settle_launch_signal(stream, sig, hipGetLastError());
}

// Fused RK4 stage vector kernel of the Westervelt solver: as rk4_stage_kernel (rk4.hpp) but the
// lumped mass changes every stage, so kv = b / m and m is reset to its steady part m0.
template <typename T>
__global__ void __launch_bounds__(256)
    rk4_stage_nl_kernel(T bw, T aw, int kind, const T* __restrict__ m0, T* __restrict__ m, T* __restrict__ b,
                        T* __restrict__ u, T* __restrict__ v, T* __restrict__ u0, T* __restrict__ v0,
                        T* __restrict__ ku, T* __restrict__ un, int64_t nlocal, int64_t ntotal) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ntotal; i += stride) {
    if (i < nlocal) {
      const T kv = b[i] / m[i];
      if (kind == 2) {  // FIRST (see rk4_stage_kernel, rk4.hpp)
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
    // owned: restart from the steady part (already reverse-scattered); ghosts: restart from zero --
    // they collect this rank's partial sums of the next stage, which the reverse scatter ADDS to the
    // owner (the reference adds m0 after scatter_rev(m), cuda/demo_nonlinear_bowl.py:611-619)
    m[i] = (i < nlocal) ? m0[i] : T(0);
    b[i] = T(0);
  }
}

// The same stage with the mass terms applied POINTWISE from two precomputed diagonals (GLL
// collocation: M(c) x = diag(M(c) 1) x):  w2 = M(c2) 1,  w5 = M(c5) 1, assembled once like m0, so
//   m = m0 + M(c2) u_n = m0 + w2 u_n ,   b += M(c5) v_n^2 = w5 v_n^2        (cuda/demo_nonlinear_bowl.py:603-632)
// and the cell pass is the stiffness part alone.  (u_n, v_n) = the stage's inputs: (u0, v0) for kind FIRST,
// else (un, ku).  No m array is read, written or reverse-scattered any more.
// ``w`` (optional): the combined stiffness input of the NEXT cell pass, w = u_n' + kappa v_n', for media where
// c4 = kappa c3 in every cell (then K(c3) u + K(c4) v = K(c3)(u + kappa v): one plain stiffness apply, one
// gather, one forward halo exchange less); for kind LAST it is formed from the new (u0, v0).
// NT (vectors far larger than the caches): every access non-temporal -- see rk4.hpp: what matters is that no line this pass
// writes stays dirty in the memory-side cache to be written back while the next cell pass runs.
template <typename T, int NT>
__global__ void __launch_bounds__(256)
    rk4_stage_nl2_kernel(T bw, T aw, int kind, const T* __restrict__ m0, const T* __restrict__ w2,
                         const T* __restrict__ w5, T* __restrict__ b, T* __restrict__ u, T* __restrict__ v,
                         T* __restrict__ u0, T* __restrict__ v0, T* __restrict__ ku, T* __restrict__ un,
                         T kappa, T* __restrict__ w, int64_t nlocal, int64_t ntotal) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ntotal; i += stride) {
    auto L = [&](const T* p) { return ld_stream<NT>(p + i); };
    auto S = [&](T* p, T val) { st_stream<NT>(p + i, val); };
    if (i < nlocal) {
      T un_new, vn_new;
      if (kind >= 4) {
        // LEAN set (rk4.hpp: kinds 4, 5, 6, 7 = the four passes of a step; bw = dt / 6, aw = dt / 2): u's accumulator runs one pass
        // ahead (its increments are the vn's), pass 1 writes no accumulator, pass 3 writes the new u into u0: 46 vector touches per
        // step instead of 52
        const T b2 = bw + bw, a4 = aw + aw;
        if (kind == 4) {
          const T u0i = L(u0), v0i = L(v0);
          const T kv = (L(b) + L(w5) * v0i * v0i) / (L(m0) + L(w2) * u0i);
          un_new = u0i + aw * v0i;
          vn_new = v0i + aw * kv;
          S(un, un_new);
          S(ku, vn_new);
        } else if (kind == 5) {
          const T uni = L(un), vn2 = L(ku), u0i = L(u0), v0i = L(v0);
          const T kv = (L(b) + L(w5) * vn2 * vn2) / (L(m0) + L(w2) * uni);
          vn_new = v0i + aw * kv;  // vn3
          un_new = u0i + aw * vn2;
          S(v, (v0i + (vn2 - v0i) * (bw / aw)) + b2 * kv);
          S(u, ((u0i + bw * v0i) + b2 * vn2) + b2 * vn_new);
          S(un, un_new);
          S(ku, vn_new);
        } else if (kind == 6) {
          const T uni = L(un), vn3 = L(ku), u0i = L(u0), v0i = L(v0);
          const T kv = (L(b) + L(w5) * vn3 * vn3) / (L(m0) + L(w2) * uni);
          vn_new = v0i + a4 * kv;  // vn4
          un_new = u0i + a4 * vn3;
          S(v, L(v) + b2 * kv);
          S(u0, L(u) + bw * vn_new);  // the new u: u0 is dead once un4 has been formed
          S(un, un_new);
          S(ku, vn_new);
        } else {  // 7: the new v; (un_new, vn_new) = the next step's first-stage inputs (u0, v0) for the optional w
          const T uni = L(un), vn4 = L(ku);
          const T kv = (L(b) + L(w5) * vn4 * vn4) / (L(m0) + L(w2) * uni);
          vn_new = L(v) + bw * kv;
          S(v0, vn_new);
          un_new = (w != nullptr) ? L(u0) : T(0);
        }
      } else if (kind == 2) {  // FIRST: stage inputs are (u0, v0); u == u0, v == v0, ku == v0
        const T u0i = L(u0), v0i = L(v0);
        const T kv = (L(b) + L(w5) * v0i * v0i) / (L(m0) + L(w2) * u0i);
        S(u, u0i + bw * v0i);
        S(v, v0i + bw * kv);
        un_new = u0i + aw * v0i;
        vn_new = v0i + aw * kv;
        S(un, un_new);
        S(ku, vn_new);
      } else {
        const T uni = L(un), kui = L(ku);
        const T kv = (L(b) + L(w5) * kui * kui) / (L(m0) + L(w2) * uni);
        if (kind == 3) {  // LAST: the next stage's inputs are the new (u0, v0)
          un_new = L(u) + bw * kui;
          vn_new = L(v) + bw * kv;
          S(u0, un_new);
          S(v0, vn_new);
        } else {
          const T ui = L(u) + bw * kui;
          const T vi = L(v) + bw * kv;
          S(u, ui);
          S(v, vi);
          T u0i, v0i;
          if (kind == 1) {
            u0i = ui;
            v0i = vi;
            S(u0, ui);
            S(v0, vi);
          } else {
            u0i = L(u0);
            v0i = L(v0);
          }
          un_new = u0i + aw * kui;
          vn_new = v0i + aw * kv;
          S(un, un_new);
          S(ku, vn_new);
        }
      }
      if (w != nullptr) S(w, un_new + kappa * vn_new);
    }
    S(b, T(0));
  }
}

template <typename T>
inline hipError_t launch_rk4_stage_nl2(T bw, T aw, int kind, const T* m0, const T* w2, const T* w5, T* b, T* u, T* v,
                                       T* u0, T* v0, T* ku, T* un, T kappa, T* w, int64_t nlocal, int64_t ntotal,
                                       hipStream_t stream) {
  if (ntotal <= 0) return hipSuccess;
  int64_t nblocks = (ntotal + 255) / 256;
  if (nblocks > 4096) nblocks = 4096;
  const int nt = vector_stream(ntotal * (int64_t)sizeof(T));
#define FUS_NL2(NT_) \
  hipLaunchKernelGGL((rk4_stage_nl2_kernel<T, NT_>), dim3((unsigned)nblocks), dim3(256), 0, stream, bw, aw, kind, m0, w2, w5, b, u, v, \
                     u0, v0, ku, un, kappa, w, nlocal, ntotal)
  if (nt == 1) FUS_NL2(1); else if (nt == 2) FUS_NL2(2); else FUS_NL2(0);
#undef FUS_NL2
  return hipGetLastError();
}

template <typename T>
inline hipError_t launch_rk4_stage_nl(T bw, T aw, int new_step, const T* m0, T* m, T* b, T* u, T* v, T* u0, T* v0,
                                      T* ku, T* un, int64_t nlocal, int64_t ntotal, hipStream_t stream) {
  if (ntotal <= 0) return hipSuccess;
  int64_t nblocks = (ntotal + 255) / 256;
  if (nblocks > 4096) nblocks = 4096;
  hipLaunchKernelGGL((rk4_stage_nl_kernel<T>), dim3((unsigned)nblocks), dim3(256), 0, stream, bw, aw, new_step, m0, m,
                     b, u, v, u0, v0, ku, un, nlocal, ntotal);
  return hipGetLastError();
}

}  // namespace fus
