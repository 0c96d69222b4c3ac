// Batch plan for the scatter side of the operators (built once per dofmap, on the device).
//
// Why: the chip executes float atomics at the memory side at ~20 G 64-byte requests/s
// (profiles/r01a_counters.json: 37.6 requests per P=4 cell, kernel time == requests / 20 G/s).
// A workgroup handles a batch of CPB consecutive cells; cells of a batch share faces, and in any
// mesh numbering with locality the batch's distinct dofs form long contiguous runs.  The plan
// stores, per batch, the SORTED list of distinct dofs and, per (cell, local dof), the 16-bit slot
// of its dof in that list.  The apply kernel then
//   * gathers x once per distinct dof with consecutive lanes on ascending addresses,
//   * pre-reduces the contributions of the batch in LDS (ds_add),
//   * issues ONE global atomic per distinct dof, consecutive lanes on ascending addresses, so a
//     wave-instruction covers few 64-byte requests.
//
// Workspace layout (caller-owned device buffer, fus_stiffness_plan_bytes() bytes, 256-B aligned):
//   [0, 256)                        header (int64: magic, P, cpb, ncell, nbatch, entries/batch)
//   nu     int32 [nbatch]           nu | (nr << 16): distinct dofs of the batch, and the number
//                                   of runs when the dof list is stored run-length coded (0 = raw)
//   udofs  int32 [nbatch][CPB*Nd]   raw: the sorted distinct dofs, first nu valid;
//                                   runs: nr pairs (first dof of the run, slot of its first dof) --
//                                   a structured numbering gives ~n^2 long runs per batch, so the
//                                   list shrinks from 4 nu bytes to 8 nr bytes (P = 4: 4100 -> 200)
//   slot   uint16[nbatch][CPB*Nd]   slot of (cell, local dof) = position in udofs[b]
// Nd = (P+1)^3; the last batch may be ragged (cells >= ncell are never touched).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "stiffness.hpp"

namespace fus {

constexpr int64_t kPlanMagic = 0x46555350314c414eLL;  // "FUSP1LAN"
constexpr int kPlanMaxRuns = 128;                      // run table of a batch: 2 ints per run, one per thread of a 256-thread workgroup
constexpr int kPlanHeaderBytes = 256;

__host__ __device__ constexpr int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
__host__ __device__ constexpr int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

template <int P>
__host__ __device__ constexpr int plan_cells_per_batch() {
  return default_cells_per_block<P>(256);
}

struct PlanView {
  int64_t nbatch;
  int64_t entries;  // CPB * Nd
  int32_t* nu;
  int32_t* udofs;
  uint16_t* slot;
  int64_t bytes;
};

// Generic plan geometry: ``nent`` entities of ``N`` dofs each, ``epb`` entities per batch.
inline PlanView plan_view_generic(void* workspace, int N, int epb, int64_t nent) {
  PlanView v;
  v.entries = (int64_t)epb * N;
  v.nbatch = (nent + epb - 1) / epb;
  char* base = static_cast<char*>(workspace);
  int64_t off = kPlanHeaderBytes;
  v.nu = reinterpret_cast<int32_t*>(base + off);
  off += align256(v.nbatch * (int64_t)sizeof(int32_t));
  v.udofs = reinterpret_cast<int32_t*>(base + off);
  off += align256(v.nbatch * v.entries * (int64_t)sizeof(int32_t));
  v.slot = reinterpret_cast<uint16_t*>(base + off);
  off += align256(v.nbatch * v.entries * (int64_t)sizeof(uint16_t));
  v.bytes = off;
  return v;
}

inline PlanView plan_view(void* workspace, int P, int cpb, int64_t ncell) {
  const int n = P + 1;
  return plan_view_generic(workspace, n * n * n, cpb, ncell);
}

// One workgroup per batch: LDS bitonic sort of (dof << 16 | position) keys, unique flags,
// block scan, write slots + distinct dofs.  M = epb * N entries per batch, M <= M2 (power of 2).
template <int M2>
__global__ void __launch_bounds__(256)
    plan_build_kernel(const int32_t* __restrict__ dofmap, int64_t nent, int N, int epb, int32_t* __restrict__ nu,
                      int32_t* __restrict__ udofs, uint16_t* __restrict__ slot, int allow_runs) {
  constexpr int CH = M2 / 256;  // elements per thread in the scan phase
  __shared__ uint64_t keys[M2];
  __shared__ int cnt[256];

  const int tid = threadIdx.x;
  const int M = epb * N;
  const int64_t batch = blockIdx.x;
  const int64_t ent0 = batch * epb;
  const int64_t left = nent - ent0;
  const int valid = (int)((left < epb ? left : epb) * N);
  const int32_t* dm = dofmap + ent0 * N;

  for (int i = tid; i < M2; i += 256)
    keys[i] = (i < valid) ? (((uint64_t)(uint32_t)dm[i] << 16) | (uint64_t)i) : ~0ull;
  __syncthreads();

  for (int k = 2; k <= M2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < M2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const uint64_t a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            keys[i] = b;
            keys[l] = a;
          }
        }
      }
      __syncthreads();
    }
  }

  // unique / run-start flags over this thread's contiguous chunk [tid*CH, tid*CH+CH); a run is a
  // maximal stretch of consecutive dof numbers among the distinct dofs
  const int i0 = tid * CH;
  int local = 0;  // distinct dofs | (run starts << 16)
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = i0 + c;
    if (i < valid) {
      const uint32_t d = (uint32_t)(keys[i] >> 16);
      const uint32_t dp = (i == 0) ? 0u : (uint32_t)(keys[i - 1] >> 16);
      const bool first = (i == 0) || (d != dp);
      const bool rstart = first && ((i == 0) || (d != dp + 1u));
      local += (first ? 1 : 0) + (rstart ? 0x10000 : 0);
    }
  }
  cnt[tid] = local;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // inclusive Hillis-Steele scan (both counts at once)
    const int v = (tid >= off) ? cnt[tid - off] : 0;
    __syncthreads();
    cnt[tid] += v;
    __syncthreads();
  }
  const int total = cnt[255];
  const int nu_b = total & 0xffff, nr_b = total >> 16;
  const bool use_runs = allow_runs && (nr_b <= kPlanMaxRuns) && (2 * nr_b < nu_b);
  const int excl = cnt[tid] - local;
  int s = excl & 0xffff;  // slot of the first new dof in this chunk
  int r = excl >> 16;     // index of the first new run in this chunk
  if (tid == 255) nu[batch] = nu_b | ((use_runs ? nr_b : 0) << 16);
  int32_t* ud = udofs + batch * (int64_t)M;
  uint16_t* sl = slot + batch * (int64_t)M;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = i0 + c;
    if (i < valid) {
      const uint64_t key = keys[i];
      const uint32_t d = (uint32_t)(key >> 16);
      const uint32_t dp = (i == 0) ? 0u : (uint32_t)(keys[i - 1] >> 16);
      const bool first = (i == 0) || (d != dp);
      const bool rstart = first && ((i == 0) || (d != dp + 1u));
      if (first) {
        if (!use_runs) ud[s] = (int32_t)d;
        if (use_runs && rstart) {
          ud[2 * r] = (int32_t)d;
          ud[2 * r + 1] = s;
          ++r;
        }
        ++s;
      }
      sl[key & 0xffffu] = (uint16_t)(s - 1);
    }
  }
  // raw lists: pad [nu, M) with the batch's first dof, so the apply kernels can issue their
  // per-slot loads without first waiting for nu (entries beyond nu are loaded but never used)
  if (!use_runs && valid > 0) {
    const int32_t d0 = (int32_t)(uint32_t)(keys[0] >> 16);
    for (int i = nu_b + tid; i < M; i += 256) ud[i] = d0;
  }
}

constexpr int kPlanMaxEntries = 4096;  // per batch; slot ids are 16-bit, keys live in LDS

inline hipError_t launch_plan_build_generic(const int32_t* dofmap, int N, int epb, int64_t nent, void* workspace,
                                            hipStream_t stream, int allow_runs = 1) {
  if (nent <= 0) return hipSuccess;
  const int M = epb * N;
  if (M < 1 || M > kPlanMaxEntries) return hipErrorInvalidValue;
  PlanView v = plan_view_generic(workspace, N, epb, nent);
  if (v.nbatch > 0x7fffffffLL) return hipErrorInvalidValue;
  int64_t hdr[6] = {kPlanMagic, N, epb, nent, v.nbatch, v.entries};
  hipError_t e = hipMemcpyAsync(workspace, hdr, sizeof(hdr), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  const dim3 grid((unsigned)v.nbatch), block(256);
  if (M <= 256)
    hipLaunchKernelGGL((plan_build_kernel<256>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.slot,
                       allow_runs);
  else if (M <= 512)
    hipLaunchKernelGGL((plan_build_kernel<512>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.slot,
                       allow_runs);
  else if (M <= 1024)
    hipLaunchKernelGGL((plan_build_kernel<1024>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.slot,
                       allow_runs);
  else if (M <= 2048)
    hipLaunchKernelGGL((plan_build_kernel<2048>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.slot,
                       allow_runs);
  else
    hipLaunchKernelGGL((plan_build_kernel<4096>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.slot,
                       allow_runs);
  return hipGetLastError();
}

template <int P>
inline hipError_t launch_plan_build(const int32_t* dofmap, int64_t ncell, void* workspace, hipStream_t stream,
                                    int allow_runs = 1) {
  constexpr int n = P + 1;
  return launch_plan_build_generic(dofmap, n * n * n, plan_cells_per_batch<P>(), ncell, workspace, stream, allow_runs);
}

// Distinct dofs owned by this thread (slots tid, tid + BLOCK, ...), for both plan encodings.
// Phase 1 (issue the global loads; call BEFORE the other HBM loads of the batch so that the x
// gather, which depends on them, can be issued while those are still in flight):
template <int SPT, int BLOCK>
__device__ __forceinline__ int batch_dofs_issue(const int32_t* __restrict__ ud, int M, int nr_b, int tid,
                                                int32_t (&mydof)[SPT]) {
  int rt = 0;
  if (nr_b == 0) {  // raw list; the builder padded [nu, M) with a valid dof
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      mydof[r] = ud[s < M ? s : 0];
    }
  } else if constexpr (BLOCK >= 2 * kPlanMaxRuns) {  // one run-table word per thread
    rt = ud[tid < 2 * nr_b ? tid : 0];
  }
  return rt;
}
// Phase 2 (run-length plans only): stage the run table in LDS and locate each slot's run.
template <int SPT, int BLOCK>
__device__ __forceinline__ void batch_dofs_resolve(int rt, int nu_b, int nr_b, int tid, int* __restrict__ s_runs,
                                                   int32_t (&mydof)[SPT]) {
  if (nr_b == 0) return;  // block-uniform
  if constexpr (BLOCK < 2 * kPlanMaxRuns) return;  // such builds only accept raw plans (host-checked)
  if (tid < 2 * kPlanMaxRuns) s_runs[tid] = rt;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    const int sc = s < nu_b ? s : 0;
    int lo = 0, hi = nr_b - 1;
    while (lo < hi) {  // largest run whose first slot is <= sc
      const int mid = (lo + hi + 1) >> 1;
      if (s_runs[2 * mid + 1] <= sc)
        lo = mid;
      else
        hi = mid - 1;
    }
    mydof[r] = s_runs[2 * lo] + (sc - s_runs[2 * lo + 1]);
  }
}

template <typename T>
__device__ __forceinline__ void lds_atomic_add(T* p, T v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Planned stiffness apply: same contraction structure as stiffness_col_kernel (stiffness.hpp),
// gather / scatter through the batch plan.
template <typename T, int P, int CPB, bool ALIAS, bool PADLDS, int MINW, int GMODE = 0>
__global__ void __launch_bounds__((col_block_threads<P, CPB>()), MINW)
    stiffness_plan_kernel(const T* __restrict__ x, const T* __restrict__ cell_constants, T* __restrict__ y,
                          const T* __restrict__ G, const int32_t* __restrict__ nu,
                          const int32_t* __restrict__ udofs, const uint16_t* __restrict__ slot,
                          const T* __restrict__ dphi, int64_t ncell, int xcd_remap,
                          const T* __restrict__ wratio = nullptr) {
  constexpr int n = P + 1, n2 = n * n, Nd = n2 * n;
  constexpr int S = PADLDS ? lds_cell_stride<T, P>() : Nd;
  constexpr int BLOCK = col_block_threads<P, CPB>();
  constexpr int M = CPB * Nd;
  constexpr int SPT = (M + BLOCK - 1) / BLOCK;  // distinct-dof slots per thread (upper bound)

  // LDS: three cubes per cell (u, f_y, f_z) + the batch's distinct-dof values.  Lifetimes:
  //   x values [load, B2)   u cube [B1, B3)   f_y/f_z [B2, B4)   y partial sums [B3, end)
  // ALIAS: x values live in the f_y region and the y sums in the u region (one more barrier).
  __shared__ T sD[n2];
  __shared__ T su[CPB * S];
  __shared__ T sfy[CPB * S];
  __shared__ T sfz[CPB * S];
  __shared__ T sxy_own[ALIAS ? 1 : M];
  T* const sx = ALIAS ? sfy : sxy_own;  // x values of the batch's distinct dofs
  T* const sy = ALIAS ? su : sxy_own;   // their y partial sums

  const int tid = threadIdx.x;
  const unsigned batch = remap_block(blockIdx.x, gridDim.x, xcd_remap);
  const int lc = tid / n2;
  const int t = tid - lc * n2;
  const int ty = t / n, tz = t - ty * n;
  const int64_t cell = (int64_t)batch * CPB + lc;
  const bool active = (lc < CPB) && (cell < ncell);
  __shared__ int s_runs[2 * kPlanMaxRuns];
  const int packed = nu[batch];
  const int nu_b = packed & 0xffff, nr_b = packed >> 16;
  const int32_t* ud = udofs + (int64_t)batch * M;

  if (tid < n2) sD[tid] = dphi[tid];

  // ---- issue every HBM load of the batch up front ---------------------------------------------
  int32_t mydof[SPT];
  const int rt = batch_dofs_issue<SPT, BLOCK>(ud, M, nr_b, tid, mydof);
  uint16_t sl[n];
  T g[(GMODE & 64) ? 1 : n][6];
  T wr_aff[(GMODE & 64) ? n : 1];
  T coeff = T(0);
  if (active) {
    const uint16_t* sp = slot + cell * Nd + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) sl[ix] = sp[ix * n2];
    if constexpr (GMODE & 64) {
      // affine cells (opt-in): the geometric factor of an affine cell is one
      // symmetric 3x3 matrix times the quadrature weight, G[c][q] = G[c][0] * (w_q / w_0), so only
      // the first record of the cell (48 B instead of 48 n^3 B) is read
      // g[0] holds the cell's record, g[1][0..n-1 mod 6]... (see phase 1): only 6 + n values are
      // kept live, the per-slab factors are formed where they are used
      load_g6<T>(G + cell * Nd * 6, g[0]);
#pragma unroll
      for (int ix = 0; ix < n; ++ix) wr_aff[ix] = wratio[ix * n2 + t];
    } else if constexpr (GMODE & 8) {  // ABLATION (timing only): no G loads
#pragma unroll
      for (int ix = 0; ix < n; ++ix)
#pragma unroll
        for (int k = 0; k < 6; ++k) g[ix][k] = T(k + 1);
    } else if constexpr (GMODE & 1) {
      // EXPERIMENT ONLY (tools/ab_stiffness.py): G pre-transposed to [cell][6][n^3], every load a
      // fully coalesced 8-byte-per-lane access -- prices the AoS access shape, not a product path
      const T* Gs = G + cell * Nd * 6 + t;
#pragma unroll
      for (int ix = 0; ix < n; ++ix)
#pragma unroll
        for (int k = 0; k < 6; ++k) g[ix][k] = Gs[(int64_t)k * Nd + ix * n2];
    } else if constexpr ((GMODE & 32) == 0) {
      const T* Gc = G + (cell * Nd + t) * 6;
#pragma unroll
      for (int ix = 0; ix < n; ++ix) load_g6<T>(Gc + (int64_t)ix * n2 * 6, g[ix]);
    }
    coeff = cell_constants[cell];
  }
  batch_dofs_resolve<SPT, BLOCK>(rt, nu_b, nr_b, tid, s_runs, mydof);
  T xv[SPT];
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    if constexpr (GMODE & 4)
      xv[r] = T(mydof[r]);  // ABLATION (timing only): no x gather
    else
      xv[r] = x[mydof[r]];
  }
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    if (s < nu_b) sx[s] = xv[r];
  }
  __syncthreads();

  T u[n];
  if (active) {
    T* cu = su + lc * S + t;
#pragma unroll
    for (int ix = 0; ix < n; ++ix) {
      u[ix] = sx[sl[ix]];
      cu[ix * n2] = u[ix];
    }
  }
  if constexpr ((GMODE & 128) == 0) __syncthreads();  // (GMODE & 128: barrier ablation, timing only)

  if constexpr (!ALIAS) {  // all reads of the x values are done: the buffer becomes the y accumulator
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
    // GMODE & 16: volatile LDS reads keep hipcc from fusing pairs into ds_read2_b64, which moves
    // 16 B/lane in 8 LDS cycles where two ds_read_b64 take 4 (MI355X_MICROARCH.md, LDS table)
    using LT = typename std::conditional<(GMODE & 16) != 0, const volatile __attribute__((address_space(3))) T,
                                         const T>::type;
    LT* cu_y = (LT*)(su + lc * S + tz);
    LT* cu_z = (LT*)(su + lc * S + ty * n);
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
      T gq[6];
      if constexpr ((GMODE & 64) != 0) {
        const T cw = wr_aff[qx];
#pragma unroll
        for (int k = 0; k < 6; ++k) gq[k] = g[0][k] * cw;
      } else {
        if constexpr ((GMODE & 32) != 0)  // stream this slab of G now (fewer live registers)
          load_g6<T>(G + (cell * Nd + t) * 6 + (int64_t)qx * n2 * 6, g[qx]);
#pragma unroll
        for (int k = 0; k < 6; ++k) gq[k] = g[(GMODE & 64) ? 0 : qx][k];
      }
      fx[qx] = coeff * (gq[0] * vx + gq[1] * vy + gq[2] * vz);
      cfy[qx * n2] = coeff * (gq[1] * vx + gq[3] * vy + gq[4] * vz);
      cfz[qx * n2] = coeff * (gq[2] * vx + gq[4] * vy + gq[5] * vz);
    }
  }
  if constexpr ((GMODE & 128) == 0) __syncthreads();
  if constexpr (ALIAS) {  // the u cube is dead: zero it as the y accumulator
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
    using LT2 = typename std::conditional<(GMODE & 16) != 0, const volatile __attribute__((address_space(3))) T,
                                          const T>::type;
    LT2* cf_y = (LT2*)(sfy + lc * S + tz);
    LT2* cf_z = (LT2*)(sfz + lc * S + ty * n);
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
  __syncthreads();

  // one global atomic per distinct dof; consecutive lanes -> ascending, mostly contiguous addresses
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    if constexpr (GMODE & 2) {  // ABLATION (timing only): no global atomics; keep the value alive
      if (s < nu_b && sy[s] == T(-1.2345e300)) y[mydof[r]] = sy[s];
    } else {
      if (s < nu_b) unsafeAtomicAdd(y + mydof[r], sy[s]);
    }
  }
}

template <typename T, int P, bool ALIAS, bool PADLDS, int MINW, int GMODE = 0, int TARGET = 256>
inline hipError_t launch_stiffness_plan(const T* x, const T* cc, T* y, const T* G, const void* workspace,
                                        const T* dphi, int64_t ncell, int xcd_remap, hipStream_t stream,
                                        const T* wratio = nullptr) {
  constexpr int CPB = default_cells_per_block<P>(TARGET);
  if (ncell <= 0) return hipSuccess;
  PlanView v = plan_view(const_cast<void*>(workspace), P, CPB, ncell);
  constexpr int threads = col_block_threads<P, CPB>();
  hipLaunchKernelGGL((stiffness_plan_kernel<T, P, CPB, ALIAS, PADLDS, MINW, GMODE>), dim3((unsigned)v.nbatch), dim3(threads), 0,
                     stream, x, cc, y, G, v.nu, v.udofs, v.slot, dphi, ncell, xcd_remap, wratio);
  return hipGetLastError();
}

}  // namespace fus
