// Builders of the batch plan (layout: plan.hpp): the sort-based build of the dof lists / run tables / slots, and the optional pass
// that marks exclusive dofs.  Host-side set-up kernels, included by fus_gpu.hip alone (the apply kernels need plan.hpp only).
#pragma once

#include "plan.hpp"

namespace fus {

// One workgroup per batch: LDS bitonic sort of (dof << 16 | position) keys, unique flags,
// block scan, write slots + distinct dofs.  M = epb * N entries per batch, M <= M2 (power of 2).
template <int M2>
__global__ void __launch_bounds__(256)
    plan_build_kernel(const int32_t* __restrict__ dofmap, int64_t nent, int N, int epb, int32_t* __restrict__ nu,
                      int32_t* __restrict__ udofs, int32_t* __restrict__ runs, uint16_t* __restrict__ slot,
                      int allow_runs, const int32_t* __restrict__ order) {
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

  for (int i = tid; i < M2; i += 256) {
    uint64_t k = ~0ull;
    if (i < valid) {
      int32_t d;
      if (order) {  // entity at batch position e = i / N is order[ent0 + e]
        const int e = i / N;
        d = dofmap[(int64_t)order[ent0 + e] * N + (i - e * N)];
      } else {
        d = dm[i];
      }
      k = ((uint64_t)(uint32_t)d << 16) | (uint64_t)i;
    }
    keys[i] = k;
  }
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
  int32_t* rn = runs + batch * (int64_t)(2 * kPlanMaxRuns);
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
        ud[s] = (int32_t)d;
        if (use_runs && rstart) {
          rn[2 * r] = (int32_t)d;
          rn[2 * r + 1] = s;
          ++r;
        }
        ++s;
      }
      sl[key & 0xffffu] = (uint16_t)(s - 1);
    }
  }
  // pad [nu, M) with the batch's first dof, so the apply kernels can issue their per-slot loads
  // without first waiting for nu (entries beyond nu are loaded but never used)
  if (valid > 0) {
    const int32_t d0 = (int32_t)(uint32_t)(keys[0] >> 16);
    for (int i = nu_b + tid; i < M; i += 256) ud[i] = d0;
  }
}

constexpr int kPlanHeaderRunBatches = 56;  // byte offset in the plan header (8th int64 word)
__global__ void __launch_bounds__(256) plan_count_runs_kernel(const int32_t* __restrict__ nu, int64_t nbatch, unsigned long long* out) {
  unsigned long long mine = 0;
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nbatch; b += (int64_t)gridDim.x * 256) mine += (nu[b] >> 16) != 0;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

inline hipError_t launch_plan_build_generic(const int32_t* dofmap, int N, int epb, int64_t nent, void* workspace,
                                            hipStream_t stream, int allow_runs = 1,
                                            const int32_t* cell_order = nullptr) {
  if (nent <= 0) return hipSuccess;
  const int M = epb * N;
  if (M < 1 || M > kPlanMaxEntries) return hipErrorInvalidValue;
  PlanView v = plan_view_generic(workspace, N, epb, nent);
  if (v.nbatch > 0x7fffffffLL) return hipErrorInvalidValue;
  int64_t hdr[8] = {kPlanMagic, N, epb, nent, v.nbatch, v.entries, cell_order ? 1 : 0, 0 /* batches with a run table: plan_count_runs */};
  hipError_t e = hipMemcpyAsync(workspace, hdr, sizeof(hdr), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  const int32_t* order = nullptr;
  if (cell_order) {  // keep a copy inside the workspace: the plan is self-contained
    e = hipMemcpyAsync(v.order, cell_order, nent * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    order = v.order;
  }
  // the run tables are read SPECULATIVELY by the apply kernels (all kPlanMaxRuns words of a batch, whatever the builder wrote): clear
  // them, so that no launch ever reads a word nobody wrote (the values are never used; initcheck-style tools would flag the reads)
  e = hipMemsetAsync(v.runs, 0, (size_t)v.nbatch * (2 * kPlanMaxRuns) * sizeof(int32_t), stream);
  if (e != hipSuccess) return e;
  const dim3 grid((unsigned)v.nbatch), block(256);
  if (M <= 256)
    hipLaunchKernelGGL((plan_build_kernel<256>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.runs, v.slot,
                       allow_runs, order);
  else if (M <= 512)
    hipLaunchKernelGGL((plan_build_kernel<512>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.runs, v.slot,
                       allow_runs, order);
  else if (M <= 1024)
    hipLaunchKernelGGL((plan_build_kernel<1024>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.runs, v.slot,
                       allow_runs, order);
  else if (M <= 2048)
    hipLaunchKernelGGL((plan_build_kernel<2048>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.runs, v.slot,
                       allow_runs, order);
  else
    hipLaunchKernelGGL((plan_build_kernel<4096>), grid, block, 0, stream, dofmap, nent, N, epb, v.nu, v.udofs, v.runs, v.slot,
                       allow_runs, order);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(plan_count_runs_kernel, dim3((unsigned)((v.nbatch + 255) / 256 < 1024 ? (v.nbatch + 255) / 256 : 1024)), block, 0, stream,
                     v.nu, v.nbatch, reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + kPlanHeaderRunBatches));
  return hipGetLastError();
}

// How many batches of a built plan carry a run table (the others kept their raw list: too many runs, or no gain).  Waits for the
// build on ``stream``; the apply entry points launch the run-coded form of their kernels only for plans where that pays
// (fus_dispatch.hpp: plan_register).
inline hipError_t plan_run_batches(const void* workspace, hipStream_t stream, int64_t* out) {
  unsigned long long c = 0;
  hipError_t e = hipMemcpyAsync(&c, static_cast<const char*>(workspace) + kPlanHeaderRunBatches, sizeof(c), hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(stream);
  *out = (int64_t)c;
  return e;
}

template <int P>
inline hipError_t launch_plan_build(const int32_t* dofmap, int64_t ncell, void* workspace, hipStream_t stream,
                                    int allow_runs = 1) {
  constexpr int n = P + 1;
  return launch_plan_build_generic(dofmap, n * n * n, plan_cells_per_batch<P>(), ncell, workspace, stream, allow_runs);
}

// ---- exclusive-dof marks (optional second pass over a built plan) ------------------------------------------------------
// use[dof] += 1 for every (batch, distinct dof) of the plan.  ``use`` comes in holding what ELSE touches each dof (0 for a
// launch that runs alone): dofs with use == 1 afterwards belong to exactly one batch and to nothing else.
__global__ void __launch_bounds__(256)
    plan_count_uses_kernel(const int32_t* __restrict__ nu, const int32_t* __restrict__ udofs, int64_t entries, int32_t* use,
                           int64_t ndofs) {
  const int64_t batch = blockIdx.x;
  const int nu_b = nu[batch] & 0xffff;
  const int32_t* ud = udofs + batch * entries;
  for (int s = threadIdx.x; s < nu_b; s += 256) {
    const int32_t d = ud[s];
    if (d >= 0 && d < ndofs) atomicAdd(&use[d], 1);
  }
}
__global__ void __launch_bounds__(256)
    plan_mark_exclusive_kernel(const int32_t* __restrict__ nu, const int32_t* __restrict__ udofs, int64_t entries,
                               const int32_t* __restrict__ use, int64_t ndofs, uint32_t* __restrict__ excl, int64_t words) {
  const int64_t batch = blockIdx.x;
  const int nu_b = nu[batch] & 0xffff;
  const int32_t* ud = udofs + batch * entries;
  uint32_t* ex = excl + batch * words;
  for (int64_t w = threadIdx.x; w < words; w += 256) {
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) {
      const int64_t s = w * 32 + b;
      if (s < nu_b) {
        const int32_t d = ud[s];
        if (d >= 0 && d < ndofs && use[d] == 1) bits |= 1u << b;
      }
    }
    ex[w] = bits;
  }
}
inline hipError_t launch_plan_mark_exclusive(void* workspace, int N, int epb, int64_t nent, int32_t* use, int64_t ndofs,
                                             hipStream_t stream) {
  if (nent <= 0) return hipSuccess;
  PlanView v = plan_view_generic(workspace, N, epb, nent);
  hipLaunchKernelGGL(plan_count_uses_kernel, dim3((unsigned)v.nbatch), dim3(256), 0, stream, v.nu, v.udofs, v.entries, use, ndofs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(plan_mark_exclusive_kernel, dim3((unsigned)v.nbatch), dim3(256), 0, stream, v.nu, v.udofs, v.entries, use, ndofs,
                     v.excl, v.excl_words);
  return hipGetLastError();
}

}  // namespace fus
