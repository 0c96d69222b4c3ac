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
//   nu     int32 [nbatch]           nu | (nr << 16): distinct dofs of the batch, and the number of
//                                   runs in its run table (0 = the batch has none: too many, or no gain)
//   udofs  int32 [nbatch][CPB*Nd]   the sorted distinct dofs, first nu valid (the rest padded)
//   runs   int32 [nbatch][2*kPlanMaxRuns]  the SAME list run-length coded: nr pairs (first dof of the run,
//                                   slot of its first dof) -- a structured numbering gives ~n^2 long
//                                   runs per batch, so a kernel that reads the table instead of the
//                                   list moves 8 nr bytes instead of 4 nu (P = 4: 4100 -> 200) and
//                                   expands it in LDS.  Which of the two a launch reads is the host's
//                                   choice per kernel (bandwidth-bound fp64 builds: the table; fp32
//                                   builds, which are latency-bound: the list)
//   slot   uint16[nbatch][CPB*Nd]   slot of (cell, local dof) = position in udofs[b]
//   order  int32 [nent]             optional cell order: batch b holds the entities order[b*CPB ..]
//                                   (set-up-time locality reordering WITHOUT moving G / detJ / constants:
//                                   the apply kernels index those arrays through it); unused otherwise
//   excl   uint32[nbatch][ceil(CPB*Nd/32)]  optional (fus_plan_mark_exclusive): bit s of batch b = the batch's distinct
//                                   dof number s is touched by NO other batch of this plan (and by nothing else the caller
//                                   declared): its partial sum is finished with a plain load + store instead of an atomic --
//                                   the float-atomic request rate of the chip (~20 G 64-byte requests/s), not HBM, bounds
//                                   the low-intensity kernels (mass: 92 % of that rate, profiles/r03_mass_counters.json)
// Nd = (P+1)^3; the last batch may be ragged (cells >= ncell are never touched).
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>
#include <stdint.h>

#include <type_traits>

#include "stiffness.hpp"

namespace fus {

// A planned operator launch can carry a FORK SIGNAL: the first workgroup of the launch stores ``seq`` in ``flag`` (device
// scope).  Every kernel enqueued before it on the stream has completed when any workgroup of it starts, so this is what a
// one-thread signal kernel in front of the launch would publish -- without that kernel's 2.4 us on the caller's stream
// (halo_comm.hpp: fork / join without events; fus_comm_fork_ex FUS_FORK_ATTACH).
struct LaunchSignal {
  uint64_t* flag;  // nullptr: nothing to publish
  uint64_t seq;
};
__device__ inline void launch_signal_publish(const LaunchSignal& s) {
  if (s.flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(s.flag, s.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// host side: the signal waiting for the next planned launch on a stream (at most one per stream)
struct PendingLaunchSignal {
  hipStream_t stream;
  LaunchSignal sig;
};
inline std::vector<PendingLaunchSignal>& pending_launch_signals() {
  static std::vector<PendingLaunchSignal> v;
  return v;
}
inline std::mutex& pending_launch_signals_mutex() {
  static std::mutex m;
  return m;
}
inline LaunchSignal take_launch_signal(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(pending_launch_signals_mutex());
  auto& v = pending_launch_signals();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].stream == stream) {
      const LaunchSignal s = v[i].sig;
      v.erase(v.begin() + (long)i);
      return s;
    }
  return LaunchSignal{nullptr, 0};
}
// Post a signal for the next planned launch on ``stream``.  A signal already waiting on that stream is handed back (the
// caller publishes it with a signal kernel): at most one per stream, and ANY later launch on the stream may carry it --
// "everything enqueued on the stream before the fork has completed" holds when any later kernel of the stream starts.
inline LaunchSignal post_launch_signal(hipStream_t stream, uint64_t* flag, uint64_t seq) {
  std::lock_guard<std::mutex> lock(pending_launch_signals_mutex());
  auto& v = pending_launch_signals();
  LaunchSignal old{nullptr, 0};
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].stream == stream) {
      old = v[i].sig;
      v.erase(v.begin() + (long)i);
      break;
    }
  v.push_back(PendingLaunchSignal{stream, LaunchSignal{flag, seq}});
  return old;
}
// A launch that took a signal and then failed (hipGetLastError() != hipSuccess) has not published it: put it back, so that
// the next planned launch of the stream -- or fus_comm_fork_flush / the next fork / join -- does.  Without this the
// communicator's wait kernel (or gated send kernel) would spin for FUS_IPC_SPIN_SECONDS and poison the halo.
inline hipError_t settle_launch_signal(hipStream_t stream, const LaunchSignal& sig, hipError_t e) {
  if (e != hipSuccess && sig.flag != nullptr) (void)post_launch_signal(stream, sig.flag, sig.seq);
  return e;
}
// the signal still waiting to be carried for ``flag`` (no planned launch has come), with its stream; {nullptr} if none
inline LaunchSignal take_launch_signal_of(const uint64_t* flag, hipStream_t* stream) {
  std::lock_guard<std::mutex> lock(pending_launch_signals_mutex());
  auto& v = pending_launch_signals();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].sig.flag == flag) {
      const LaunchSignal s = v[i].sig;
      *stream = v[i].stream;
      v.erase(v.begin() + (long)i);
      return s;
    }
  return LaunchSignal{nullptr, 0};
}

constexpr int64_t kPlanMagic = 0x46555350314c414eLL;  // "FUSP1LAN"
constexpr int kPlanMaxRuns = 128;                      // runs of a batch: one per thread of (at least) two waves
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
  int32_t* runs;
  uint16_t* slot;
  int32_t* order;
  uint32_t* excl;
  int64_t excl_words;  // per batch
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
  v.runs = reinterpret_cast<int32_t*>(base + off);
  off += align256(v.nbatch * (int64_t)(2 * kPlanMaxRuns) * (int64_t)sizeof(int32_t));
  v.slot = reinterpret_cast<uint16_t*>(base + off);
  off += align256(v.nbatch * v.entries * (int64_t)sizeof(uint16_t));
  v.order = reinterpret_cast<int32_t*>(base + off);
  off += align256(nent * (int64_t)sizeof(int32_t));
  v.excl = reinterpret_cast<uint32_t*>(base + off);
  v.excl_words = (v.entries + 31) / 32;
  off += align256(v.nbatch * v.excl_words * (int64_t)sizeof(uint32_t));
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

constexpr int kPlanMaxEntries = 4096;  // per batch; slot ids are 16-bit, keys live in LDS

inline hipError_t launch_plan_build_generic(const int32_t* dofmap, int N, int epb, int64_t nent, void* workspace,
                                            hipStream_t stream, int allow_runs = 1,
                                            const int32_t* cell_order = nullptr) {
  if (nent <= 0) return hipSuccess;
  const int M = epb * N;
  if (M < 1 || M > kPlanMaxEntries) return hipErrorInvalidValue;
  PlanView v = plan_view_generic(workspace, N, epb, nent);
  if (v.nbatch > 0x7fffffffLL) return hipErrorInvalidValue;
  int64_t hdr[7] = {kPlanMagic, N, epb, nent, v.nbatch, v.entries, cell_order ? 1 : 0};
  hipError_t e = hipMemcpyAsync(workspace, hdr, sizeof(hdr), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  const int32_t* order = nullptr;
  if (cell_order) {  // keep a copy inside the workspace: the plan is self-contained
    e = hipMemcpyAsync(v.order, cell_order, nent * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    order = v.order;
  }
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
  return hipGetLastError();
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

// Distinct dofs owned by this thread (slots tid, tid + BLOCK, ...), for both plan encodings.
// Phase 1 (issue the global loads; call BEFORE the other HBM loads of the batch so that the x
// gather, which depends on them, can be issued while those are still in flight).  Raw list: one
// dof per slot.  Run-length list: thread t < nr holds run t = (first dof, first slot, end slot).
struct RunWords {
  int32_t d0, s0, s1;
};
// ``rn`` = the batch's run table, or nullptr when this launch reads the lists; ``nr_b`` is then forced to 0
// by the caller (plan_runs_of) so that both phases take the list path.
__device__ __forceinline__ int plan_runs_of(int packed, const int32_t* runs) { return runs != nullptr ? (packed >> 16) : 0; }

template <int SPT, int BLOCK>
__device__ __forceinline__ RunWords batch_dofs_issue(const int32_t* __restrict__ ud, const int32_t* __restrict__ rn,
                                                     int M, int nu_b, int nr_b, int tid, int32_t (&mydof)[SPT]) {
  RunWords rw = {0, 0, 0};
  if (nr_b == 0) {  // the list; the builder padded [nu, M) with a valid dof
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      mydof[r] = ud[s < M ? s : 0];
    }
  } else if (tid < nr_b) {
    rw.d0 = rn[2 * tid];
    rw.s0 = rn[2 * tid + 1];
    rw.s1 = (tid + 1 < nr_b) ? rn[2 * tid + 3] : nu_b;
  }
  return rw;
}
// Phase 2 (run-length plans only): the owners of the runs expand them into ``s_dofs`` (an LDS region
// of >= 4 * nu_b bytes that nothing else uses until the next barrier of the caller -- every kernel
// passes a cube that is written only after its gather), one barrier, every thread reads its slots.
// A run is <= a few dozen consecutive dofs, so the serial expansion by <= 128 threads is a fraction of
// a microsecond; what it buys is 8 bytes per RUN instead of 4 per DOF in HBM (P = 4: 4.1 kB -> 0.2 kB
// per batch).
// TRAIL: end with a barrier (for callers that overwrite the region before their next barrier).
template <int SPT, int BLOCK, bool TRAIL = false>
__device__ __forceinline__ void batch_dofs_resolve(const RunWords& rw, int nu_b, int nr_b, int tid,
                                                   int32_t* __restrict__ s_dofs, int32_t (&mydof)[SPT]) {
  if (nr_b == 0) return;  // block-uniform
  if (tid < nr_b) {
    for (int s = rw.s0; s < rw.s1; ++s) s_dofs[s] = rw.d0 + (s - rw.s0);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    mydof[r] = s_dofs[s < nu_b ? s : 0];
  }
  if constexpr (TRAIL) __syncthreads();
}

template <typename T>
__device__ __forceinline__ void lds_atomic_add(T* p, T v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace fus
