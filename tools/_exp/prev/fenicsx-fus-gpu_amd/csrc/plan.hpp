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
#include <type_traits>
#include <vector>
#include <stdint.h>

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
// A signal another fork posted on the same stream between take and settle is handed back by post_launch_signal: it is published at
// once by a one-thread kernel (as halo_comm.hpp's fork does with a displaced signal) -- dropped, its wait kernel would spin out.
template <int = 0>
__global__ void displaced_signal_kernel(uint64_t* flag, uint64_t seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
inline hipError_t settle_launch_signal(hipStream_t stream, const LaunchSignal& sig, hipError_t e) {
  if (e != hipSuccess && sig.flag != nullptr) {
    const LaunchSignal displaced = post_launch_signal(stream, sig.flag, sig.seq);
    if (displaced.flag != nullptr) hipLaunchKernelGGL(displaced_signal_kernel<0>, dim3(1), dim3(1), 0, stream, displaced.flag, displaced.seq);
  }
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

constexpr int kPlanMaxEntries = 4096;  // per batch; slot ids are 16-bit, keys live in LDS

// ---- the preamble every planned kernel shares ------------------------------------------------------------------------
// A workgroup lives ~9 us and a global load under load costs ~1.3 us: the preamble must be TWO round trips deep (the plan's lists and
// per-cell data; then what they point to: x and, with in-kernel geometry, the vertex coordinates), not one per array -- and nothing
// may wait for the G slab before the x gather is on its way.  The compiler waits at the first USE of a loaded register, vmcnt counts in
// order (a wait for one load is a wait for EVERY older one), and after an exec-masked block its counts are the minimum over both paths
// (a later wait for an OLD load then waits for nearly every younger one as well).  Hence the rules (phase clocks of the kernel before:
// profiles/r05p_ablate_geom_phases.log, 54 % of a workgroup's life before its first barrier; A/B: profiles/r05q_*, r05s_*, r05t_*):
//  (1) no load whose value is used inside the conditional block it was issued in (an LDS store of a table, the sign extension of an
//      index): tables are loaded with clamped indices into registers and stored later, by every thread (plan_table_store);
//  (2) nothing in the issue phase depends on nu[batch] (the run words are read speculatively, batch_dofs_issue);
//  (3) what is uniform over the LAUNCH -- the plan carries a cell order (ORDERED), the launch reads the run tables (RUNS) -- is a
//      template parameter, not a pointer test: a runtime select would make every launch wait for the order load, and the two list
//      encodings would share one wait (the library is several objects compiled in parallel for it: Makefile);
//  (4) every thread issues the preamble's loads, no exec-masked block around them.  A thread without a column of its own (the spare
//      threads of the block; the ragged last batch) loads what the last valid cell of the batch loads anyway (plan_load_pos): the same
//      lines, no HBM traffic of its own;
//  (5) the 16-bit slots are loaded as 32-bit words and narrowed (their first use) after the x gather has been issued (PlanSlotWord);
//  (6) where the compiler would sink a load into the conditional block of its only use, an empty asm pins it (batch_dofs_resolve).
// (4)-(5) hold up to degree 8 (P = 7, 8 keep their three waves per SIMD with them: 163-164 VGPRs; fp32 -6 %, fp64 0 ... -4 %,
// profiles/r05y_sweep_ab_preamble_p78.log).  From degree 9 on the loads stay under ``active`` and the slots are narrowed where they are
// loaded: the G slab is a ring of one plane there (little in flight to wait for), and the unconditional forms cost registers the P = 9
// kernels do not have (169 VGPRs, +46 spilled SGPRs: their third wave per SIMD).
template <int n>
__host__ __device__ constexpr bool plan_loads_by_all() {
  return n <= 9;
}
template <int CPB>
__device__ __forceinline__ int64_t plan_load_pos(int64_t cell0, int lc, int64_t ncell) {
  const int64_t p = cell0 + (lc < CPB ? lc : CPB - 1);
  return p < ncell ? p : ncell - 1;
}
// Row of the per-cell arrays of the cell at position ``pos`` of the plan's order.  Issue: unsigned, so that widening it later is
// no use of the loaded register (a sign extension would be hoisted to the load and wait for it).
template <bool ORDERED>
__device__ __forceinline__ uint32_t plan_row_issue(const int32_t* __restrict__ order, int64_t pos) {
  if constexpr (ORDERED) return (uint32_t)order[pos];
  return 0u;
}
template <bool ORDERED>
__device__ __forceinline__ int64_t plan_row(uint32_t row, int64_t pos) {
  if constexpr (ORDERED) return (int64_t)row;
  return pos;
}

// Rule (5): the word a slot is loaded into.
template <int n>
using PlanSlotWord = std::conditional_t<plan_loads_by_all<n>(), uint32_t, uint16_t>;

// Rule (1): store of a small table (dphi, the GLL points / weights) whose entries the first COUNT threads loaded in the preamble, by
// EVERY thread (the others into the spare entry table[COUNT]): under a condition the compiler sinks the load into the block with the
// store, behind every load issued since and a full wait.
template <int n, int COUNT, typename T>
__device__ __forceinline__ void plan_table_store(T* __restrict__ table, int tid, T v) {
  if constexpr (plan_loads_by_all<n>()) {
    table[tid < COUNT ? tid : COUNT] = v;
  } else {
    if (tid < COUNT) table[tid] = v;
  }
}

// Distinct dofs owned by this thread (slots tid, tid + BLOCK, ...), for both plan encodings.
// Phase 1 (issue the global loads; call BEFORE the other HBM loads of the batch so that the x gather, which depends on
// them, can be issued while those are still in flight).  Raw list (!RUNS): one dof per slot, slots clamped (the builder
// padded [nu, M) with a valid dof).  Run-length list (RUNS): thread t holds run t = (first dof, first slot, first slot of
// the next run), read SPECULATIVELY for all kPlanMaxRuns entries of the batch's table -- entries beyond the batch's runs
// hold whatever the allocation held and are never used (phase 2 masks them with nu[batch], which has arrived by then).
struct RunWords {
  int32_t d0, s0, s1;
};
template <bool RUNS, int SPT, int BLOCK>
__device__ __forceinline__ RunWords batch_dofs_issue(const int32_t* __restrict__ ud, const int32_t* __restrict__ rn, int M, int tid,
                                                     int32_t (&mydof)[SPT]) {
  RunWords rw = {0, 0, 0};
  if constexpr (RUNS) {
    constexpr int last = 2 * kPlanMaxRuns - 1;
    const int i0 = 2 * tid, i1 = 2 * tid + 1, i3 = 2 * tid + 3;
    rw.d0 = rn[i0 < last ? i0 : last];
    rw.s0 = rn[i1 < last ? i1 : last];
    rw.s1 = rn[i3 < last ? i3 : last];
  } else {
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      mydof[r] = ud[s < M ? s : 0];
    }
  }
  return rw;
}
// nu[batch]: distinct dofs (low half) and, in a run-coded launch, runs (high half; 0 = this batch's list did not compress)
template <bool RUNS>
__device__ __forceinline__ int plan_runs_of(int packed) {
  return RUNS ? (packed >> 16) : 0;
}
// Phase 2 (RUNS only): the owners of the runs expand them into ``s_dofs`` (an LDS region of >= 4 * nu_b bytes that nothing
// else uses until the next barrier of the caller -- every kernel passes a cube that is written only after its gather), one
// barrier, every thread reads its slots.  A run is <= a few dozen consecutive dofs, so the serial expansion by <= 128
// threads is a fraction of a microsecond; what it buys is 8 bytes per RUN instead of 4 per DOF in HBM (P = 4: 4.1 kB ->
// 0.2 kB per batch).  TRAIL: end with a barrier (for callers that overwrite the region before their next barrier).
template <bool RUNS, int SPT, int BLOCK, bool TRAIL = false>
__device__ __forceinline__ void batch_dofs_resolve(RunWords rw, const int32_t* __restrict__ ud, int M, int nu_b, int nr_b, int tid,
                                                   int32_t* __restrict__ s_dofs, int32_t (&mydof)[SPT]) {
  if constexpr (!RUNS) return;
  // rule (6): an unconditional (empty) use of the run words; without it the compiler sinks their loads into the ``tid < nr_b`` block
  // below, i.e. behind every load the kernel has issued since, with a full wait
  asm volatile("" : "+v"(rw.d0), "+v"(rw.s0), "+v"(rw.s1));
  if (nr_b == 0) {
    // a batch whose list did not compress: its raw list goes through the same LDS region, one round trip late (were it loaded straight
    // into ``mydof``, the x gather of EVERY batch would wait on the merged state of both paths: for all its outstanding loads)
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int s = tid + r * BLOCK;
      if (s < nu_b) s_dofs[s] = ud[s];
    }
  } else if (tid < nr_b) {
    const int s1 = (tid + 1 < nr_b) ? rw.s1 : nu_b;
    for (int s = rw.s0; s < s1; ++s) s_dofs[s] = rw.d0 + (s - rw.s0);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const int s = tid + r * BLOCK;
    mydof[r] = s_dofs[s < nu_b ? s : 0];
  }
  if constexpr (TRAIL) __syncthreads();
}

// Launch a planned kernel compiled for (ORDERED, RUNS): K is a generic lambda taking two std::bool_constant tags.
template <typename K>
inline void plan_dispatch(bool ordered, bool runs, K&& k) {
  if (ordered) {
    if (runs) k(std::true_type{}, std::true_type{});
    else k(std::true_type{}, std::false_type{});
  } else {
    if (runs) k(std::false_type{}, std::true_type{});
    else k(std::false_type{}, std::false_type{});
  }
}

template <typename T>
__device__ __forceinline__ void lds_atomic_add(T* p, T v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace fus
