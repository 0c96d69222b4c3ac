// PEER transport of the ghost-dof halo exchange: no RCCL kernel, no copy engine call, no host in the loop.
//
// Replaces the same closures as halo_comm.hpp (cuda/scatterer.py:104-188 scatter_reverse, :191-277 scatter_forward:
// pack kernel -> device synchronise -> MPI Isend/Irecv on device pointers -> wait -> unpack kernel -> synchronise).
//
// Why (profiles/r02z_overlap_probe.log, r03a_*): RCCL's send/recv kernel needs 264 VGPRs per lane and is not scheduled
// next to an operator launch that holds every vector register of every CU, so an exchange posted "under" interior cells
// ran after them (+41..46 us per apply).  hipMemcpyAsync into peer-mapped memory costs 150 us per hop on this runtime
// (r03a_ipc_probe.log).  What does run next to the operator is a kernel of a few registers.  So each rank maps its
// neighbours' receive arenas once (hipIpcGetMemHandle / hipIpcOpenMemHandle: xGMI stores between GPUs, plain stores
// inside one GPU), and an exchange is two small kernels per rank:
//
//   send  (workgroup = up to 8192 message elements of ONE neighbour): wait until that neighbour has consumed the previous
//         message (credit flag in MY arena), gather the elements from the vector and store them straight into the
//         neighbour's receive buffer (write-through stores), wait for their acknowledgement; the last workgroup of a
//         neighbour's segment publishes the exchange's sequence number in the neighbour's "arrived" flag;
//   recv  wait (bounded) for the "arrived" flag of the chunk's neighbour in MY arena, read the chunk with system-scope
//         loads, store (forward) or atomically add (reverse) it into the vector; the last workgroup of a segment
//         returns the credit to the sender's arena.
//
// Flags carry sequence numbers (exchange 1, 2, ...), so nothing depends on the order in which the processes' hosts
// issue their calls, and the hosts never synchronise with each other after the one-off exchange of the arena handles.
// The arenas are uncached device memory (what RCCL's own peer-to-peer flag buffers are on this architecture), so a reader
// never sees a stale cache line.
//
// Host-side contract (the one MPI's non-blocking collectives and RCCL have too): ALL RANKS POST THE EXCHANGES OF A
// COMMUNICATOR IN THE SAME ORDER.  Send and receive kernels share the communicator's one stream by default, so rank A
// posting (halo 1, halo 2) while rank B posts (halo 2, halo 1) is a cycle: A's receive 1 waits for B's send 1, which is
// queued behind B's receive 2, which waits for A's send 2, queued behind A's receive 1.  (FUS_IPC_TWO_STREAMS=1 puts the
// receive kernels on a stream of their own -- a send never waits for remote data -- at the price of an event edge per
// exchange; the drivers of this repository post in program order on every rank and do not need it.)
//
// Every wait is bounded (FUS_IPC_SPIN_SECONDS, default 20 s of the device's wall clock) and A FAILED EXCHANGE IS LOUD ON
// BOTH SIDES: on a time-out the kernel records it in the halo's status words, stops waiting for the rest of the run,
// drains -- and every flag this halo publishes from then on carries the POISON bit (bit 63).  A neighbour that reads a
// poisoned flag does not consume the arena (stale data), counts it (ST_POISONED), becomes dead itself and poisons what it
// publishes: the failure reaches every rank connected to the one that timed out within a few exchanges, and
// fus_halo_ipc_status() != 0 there.  The solvers / demos / C++ host check it at the end of every rk4() call and raise.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "halo.hpp"

namespace fus {

// Workgroup shape of the exchange kernels (FUS_IPC_THREADS x FUS_IPC_CHUNK message elements per workgroup), measured
// next to the config-3 operator launch with config-4 messages (profiles/r03f_sweep.log, exposed cost of both exchanges
// of an apply): 64..512 threads with 512..4096 elements all cost the same 11-13 us; 256-element workgroups cost 22 (too
// many), 1024-thread workgroups 50 (a workgroup that needs 16 wave slots of one CU at once waits long for them next to
// an operator that fills every CU).  Default 256 x 1024: the footprint of the operator workgroup it displaces.
constexpr int kIpcMaxThreads = 1024;
constexpr int kIpcEpt = 8;  // elements per thread at most: all loads issued before the first store
inline int ipc_env_int(const char* name, int dflt, int lo, int hi) {
  const char* v = std::getenv(name);
  const int x = v ? std::atoi(v) : dflt;
  return x < lo ? lo : (x > hi ? hi : x);
}
// workgroup size and message elements per workgroup of the exchange kernels (chunk <= threads * kIpcEpt)
inline int ipc_threads() {
  static const int t = ipc_env_int("FUS_IPC_THREADS", 256, 64, kIpcMaxThreads) / 64 * 64;
  return t;
}
inline int ipc_chunk() {
  static const int c = ipc_env_int("FUS_IPC_CHUNK", 1024, 64, ipc_threads() * kIpcEpt);
  return c;
}
constexpr int kIpcFlagStride = 64;  // bytes between two flags: one flag per 64-byte line
constexpr uint32_t kIpcMagic = 0x46555349u;  // "FUSI"
constexpr uint64_t kIpcPoison = 1ull << 63;  // in a flag: the publisher's halo has failed (time-out here or upstream)

// flag kinds inside an arena; slot = index of the neighbour in the owners-side (kinds 0, 1) or ghosts-side (2, 3) list
enum IpcFlag { ARRIVED_FWD = 0, CREDIT_REV = 1, ARRIVED_REV = 2, CREDIT_FWD = 3 };
enum IpcStatus { ST_TIMEOUTS = 0, ST_DEAD = 1, ST_POISONED = 2, ST_WORDS = 8 };

struct IpcChunk {
  int32_t nbr;    // neighbour slot on the side the kernel walks
  int32_t count;  // elements in this chunk
  int64_t start;  // element offset inside the side's concatenated index list
};

struct IpcPeer {       // one per neighbour slot and kernel role, in device memory
  char* data;          // send: my segment inside the neighbour's receive buffer (mapped); recv: my receive buffer
  uint64_t* flag_out;  // send: neighbour's "arrived" flag (mapped);  recv: neighbour's credit flag (mapped)
  uint64_t* flag_in;   // send: my credit flag;                         recv: my "arrived" flag
  int64_t seg_off;     // first element of the neighbour's segment in my concatenated list
  int32_t nchunks;     // workgroups of this neighbour's segment
  int32_t pad_;
};

// Memory ordering without fences.  A release / acquire fence at agent or system scope is an L2 write-back / invalidate on
// gfx942 / gfx950 (buffer_wbl2 / buffer_inv): issued by every workgroup of an exchange next to an operator launch whose
// scatter-adds keep the L2 full of dirty lines it cost +75 us per exchange (profiles/r03b_overlap.log).  Instead every
// access to an arena is a RELAXED SYSTEM-SCOPE atomic -- a write-through store / cache-bypassing load (sc0 sc1) on memory
// that is fine-grained anyway -- and the order "data before flag" is kept by waiting for the stores' acknowledgements
// (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the flag store; on the reader's side the data loads
// are issued after the barrier that follows the flag load.
__device__ inline uint64_t ipc_load_flag(const uint64_t* f) {
  return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void ipc_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// FENCED variant of the same protocol (FUS_IPC_FENCED=1 when the halo object is created; bench.py's ladder tries it as "peer:fenced" after
// a PEER transport that came up but failed the run's halo check ON DATA): the conservatively ordered form of the microarchitecture
// guide's hand-off -- producer: every storing wave's s_waitcnt vmcnt(0), the workgroup barrier, then ONE lane's SYSTEM-scope release
// (L2 write-back) + s_waitcnt vmcnt(0) before the relaxed flag store; consumer: ONE relaxed poll, then ONE lane's system-scope acquire
// (cache invalidate) + s_waitcnt vmcnt(0) before the workgroup barrier that precedes the data loads.  The fence-free default rests on
// every arena access being a write-through / cache-bypassing system-scope atomic on uncached memory; should that not hold between two
// DIFFERENT devices (never run on this pool: one GPU per box), this rung costs two fences per workgroup instead of RCCL's 40 us per apply.
__device__ inline void ipc_release_system() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind buffer_wbl2 when it thinks the scoreboard empty)
}
__device__ inline void ipc_acquire_system() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate completes asynchronously: hold the barrier until it has
}

// Wait until *flag >= want.  Bounded: gives up after ``budget`` wall-clock ticks, or at once if an earlier wait of this
// halo has already failed.  A poisoned flag (the publisher's halo is dead) ends the wait at once, is counted and kills this
// halo too.  Returns 1 only if the data behind the flag may be consumed.
__device__ inline int ipc_wait(const uint64_t* flag, uint64_t want, uint64_t* status, uint64_t budget) {
  uint64_t v = ipc_load_flag(flag);
  if (!(v & kIpcPoison) && v >= want) return 1;
  const uint64_t t0 = wall_clock64();
  for (;;) {
    if (v & kIpcPoison) {
      atomicAdd((unsigned long long*)&status[ST_POISONED], 1ull);
      __hip_atomic_store(&status[ST_DEAD], (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
    if (v >= want) return 1;
    if (__hip_atomic_load(&status[ST_DEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - t0 > budget) {
      atomicAdd((unsigned long long*)&status[ST_TIMEOUTS], 1ull);
      __hip_atomic_store(&status[ST_DEAD], (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
    v = ipc_load_flag(flag);
  }
}

// Wait (bounded) for a flag of THIS device (agent scope): the fork flag of the communicator, folded into the first send
// kernel of an apply instead of a wait kernel of its own (halo_comm.hpp comm_fork_join).
__device__ inline void ipc_wait_gate(const uint64_t* gate, uint64_t want, uint64_t* gate_status, uint64_t budget) {
  if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
  if (__hip_atomic_load(&gate_status[ST_DEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const uint64_t t0 = wall_clock64();
  while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > budget) {
      atomicAdd((unsigned long long*)&gate_status[ST_TIMEOUTS], 1ull);
      __hip_atomic_store(&gate_status[ST_DEAD], (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}

// last workgroup of a neighbour's segment publishes ``seq`` in ``flag_out`` -- poisoned if this halo has failed
__device__ inline void ipc_segment_done(unsigned* counter, int nchunks, uint64_t* flag_out, uint64_t seq, const uint64_t* status) {
  const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (done == (unsigned)nchunks - 1u) {  // every other workgroup's stores were acknowledged before it counted itself
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t dead = __hip_atomic_load(&status[ST_DEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(flag_out, dead ? (seq | kIpcPoison) : seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// A kernel's last workgroup publishes ``seq`` in a flag of THIS device: the join flag of the communicator, folded into the
// last receive kernel of an apply instead of a signal kernel of its own.  Every workgroup has waited for its own stores /
// atomics (s_waitcnt vmcnt(0)) before it counts itself.
struct IpcJoin {
  uint64_t* flag;     // nullptr: nothing to publish
  uint64_t seq;
  unsigned* counter;  // zero between launches
};
__device__ inline void ipc_kernel_done(const IpcJoin& j) {
  if (!j.flag) return;
  const unsigned done = __hip_atomic_fetch_add(j.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (done == gridDim.x - 1u) {
    __hip_atomic_store(j.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(j.flag, j.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct IpcGate {
  const uint64_t* flag;  // nullptr: no gate
  uint64_t seq;
  uint64_t* status;      // the communicator's fork / join status words
};

template <typename T>
__device__ inline T ipc_load_elem(const T* p);
template <typename T>
__device__ inline void ipc_store_elem(T* p, T v);
template <>
__device__ inline void ipc_store_elem<double>(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), (uint64_t)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <>
__device__ inline void ipc_store_elem<float>(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <>
__device__ inline double ipc_load_elem<double>(const double* p) {
  const uint64_t u = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return __longlong_as_double((long long)u);
}
template <>
__device__ inline float ipc_load_elem<float>(const float* p) {
  const uint32_t u = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return __uint_as_float(u);
}

// an empty, unconditional use of loaded values (keeps their loads where they were issued)
template <typename V, int N>
__device__ __forceinline__ void ipc_pin(V (&a)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(a[k]));
}

// GATHER: element i of the message is vec[index[i] + offset]; otherwise vec[offset + i] (ghosts numbered owner by owner)
template <typename T, bool GATHER>
__global__ void __launch_bounds__(kIpcMaxThreads)
    ipc_send_kernel(const T* __restrict__ vec, const int64_t* __restrict__ index, int64_t offset,
                    const IpcChunk* __restrict__ chunks, const IpcPeer* __restrict__ peers, unsigned* counters,
                    uint64_t* status, uint64_t seq, uint64_t budget, IpcGate gate, int fenced) {
  const IpcChunk c = chunks[blockIdx.x];
  const IpcPeer p = peers[c.nbr];
  __shared__ int ok;
  if (threadIdx.x == 0) {
    if (gate.flag) ipc_wait_gate(gate.flag, gate.seq, gate.status, budget);  // the caller's stream has produced the vector
    ok = ipc_wait(p.flag_in, seq - 1, status, budget);                       // the neighbour has consumed message seq-1
  }
  __syncthreads();
  if (ok) {
    T* dst = reinterpret_cast<T*>(p.data) + (c.start - p.seg_off);
    // Two round trips for the whole chunk -- every index, then every value -- not two per element: the loads are issued by every
    // thread with a clamped element (a thread beyond the chunk re-reads its first element) instead of under ``e < count``, where the
    // compiler waits for each index inside its own block (16 serial round trips for 8 elements per thread; the same rule as the
    // preamble of the planned kernels, plan.hpp)
    int64_t src[kIpcEpt];
    T v[kIpcEpt];
    if (c.count > 0) {  // block-uniform
#pragma unroll
      for (int k = 0; k < kIpcEpt; ++k) {
        const int e = (int)threadIdx.x + k * (int)blockDim.x;
        const int64_t i = c.start + (e < c.count ? e : 0);
        src[k] = GATHER ? index[i] : i;
      }
#pragma unroll
      for (int k = 0; k < kIpcEpt; ++k) v[k] = vec[src[k] + offset];
      ipc_pin(v);  // (an empty use: the compiler would sink each load into the block of its store again)
#pragma unroll
      for (int k = 0; k < kIpcEpt; ++k) {
        const int e = (int)threadIdx.x + k * (int)blockDim.x;
        if (e < c.count) ipc_store_elem<T>(dst + e, v[k]);
      }
    }
  }
  ipc_stores_done();  // my stores have reached the neighbour's memory before the flag can
  __syncthreads();
  if (threadIdx.x == 0) {
    if (fenced) ipc_release_system();
    ipc_segment_done(&counters[c.nbr], p.nchunks, p.flag_out, seq, status);
  }
}

// MODE: UNPACK_SET (forward: ghosts overwritten) or UNPACK_ADD (reverse: partial sums added into the owners' entries)
template <typename T, int MODE, bool GATHER>
__global__ void __launch_bounds__(kIpcMaxThreads)
    ipc_recv_kernel(T* __restrict__ vec, const int64_t* __restrict__ index, int64_t offset,
                    const IpcChunk* __restrict__ chunks, const IpcPeer* __restrict__ peers, unsigned* counters,
                    uint64_t* status, uint64_t seq, uint64_t budget, IpcJoin join, int fenced) {
  const IpcChunk c = chunks[blockIdx.x];
  const IpcPeer p = peers[c.nbr];
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = ipc_wait(p.flag_in, seq, status, budget);  // the neighbour's message seq is complete
    if (fenced) ipc_acquire_system();
  }
  __syncthreads();
  if (ok) {
    const T* src = reinterpret_cast<const T*>(p.data);
    T v[kIpcEpt];
    int64_t j[kIpcEpt];
    if (c.count > 0) {  // block-uniform; loads by every thread with a clamped element: one round trip for the chunk (see ipc_send_kernel)
#pragma unroll
      for (int k = 0; k < kIpcEpt; ++k) {
        const int e = (int)threadIdx.x + k * (int)blockDim.x;
        const int64_t i = c.start + (e < c.count ? e : 0);
        v[k] = ipc_load_elem<T>(src + i);
        j[k] = GATHER ? index[i] : i;
      }
      ipc_pin(j);
#pragma unroll
      for (int k = 0; k < kIpcEpt; ++k) {
        const int e = (int)threadIdx.x + k * (int)blockDim.x;
        if (e < c.count) {
          if constexpr (MODE == UNPACK_SET)
            vec[j[k] + offset] = v[k];
          else
            unsafeAtomicAdd(vec + j[k] + offset, v[k]);
        }
      }
    }
  }
  if (join.flag) ipc_stores_done();  // my stores / adds into the vector have been performed before the join flag can say so
  __syncthreads();                   // every load of this chunk has been consumed
  if (threadIdx.x == 0) {
    ipc_segment_done(&counters[c.nbr], p.nchunks, p.flag_out, seq, status);
    ipc_kernel_done(join);
  }
}

// ------------------------------------------------------------------------------------------- host side
struct IpcSideInfo {  // what a peer needs to know about one side of my plan
  std::vector<int32_t> ranks;
  std::vector<int64_t> counts, offsets;
};

struct IpcRole {  // device tables of one kernel role (send or recv) over one side
  IpcChunk* chunks = nullptr;
  IpcPeer* peers = nullptr;
  unsigned* counters = nullptr;
  int nchunks = 0, nnbr = 0;
  std::vector<IpcPeer> host_peers;
};

constexpr int32_t kIpcBlobVersion = 2;

// Identity of THIS process, drawn once: a pid alone does not identify an address space (ranks in different PID namespaces
// -- one container per rank -- can share a pid, and the importer would then dereference a foreign virtual address).
struct IpcProcessToken {
  uint64_t w[2];
};
inline const IpcProcessToken& ipc_process_token() {
  static const IpcProcessToken tok = [] {
    IpcProcessToken t{{0, 0}};
    if (FILE* f = std::fopen("/dev/urandom", "rb")) {
      if (std::fread(&t, sizeof t, 1, f) != 1) t = IpcProcessToken{{0, 0}};
      std::fclose(f);
    }
    if (!t.w[0] && !t.w[1]) {  // no /dev/urandom: pid + a high-resolution clock + an address of this image
      t.w[0] = ((uint64_t)getpid() << 32) ^ (uint64_t)(uintptr_t)&ipc_process_token;
      t.w[1] = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    }
    t.w[0] |= 1;  // never all-zero
    return t;
  }();
  return tok;
}

struct IpcBlobHeader {
  uint32_t magic;
  int32_t version;
  int32_t rank;
  int32_t elem_bytes;
  int64_t pid;    // informational (error messages); the address space is identified by ``token``
  IpcProcessToken token;
  char pci_bus_id[32];  // of the exporter's device: ordinals are process-local (HIP_VISIBLE_DEVICES per rank)
  uint64_t base;  // arena address in the exporting process (used directly when importer == exporter process)
  hipIpcMemHandle_t handle;
  int64_t arena_bytes;
  int64_t off_flags, off_recv_fwd, off_recv_rev;
  int32_t n_owner, n_ghost, nmax, device;
  // followed by n_owner x (int64 rank, count, offset), then n_ghost x the same
};

struct IpcState {
  char* arena = nullptr;  // fine-grained device memory: flags, forward receive buffer, reverse receive buffer
  int64_t arena_bytes = 0, off_flags = 0, off_recv_fwd = 0, off_recv_rev = 0;
  int nmax = 1;
  uint64_t* status = nullptr;  // ST_WORDS words, ordinary device memory
  bool connected = false;
  std::vector<void*> opened;  // peer arenas mapped with hipIpcOpenMemHandle
  IpcRole send_fwd, recv_fwd, send_rev, recv_rev;
  uint64_t seq[2] = {0, 0};
  // A receive kernel that waits occupies its hardware queue, and one process has only a few of them (4 by default) for
  // all its streams.  One process per rank: every rank's send precedes its receive, no cycle.  Several ranks in ONE
  // process (tests): rank A's waiting receive can sit in front of rank B's send in a shared queue.  There the receive
  // kernel is posted by *_end, under the in-process contract of the LOCAL transport (every rank's *_begin before any
  // rank's *_end), so it never waits for a send that has not been queued.
  bool defer_recv = false;
  uint64_t pending[2] = {0, 0};
  uint64_t budget = 0;
  hipEvent_t ev_sent = nullptr;
  bool sent_recorded = false, done_recorded = false;  // the events of the exchange in flight were recorded (caller not on the communicator's stream)
  unsigned* join_counter = nullptr;  // workgroups of a receive kernel that have finished (ipc_kernel_done)
  int memory_kind = 0;  // 0 fine-grained, 1 uncached, 2 ordinary
  int fenced = 0;       // FUS_IPC_FENCED=1 at creation: system-scope release / acquire around the flags (ipc_release_system)
};

inline int64_t ipc_align(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

inline uint64_t* ipc_flag_ptr(char* arena, int64_t off_flags, int nmax, int kind, int slot) {
  return reinterpret_cast<uint64_t*>(arena + off_flags + ((int64_t)kind * nmax + slot) * kIpcFlagStride);
}

inline void ipc_role_free(IpcRole& r) {
  if (r.chunks) (void)hipFree(r.chunks);
  if (r.peers) (void)hipFree(r.peers);
  if (r.counters) (void)hipFree(r.counters);
  r = IpcRole();
}

// chunk table of one side: every neighbour's segment cut into pieces of kIpcChunk elements
inline hipError_t ipc_role_init(IpcRole& r, const std::vector<int64_t>& counts, const std::vector<int64_t>& offsets) {
  std::vector<IpcChunk> ch;
  r.nnbr = (int)counts.size();
  r.host_peers.assign(r.nnbr, IpcPeer());
  for (int k = 0; k < r.nnbr; ++k) {
    int n = 0;
    const int chunk = ipc_chunk();
    for (int64_t s = 0; s < counts[k]; s += chunk, ++n)
      ch.push_back(IpcChunk{k, (int32_t)std::min<int64_t>(chunk, counts[k] - s), offsets[k] + s});
    r.host_peers[k].nchunks = n;
    r.host_peers[k].seg_off = offsets[k];
  }
  r.nchunks = (int)ch.size();
  if (r.nchunks == 0) return hipSuccess;
  hipError_t e = hipMalloc(&r.chunks, ch.size() * sizeof(IpcChunk));
  if (e == hipSuccess) e = hipMemcpy(r.chunks, ch.data(), ch.size() * sizeof(IpcChunk), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&r.peers, r.nnbr * sizeof(IpcPeer));
  if (e == hipSuccess) e = hipMalloc(&r.counters, r.nnbr * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(r.counters, 0, r.nnbr * sizeof(unsigned));
  return e;
}

inline hipError_t ipc_role_upload(IpcRole& r) {
  if (r.nchunks == 0) return hipSuccess;
  return hipMemcpy(r.peers, r.host_peers.data(), r.nnbr * sizeof(IpcPeer), hipMemcpyHostToDevice);
}

inline hipError_t ipc_arena_alloc(IpcState& st, int64_t bytes) {
  // Arena memory: UNCACHED device memory first (what RCCL gives its own peer-to-peer flag / LL buffers on gfx942 / gfx950;
  // every access of the exchange kernels bypasses the caches anyway, and nothing else may ever find a stale line of it),
  // then fine-grained, then ordinary memory.  FUS_IPC_MEMORY = uncached | finegrained | coarse picks the first choice
  // (all three export / open through HIP IPC and pass the two-process probe: profiles/r03a_ipc_probe.log).
  const char* force = std::getenv("FUS_IPC_MEMORY");
  const int first = force ? (!std::strcmp(force, "finegrained") ? 0 : !std::strcmp(force, "coarse") ? 2 : 1) : 1;
  const int order[3] = {first, first == 1 ? 0 : 1, first == 2 ? 0 : 2};
  hipError_t e = hipErrorOutOfMemory;
  for (int i = 0; i < 3; ++i) {
    const int kind = order[i];
    if (i > 0 && kind == order[0]) continue;
    void* p = nullptr;
    e = kind == 0   ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained)
        : kind == 1 ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)
                    : hipMalloc(&p, bytes);
    if (e == hipSuccess) {
      st.arena = static_cast<char*>(p);
      st.memory_kind = kind;
      break;
    }
    (void)hipGetLastError();
  }
  if (e != hipSuccess) return e;
  st.arena_bytes = bytes;
  return hipMemset(st.arena, 0, bytes);
}

}  // namespace fus
